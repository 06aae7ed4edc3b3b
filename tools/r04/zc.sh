#!/bin/bash
# round 4, call zc: final numbers of the neighbour search (trained 3,000 steps, kernel trace at 100,000 rows and at the 15K shape)
O=gpurun_out/r04zc; mkdir -p $O
export TMPDIR=/tmp
STEPS=3000 timeout 200 python tools/_exp/knn_trained.py 2>&1 | grep refresh | tail -1 > $O/trained.log
timeout 200 python tools/_exp/knn_15k.py 2>&1 | grep -E "random|trained" >> $O/trained.log
timeout 200 tools/prof.sh trace r04zc_trace100k -- python tools/_exp/knn_abl.py
cat $O/trained.log
