#!/bin/bash
# round 4, call v: kernel trace of the stream neighbour search with the overflow pool (random 100,000 rows)
O=gpurun_out/r04v; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/_exp/knn_abl.py 2>/dev/null | tail -1 > $O/time.log
tools/prof.sh trace r04v_trace -- python tools/_exp/knn_abl.py > $O/trace.log 2>&1
cat $O/time.log
