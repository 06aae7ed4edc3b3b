#!/bin/bash
# round 4, call t: stream neighbour search on TRAINED tables (fallback rows, time) + phases of the select from compact lists
O=gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp
for st in 400 3000; do
  STEPS=$st OEA_TOPK_DEBUG=1 timeout 300 python tools/_exp/knn_trained.py 2>&1 | grep -E "refresh|redone" | tail -4
  STEPS=$st OEA_TOPK_STREAM=0 OEA_TOPK_DEBUG=1 timeout 300 python tools/_exp/knn_trained.py 2>&1 | grep -E "refresh|redone" | tail -4 | sed 's/^/   stream off: /'
done > $O/trained.log 2>&1
for s in 1 2 3 4 5 0; do OEA_TOPK_SELECT_STOP=$s timeout 120 python tools/_exp/knn_abl.py 2>/dev/null | tail -1; done > $O/select_phases.log 2>&1
cat $O/trained.log $O/select_phases.log
