#!/bin/bash
# round 6, call r: where list_select_kernel's time goes (OEA_TOPK_SELECT_STOP leaves after phase 1 lengths + scan, 2 gather, 3 histogram +
# bucket, 4 threshold bucket ranked, 5 bitmap set; 0 = all) -- wall time of the whole search, differences = phases
set -u
for S in 1 2 3 4 5 0; do
  echo -n "stop=$S  "
  OEA_TOPK_SELECT_STOP=$S KNN_QUICK=1 python tools/_exp/knn_time.py 2>&1 | tail -1
done
