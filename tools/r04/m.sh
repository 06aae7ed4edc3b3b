#!/bin/bash
# round 4, call m: epilogue ablation of the bf16 symmetric sweep (select stopped after phase 1 so that the sweep dominates)
O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
for x in 0 1 2 3 4 8 12 16 32 48 64 76; do
  OEA_TOPK_SELECT_STOP=1 OEA_TOPK_EXP=$x timeout 120 python tools/_exp/knn_abl.py 2>/dev/null | tail -1
done > $O/abl.log 2>&1
OEA_TOPK_EXP=0 timeout 120 python tools/_exp/knn_abl.py 2>/dev/null | tail -1 >> $O/abl.log
cat $O/abl.log
