#!/bin/bash
# round 4, call w: the stream form at the 15K shape (OEA_TOPK_SYM_MIN=8192) against the strip path
O=gpurun_out/r04w; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/_exp/knn_15k.py 2>&1 | grep -E "random|trained|redone|overflow"
  OEA_TOPK_SYM_MIN=8192 timeout 300 python tools/_exp/knn_15k.py 2>&1 | grep -E "random|trained"
  OEA_TOPK_SYM_MIN=8192 OEA_TOPK_DEBUG=1 timeout 300 python tools/_exp/knn_15k.py 2>&1 | grep -E "redone|overflow" | sort | uniq -c | sort -rn | head -6 ) > $O/knn15k.log 2>&1
cat $O/knn15k.log
