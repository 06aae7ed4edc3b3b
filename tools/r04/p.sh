#!/bin/bash
# round 4, call p: stage A of the stream rewrite of the neighbour sweep: the stream sweep alone against the append sweep
O=gpurun_out/r04p; mkdir -p $O
export TMPDIR=/tmp
( OEA_TOPK_SELECT_STOP=1 timeout 120 python tools/_exp/knn_abl.py 2>/dev/null | tail -1
  OEA_TOPK_STREAM_EXP=1 OEA_TOPK_SELECT_STOP=1 timeout 120 python tools/_exp/knn_abl.py 2>/dev/null | tail -1 ) > $O/stream.log 2>&1
cat $O/stream.log
