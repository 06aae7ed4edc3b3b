#!/bin/bash
# round 6, call l: neighbour sweep's record epilogue without a branch per accumulator (stream_tile_fast)
set -u
for F in 1 0; do
  echo "== OEA_TOPK_STREAM_FAST=$F"
  OEA_TOPK_STREAM_FAST=$F KNN_QUICK=1 python tools/_exp/knn_time.py 2>&1 | tail -1
done
python -m pytest tests/test_fullsize_gpu.py tests/test_reference_fullsize.py tests/test_kernels_gpu.py -x -q -m gpu -k "neighbour or knn or topk" 2>&1 | tail -3
