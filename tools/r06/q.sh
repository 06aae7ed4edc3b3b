#!/bin/bash
# round 6, call q: the neighbour search's sample strip on the bf16 split
set -u
for B in 1 0; do
  OEA_TOPK_BF16_STRIP=$B KNN_QUICK=1 python tools/_exp/knn_time.py 2>&1 | tail -1
  OEA_TOPK_BF16_STRIP=$B python tools/_exp/knn_asym.py 2>&1 | grep kNN
done
python -m pytest tests/test_fullsize_gpu.py tests/test_reference_fullsize.py tests/test_kernels_gpu.py -x -q -m gpu -k "neighbour or knn or topk" 2>&1 | tail -3
