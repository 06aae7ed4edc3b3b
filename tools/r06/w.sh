#!/bin/bash
# round 6, call w: the sweeps' per-tile loads (row bounds / column means of the rank sweep, candidate thresholds of the neighbour sweep) one
# tile ahead; A / B against the previous build on the same box (OPENEA_HIP_LIB)
B=$PWD/openea_amd/csrc/_build/libopenea_hip_before.so
for L in new old new old; do
  echo "== $L"
  if [ $L = old ] && [ -f $B ]; then export OPENEA_HIP_LIB=$B; else unset OPENEA_HIP_LIB; fi
  python tools/_exp/eval_time.py 2>&1 | grep eval
  KNN_QUICK=1 python tools/_exp/knn_time.py 2>&1 | tail -1
done
unset OPENEA_HIP_LIB
python -m pytest tests/test_fullsize_gpu.py tests/test_reference_fullsize.py tests/test_kernels_gpu.py -x -q -m gpu -k "neighbour or knn or topk or bf16 or rank_eval or csls" 2>&1 | tail -3
