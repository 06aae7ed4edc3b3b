#!/bin/bash
# round 6, call w: the padding k-step of an odd number of 16-k steps skipped in the register-operand sweeps (dim = 100: 7 of 8)
python tools/_exp/eval_time.py 2>&1 | grep "x 100 "
KNN_QUICK=1 python tools/_exp/knn_time.py 2>&1 | tail -1
python tools/_exp/knn_asym.py 2>&1 | grep "x 100,"
python -m pytest tests/test_fullsize_gpu.py tests/test_reference_fullsize.py tests/test_kernels_gpu.py -x -q -m gpu -k "neighbour or knn or topk or bf16 or rank_eval or csls" 2>&1 | tail -3
