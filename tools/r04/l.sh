#!/bin/bash
# round 4, call l: neighbour search with the bf16 sweep + exact band resolve: tests, timing with / without, kernel trace
O=gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_reference_fullsize.py -m gpu -q -x -k "neighbour or topk or greedy or knn" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
OEA_TOPK_BF16=1 timeout 300 python tools/_exp/knn_bf16.py > $O/knn_bf16_on.log 2>&1
OEA_TOPK_BF16=0 timeout 300 python tools/_exp/knn_bf16.py > $O/knn_bf16_off.log 2>&1
tools/prof.sh trace r04l_knn_trace -- python tools/_exp/knn_bf16.py > $O/trace.log 2>&1
tail -6 $O/pytest.log; cat $O/knn_bf16_on.log $O/knn_bf16_off.log; tail -3 $O/trace.log
