#!/bin/bash
# round 4, call za: bucketing with several workgroups per target tile: tests, 100K / 15K timings
O=gpurun_out/r04za; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_reference_fullsize.py -m gpu -q -x -k "neighbour or topk or knn" 2>&1 | tail -3 ) > $O/pytest.log 2>&1
timeout 300 python tools/_exp/knn_bf16.py 2>&1 | grep "n=" > $O/knn.log
( timeout 300 python tools/_exp/knn_15k.py 2>&1 | grep -E "random|trained"; OEA_TOPK_SYM_MIN=8192 timeout 300 python tools/_exp/knn_15k.py 2>&1 | grep -E "random|trained" ) > $O/knn15k.log 2>&1
STEPS=400 timeout 300 python tools/_exp/knn_trained.py 2>&1 | grep refresh | tail -1 >> $O/knn.log
tail -2 $O/pytest.log; cat $O/knn.log $O/knn15k.log
