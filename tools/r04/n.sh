#!/bin/bash
# round 4, call n: CSLS means on the bf16 sweep: tests + the experiment script
O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_reference_fullsize.py -m gpu -q -x -k "csls or greedy or eval" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
timeout 600 python tools/_exp/bf16_eval.py > $O/bf16_eval.log 2>&1
tail -6 $O/pytest.log; tail -12 $O/bf16_eval.log
