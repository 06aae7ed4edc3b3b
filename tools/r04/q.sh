#!/bin/bash
# round 4, call q: resident-B pipeline in the bf16 evaluation sweep: tests + timings with / without
O=gpurun_out/r04q; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "bf16 or greedy or rank_eval" 2>&1 | tail -6 ) > $O/pytest.log 2>&1
timeout 600 python tools/_exp/bf16_eval.py 2>&1 | grep -E "^eval|^CSLS|IDENT|MISM" > $O/res_on.log
OEA_BF16_RES=0 timeout 600 python tools/_exp/bf16_eval.py 2>&1 | grep -E "^eval|^CSLS|IDENT|MISM" > $O/res_off.log
tail -4 $O/pytest.log; echo ON; cat $O/res_on.log; echo OFF; cat $O/res_off.log
