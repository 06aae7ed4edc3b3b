#!/bin/bash
# round 4, call x: bf16 evaluation sweep with the candidates' operand... (B) in registers: tests + timings with / without
O=gpurun_out/r04x; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "bf16 or greedy or rank_eval or csls" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
timeout 600 python tools/_exp/bf16_eval.py 2>&1 | grep -E "^eval|^CSLS|IDENT|MISM" > $O/breg_on.log
OEA_BF16_BREG=0 timeout 600 python tools/_exp/bf16_eval.py 2>&1 | grep -E "^eval|^CSLS|IDENT|MISM" > $O/breg_off.log
tail -3 $O/pytest.log; echo ON; cat $O/breg_on.log; echo OFF; cat $O/breg_off.log
