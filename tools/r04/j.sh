#!/bin/bash
# round 4, call j: bf16 prefilter wired into greedy_alignment: whole GPU suite + the experiment script
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_gpu.log 2>&1
timeout 600 python tools/_exp/bf16_eval.py > $O/bf16_eval.log 2>&1
tail -6 $O/pytest_gpu.log; tail -10 $O/bf16_eval.log
