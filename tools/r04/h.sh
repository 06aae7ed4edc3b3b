#!/bin/bash
# round 4, call h: certified bf16 prefilter of the evaluation sweep: error bound, exactness, timing
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/_exp/bf16_eval.py > $O/bf16_eval.log 2>&1
cat $O/bf16_eval.log | tail -40
