#!/bin/bash
# round 5, call k: producer / consumer wave specialisation of the big pipeline (OEA_BF16_BIG_MODE=2) against mode 1
set -u
O=gpurun_out/r05k; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16_prefilter or csls_means_one_sweep or rank_eval_bit_exact or greedy_alignment_takes or csls_pipeline" 2>&1 | tail -6 ) > $O/pytest.log 2>&1
( OEA_BF16_BIG_MODE=2 timeout 600 python tools/_exp/eval_shapes.py "300,1200" 8 check 2>&1 | grep -v amdgpu.ids ) > $O/mode2.log 2>&1
( OEA_BF16_BIG_MODE=1 timeout 600 python tools/_exp/eval_shapes.py "300,1200" 8 2>&1 | grep -v amdgpu.ids ) > $O/mode1.log 2>&1
OEA_BF16_BIG_MODE=2 tools/prof.sh trace r05k -- python tools/_exp/eval1200_trace.py 1200 2
tail -3 $O/pytest.log; echo "== mode 2"; cat $O/mode2.log; echo "== mode 1"; cat $O/mode1.log; head -4 $O/trace_stats.csv | cut -c1-70,200-320
