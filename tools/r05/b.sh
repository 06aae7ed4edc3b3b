#!/bin/bash
# round 5, call b: K-blocked certificate + XCD-aware item order -- parity at d = 500 / 1,200, ablation at 70,000^2 x {100, 300, 1200}
set -u
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16_prefilter or csls_means_one_sweep or rank_eval_bit_exact or greedy_alignment_takes" 2>&1 | tail -15 ) > $O/pytest_d1200.log 2>&1
( OEA_XCD_MAP=1 timeout 600 python tools/_exp/eval_shapes.py "100,300,1200" 0.6 check 2>&1 | grep -v amdgpu.ids ) > $O/xcd1.log 2>&1
( OEA_XCD_MAP=0 timeout 600 python tools/_exp/eval_shapes.py "100,300,1200" 0.6 2>&1 | grep -v amdgpu.ids ) > $O/xcd0.log 2>&1
( OEA_XCD_MAP=1 timeout 600 python tools/_exp/eval_shapes.py "1200" 8 check 2>&1 | grep -v amdgpu.ids ) > $O/hard1200.log 2>&1
( timeout 600 python tools/_exp/alinet_eval.py 70000 0.6 2>&1 | tail -12 ) > $O/alinet_eval.log 2>&1
tail -5 $O/pytest_d1200.log; echo "== xcd map on"; cat $O/xcd1.log; echo "== xcd map off"; cat $O/xcd0.log; echo "== hard"; cat $O/hard1200.log; echo "== alinet_eval"; cat $O/alinet_eval.log
