#!/bin/bash
# round 5, call c: 256 x 128 tiles + three-stage ring (rank sweep, CSLS means sweep) for rows wider than 128
set -u
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16_prefilter or csls_means_one_sweep or rank_eval_bit_exact or greedy_alignment_takes or csls_pipeline" 2>&1 | tail -15 ) > $O/pytest.log 2>&1
( OEA_BF16_BIG=1 timeout 600 python tools/_exp/eval_shapes.py "300,1200" 0.6 check 2>&1 | grep -v amdgpu.ids ) > $O/big1.log 2>&1
( OEA_BF16_BIG=0 timeout 600 python tools/_exp/eval_shapes.py "300,1200" 0.6 2>&1 | grep -v amdgpu.ids ) > $O/big0.log 2>&1
( OEA_BF16_BIG=1 timeout 600 python tools/_exp/eval_shapes.py "1200" 8 check 2>&1 | grep -v amdgpu.ids ) > $O/hard1200.log 2>&1
( OEA_BF16_BIG=1 OEA_XCD_MAP=0 timeout 600 python tools/_exp/eval_shapes.py "1200" 0.6 2>&1 | grep -v amdgpu.ids ) > $O/big1_xcd0.log 2>&1
tail -5 $O/pytest.log; echo "== big on"; cat $O/big1.log; echo "== big off"; cat $O/big0.log; echo "== hard"; cat $O/hard1200.log; echo "== big on, xcd map off"; cat $O/big1_xcd0.log
