#!/bin/bash
# round 5, call j: big pipeline with prefetched fragments (OEA_BF16_BIG_PF=1) against the first form (=0)
set -u
O=gpurun_out/r05j; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16_prefilter or csls_means_one_sweep or rank_eval_bit_exact or greedy_alignment_takes or csls_pipeline" 2>&1 | tail -6 ) > $O/pytest.log 2>&1
( OEA_BF16_BIG_PF=1 timeout 600 python tools/_exp/eval_shapes.py "300,1200" 8 check 2>&1 | grep -v amdgpu.ids ) > $O/pf1.log 2>&1
( OEA_BF16_BIG_PF=0 timeout 600 python tools/_exp/eval_shapes.py "300,1200" 8 2>&1 | grep -v amdgpu.ids ) > $O/pf0.log 2>&1
tools/prof.sh trace r05j -- python tools/_exp/eval1200_trace.py 1200 2
tail -3 $O/pytest.log; echo "== PF on"; cat $O/pf1.log; echo "== PF off"; cat $O/pf0.log; head -4 $O/trace_stats.csv | cut -c1-70,200-320
