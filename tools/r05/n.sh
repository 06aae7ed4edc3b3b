#!/bin/bash
# round 5, call n: the evaluation tests on the last build (status-word fallback instead of OEA_REQUIRE), incl. OEA_RANK_WGS raised
set -u
O=gpurun_out/r05n; mkdir -p $O
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16_prefilter or greedy_alignment_takes or rank_eval_metrics" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
( OEA_RANK_WGS=200000 OEA_EVAL_BF16_MIN_PAIRS=1 timeout 200 python - <<'PY' 2>&1 | tail -5
import numpy as np, torch, sys
sys.path.insert(0, ".")
from openea_amd import ops
rng = np.random.RandomState(0)
e1 = rng.standard_normal((3000, 64)).astype(np.float32); e2 = (e1 + 0.5 * rng.standard_normal((3000, 64))).astype(np.float32)
t1, t2 = ops.to_table(e1), ops.to_table(e2)
st = {}
r1, a1 = ops.rank_eval_bf16(t1, t2, 64, stats=st)          # oversized launch: must fall back, not raise
r0, a0 = ops.rank_eval(t1, t2, 64, "inner", allow_bf16=False)
print("fallback", st["fallback"], "same", bool(torch.equal(r0, r1) and torch.equal(a0, a1)))
m = ops.rank_eval_metrics_bf16(t1, t2, 64, [1, 5, 10, 50])
print("metrics entry returns None:", m is None)
PY
) > $O/wgs.log 2>&1
cat $O/pytest.log $O/wgs.log
