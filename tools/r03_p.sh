#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for M in 32768 8192; do
OEA_TOPK_SYM_MIN=$M python - <<PY 2>&1 | grep -v amdgpu
import time, numpy as np, torch
from openea_amd import ops
from oracle import cport
ops.lib()
rng = np.random.RandomState(0)
for n, d, k, reps in ((15000, 100, 1499, 20), (15000, 75, 1499, 20), (30000, 100, 600, 10)):
    x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    t = ops.to_table(x)
    out = ops.topk_inner(t, t, d, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = ops.topk_inner(t, t, d, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    rows = rng.choice(n, 20, replace=False)
    ok = np.array_equal(out.cpu().numpy()[rows], cport.topk_inner(x[rows], x, k))
    print("SYM_MIN=$M: kNN %d x %d, k=%d: %.3f ms, equals the oracle on 20 rows: %s" % (n, d, k, ms, ok))
PY
done
