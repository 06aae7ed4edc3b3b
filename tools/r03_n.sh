#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gnn_gpu.py -m gpu -q -k "dropout or hard_negative" 2>&1 | tail -3
for W in 512 1024 1536 2048 3072 4096 8192; do
OEA_RANK_WGS=$W python - <<PY 2>&1 | grep -v amdgpu
import time, numpy as np, torch
from openea_amd import ops
from openea_amd.modules.finding.alignment import greedy_alignment_device
ops.lib()
rng = np.random.RandomState(0)
res = []
for n, d, reps in ((10500, 75, 20), (10500, 100, 20), (70000, 100, 3)):
    e = rng.standard_normal((n, d)).astype(np.float32); e /= np.linalg.norm(e, axis=1, keepdims=True)
    t1 = ops.to_table(e); t2 = ops.to_table(e + 0.4 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d))
    for csls in (0, 10):
        greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls)
        torch.cuda.synchronize()
        res.append("%dx%d%s %.3f ms" % (n, d, "+csls" if csls else "", (time.perf_counter() - t0) / reps * 1e3))
print("WGS=$W: " + "; ".join(res))
PY
done
