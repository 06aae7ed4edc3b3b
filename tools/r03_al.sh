#!/bin/bash
set -u
TAG=${1:-r03al}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/knn15.py <<PY
import sys, time, numpy as np, torch
sys.path.insert(0, "$R")
from openea_amd import ops
rng = np.random.RandomState(0)
n, d, k = 15000, 100, 1499
x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
t = ops.to_table(x)
for _ in range(3): ops.topk_inner(t, t, d, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): ops.topk_inner(t, t, d, k)
torch.cuda.synchronize()
print("kNN 15K: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
PY
for S in 0 1 2 3; do
  echo "STOP=$S $(OEA_TOPK_SELECT_STOP=$S python /tmp/knn15.py 2>&1 | grep kNN)" | tee -a $OUT/row_select_phases.txt
done
