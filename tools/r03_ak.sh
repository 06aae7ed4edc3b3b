#!/bin/bash
set -u
TAG=${1:-r03ak}
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
for _ in range(20): ops.topk_inner(t, t, d, k)
torch.cuda.synchronize()
print("kNN 15K: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
PY
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python /tmp/knn15.py > $OUT/log.txt 2>&1
grep "kNN 15K" $OUT/log.txt
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/knn15_kernel_stats.csv
rm -rf $OUT/tr
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/knn15_kernel_stats.csv")))[:8]:
    print("%-80s %5s avg %8.1f us total %8.2f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
PY
