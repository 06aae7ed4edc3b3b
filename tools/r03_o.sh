#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03o}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_reference_fullsize.py -m gpu -q -k "topk or csls or neighbour or knn or greedy or rank" > $OUT/pytest.log 2>&1
echo "rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest.log | tail -8
python tools/_exp/knn_time.py 2>&1 | grep kNN
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/knn_stats -- python $R/tools/_exp/knn_time.py > $OUT/knn_stats.log 2>&1
f=$(ls $OUT/knn_stats/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/knn_kernel_stats.csv && head -9 $f | cut -c1-150
rm -rf $OUT/knn_stats
cd $R
OEA_RANK_WGS=3072 python - <<PY 2>&1 | grep -v amdgpu
import time, numpy as np, torch
from openea_amd import ops
from openea_amd.modules.finding.alignment import greedy_alignment_device
ops.lib()
rng = np.random.RandomState(0)
res = []
for n, d, reps in ((10500, 75, 20), (70000, 100, 3)):
    e = rng.standard_normal((n, d)).astype(np.float32); e /= np.linalg.norm(e, axis=1, keepdims=True)
    t1 = ops.to_table(e); t2 = ops.to_table(e + 0.4 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d))
    for csls in (0, 10):
        greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls)
        torch.cuda.synchronize()
        res.append("%dx%d%s %.3f ms" % (n, d, "+csls" if csls else "", (time.perf_counter() - t0) / reps * 1e3))
print("; ".join(res))
PY
