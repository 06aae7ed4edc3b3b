"""neighbour search at the BootEA 100K refresh size (100,000 x 100,000 x 100, k = 2,000) and at 15K: ms per call.
OEA_TOPK_LISTS=0 selects the strip path."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops  # noqa: E402

ops.lib()
rng = np.random.RandomState(0)
cases = ((100000, 100, 2000, 2),) if os.environ.get("KNN_QUICK") else ((100000, 100, 2000, 5), (15000, 100, 1499, 20))
for n, d, k, reps in cases:
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    t = ops.to_table(x)
    ops.topk_inner(t, t, d, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ops.topk_inner(t, t, d, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("kNN %d x %d x %d, k=%d: %.2f ms  (%.1f TFLOP/s, lists=%s)" % (n, n, d, k, ms, 2.0 * n * n * d / ms / 1e9,
                                                                      os.environ.get("OEA_TOPK_LISTS", "1")), flush=True)
