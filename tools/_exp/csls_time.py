"""alignment evaluation with and without CSLS at the 15K and 100K test-split sizes: ms per greedy_alignment call"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops  # noqa: E402
from openea_amd.modules.finding.alignment import greedy_alignment_device  # noqa: E402

ops.lib()
rng = np.random.RandomState(0)
sizes = ((10500, 100, 20),) if len(sys.argv) > 1 else ((10500, 75, 20), (10500, 100, 20), (70000, 100, 3))
for n, d, reps in sizes:
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = x + 0.3 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)
    t1, t2 = ops.to_table(x), ops.to_table(y)
    for csls in (0, 10):
        greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("eval %d x %d x %d csls=%d: %.3f ms (%.1f M pairs/s)" % (n, n, d, csls, ms, n / ms / 1e3), flush=True)
