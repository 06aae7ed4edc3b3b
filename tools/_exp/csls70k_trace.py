"""one greedy_alignment with CSLS at 70,000^2 x 100 (after a warm-up call) -- for a kernel trace"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops  # noqa: E402
from openea_amd.modules.finding.alignment import greedy_alignment_device  # noqa: E402

rng = np.random.RandomState(0)
n, d = 70000, 100
x = rng.standard_normal((n, d)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
y = x + 0.3 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)
t1, t2 = ops.to_table(x), ops.to_table(y)
for _ in range(3):
    greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, 10)
torch.cuda.synchronize()
