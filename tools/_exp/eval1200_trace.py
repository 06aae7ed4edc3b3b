"""a few launches of the 70,000^2 x 1,200 evaluation (plain + CSLS 10) for rocprofv3 (kernel trace / counter passes)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openea_amd import ops                                                            # noqa: E402
from openea_amd.modules.finding.alignment import greedy_alignment_device             # noqa: E402

n, d = int(os.environ.get("N", "70000")), int(sys.argv[1]) if len(sys.argv) > 1 else 1200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.RandomState(0)
blocks = (500, 400, 300) if d == 1200 else (d,)
b1s, b2s = [], []
for db in blocks:
    b1 = rng.standard_normal((n, db)).astype(np.float32)
    b2 = (b1 + 8.0 * rng.standard_normal((n, db)).astype(np.float32)).astype(np.float32)
    b1s.append(b1 / np.linalg.norm(b1, axis=1, keepdims=True))
    b2s.append(b2 / np.linalg.norm(b2, axis=1, keepdims=True))
t1, t2 = ops.to_table(np.concatenate(b1s, 1)), ops.to_table(np.concatenate(b2s, 1))
for _ in range(reps):
    greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, 0)
    greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, 10)
torch.cuda.synchronize()
