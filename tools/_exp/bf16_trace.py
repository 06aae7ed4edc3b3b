"""kernel trace target: the bf16-prefilter evaluation, one size per run: python bf16_trace.py near|random n d reps"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops  # noqa: E402

ops.lib()
rng = np.random.RandomState(0)
mode, n, d, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
e1 = rng.standard_normal((n, d)).astype(np.float32)
e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
e2 = e1 + 0.4 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d) if mode == "near" else rng.standard_normal((n, d)).astype(np.float32)
e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
t1, t2 = ops.to_table(e1), ops.to_table(e2.astype(np.float32))
for _ in range(reps):
    st = {}
    ops.rank_eval_metrics_bf16(t1, t2, d, [1, 5, 10, 50], stats=st)
torch.cuda.synchronize()
print(n, d, mode, st, flush=True)
