"""whole greedy_alignment call at 10,500^2 x 75, CSLS k = 10 (the 15K datasets' evaluation) for a kernel trace"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops
from openea_amd.modules.finding.alignment import greedy_alignment_device
ops.lib()
rng = np.random.RandomState(0)
n, d = 10500, 75
e1 = rng.standard_normal((n, d)).astype(np.float32); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
e2 = (e1 + 0.4 * rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32); e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
t1, t2 = ops.to_table(e1), ops.to_table(e2)
for csls in (10, 0):
    greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls)
    torch.cuda.synchronize()
    e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, csls)
    e1_.record()
    torch.cuda.synchronize()
    print("greedy_alignment 10,500^2 x 75 csls=%d: %.3f ms" % (csls, e0.elapsed_time(e1_) / 20), flush=True)
