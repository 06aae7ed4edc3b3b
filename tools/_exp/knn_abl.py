"""ablation of the bf16 symmetric sweep (OEA_TOPK_EXP mask compiled in for the experiment only): wall time of oea_topk_inner at
100,000 x 100, k = 2,000 (results are garbage when the mask is set)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops
ops.lib()
n, d, k = 100000, 100, 2000
rng = np.random.RandomState(2)
x = rng.standard_normal((n, d)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
t = ops.to_table(x)
for _ in range(2):
    out = ops.topk_inner(t, t, d, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = ops.topk_inner(t, t, d, k)
e1.record()
torch.cuda.synchronize()
print("exp=%s select_stop=%s: %.3f ms" % (os.environ.get("OEA_TOPK_EXP", "0"), os.environ.get("OEA_TOPK_SELECT_STOP", "0"), e0.elapsed_time(e1) / 5), flush=True)
