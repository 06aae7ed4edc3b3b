import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from openea_amd import ops
ops.lib()
rng = np.random.RandomState(0)
n, d, k = 100000, 100, 2000
x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
t = ops.to_table(x)
for mb in (1536, 3072, 6144, 12288):
    ops.topk_inner(t, t, d, k, ws_bytes=mb << 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); ops.topk_inner(t, t, d, k, ws_bytes=mb << 20); torch.cuda.synchronize()
    print("strip %5d MB: %.1f ms" % (mb, (time.perf_counter() - t0) * 1e3), flush=True)
