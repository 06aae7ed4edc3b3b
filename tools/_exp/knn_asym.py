"""neighbour search with queries != candidates (BootEA's bootstrapping candidates, AliNet's negative neighbours, a rank's row block of a
sharded refresh): ms per call, bf16-split sweep (default) against OEA_TOPK_BF16=0 in a second process"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops
ops.lib()
rng = np.random.RandomState(0)
for nq, nc, d, k in ((70000, 70000, 100, 10), (50000, 100000, 100, 2000), (70000, 70000, 300, 10)):
    x = rng.standard_normal((nq, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rng.standard_normal((nc, d)).astype(np.float32); y /= np.linalg.norm(y, axis=1, keepdims=True)
    tq, tc = ops.to_table(x), ops.to_table(y)
    out = ops.topk_inner(tq, tc, d, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = ops.topk_inner(tq, tc, d, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    chk = int(out.long().sum().item())
    print("kNN %d x %d x %d, k=%d: %.2f ms (bf16=%s) checksum %d" % (nq, nc, d, k, ms, os.environ.get("OEA_TOPK_BF16", "1"), chk), flush=True)
