"""A/B of the two-stream strip pipeline of oea_topk_inner (run once per setting: OEA_KNN_OVERLAP=0|1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops
ops.lib()
rng = np.random.RandomState(0)
for n, d, k in ((100000, 100, 2000), (100000, 75, 2000), (30000, 100, 1499)):
    x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    t = ops.to_table(x)
    ref = None
    for mb in (1024, 2048, 4096, 8192):
        out = ops.topk_inner(t, t, d, k, ws_bytes=mb << 20); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); out = ops.topk_inner(t, t, d, k, ws_bytes=mb << 20); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        srt = torch.sort(out[:2000], dim=1).values
        same = True if ref is None else bool(torch.equal(srt, ref))
        ref = srt if ref is None else ref
        print("overlap=%s n=%d d=%d k=%d ws %5d MB: %.1f ms  same_sets=%s" % (os.environ.get("OEA_KNN_OVERLAP", "1"), n, d, k, mb, min(ts), same), flush=True)
