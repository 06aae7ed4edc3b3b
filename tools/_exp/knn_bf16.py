"""neighbour search 100,000 x 100,000, k = 2,000 (and 40,100 / 800): wall time of oea_topk_inner; OEA_TOPK_BF16 is read once per
process, so run it twice.  Prints a checksum of the result so that the two runs can be compared."""
import os, sys, time, zlib
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops
ops.lib()
for n, d, k, trained in ((100000, 100, 2000, False), (100000, 100, 2000, True), (40100, 64, 800, False)):
    rng = np.random.RandomState(2)
    x = rng.standard_normal((n, d)).astype(np.float32)
    if trained:                                   # clustered rows, as trained tables have them
        cen = rng.standard_normal((n // 500 + 1, d)).astype(np.float32)
        x = cen[np.arange(n) // 500] + 0.6 * x
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    t = ops.to_table(x)
    out = ops.topk_inner(t, t, d, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        out = ops.topk_inner(t, t, d, k)
    e1.record()
    torch.cuda.synchronize()
    h = out.cpu().numpy()
    print("n=%d d=%d k=%d %s bf16=%s: %.3f ms  crc %08x" % (n, d, k, "clustered" if trained else "random", os.environ.get("OEA_TOPK_BF16", "1"),
                                                          e0.elapsed_time(e1) / reps, zlib.crc32(h.tobytes())), flush=True)
