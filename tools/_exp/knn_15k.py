"""neighbour search at the 15K shape (15,000 x 75, k = 1,499): random unit rows and the trained table of the 15K workload after STEPS steps;
OEA_TOPK_SYM_MIN decides between the N x N strip path (default below 32,768 rows) and the symmetric stream form."""
import os, sys, time, zlib
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from openea_amd import ops
from openea_amd.models.trainer import refresh_neighbours
ops.lib()
dev = torch.device("cuda", 0)
n, d, k = 15000, 75, 1499
rng = np.random.RandomState(2)
x = rng.standard_normal((n, d)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
t = ops.to_table(x)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
ms, out = timed(lambda: ops.topk_inner(t, t, d, k))
print("random 15,000 x 75 k=1,499 SYM_MIN=%s: %.3f ms crc %08x" % (os.environ.get("OEA_TOPK_SYM_MIN", "32768"), ms, zlib.crc32(out.cpu().numpy().tobytes())), flush=True)
steps = int(os.environ.get("STEPS", "2000"))
wl = bench.Workload(torch, ops, "EN-FR-15K-V1", 75, 5000, 10, 0.9, dev)
wl.epochs.run_steps(wl.trainer, steps)
torch.cuda.synchronize()
ids = wl.kgs.kg1.entities_list
ms, _ = timed(lambda: refresh_neighbours(wl.ent, ids, wl.k1), 10)
nb = wl.trainer.neighbours if hasattr(wl.trainer, "neighbours") else None
print("trained (%d steps) refresh_neighbours of KG1 (%d rows, k=%d): %.3f ms" % (steps, len(ids), wl.k1, ms), flush=True)
