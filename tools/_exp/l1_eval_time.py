"""RDGCN's evaluation (manhattan, d = 300, 70,000 pairs): all-pairs fp64 vs grid distances + exact doubts"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops  # noqa: E402

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (70000, 300)
rng = np.random.RandomState(0)
e1 = rng.standard_normal((n, d)).astype(np.float32)
e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
for label, e2 in (("trained-like", (e1 + 0.4 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)).astype(np.float32)),
                  ("random", rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d))):
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    res = {}
    for mode in ("f64", "grid"):
        os.environ["OEA_L1_EVAL"] = mode
        ops.rank_eval(t1, t2, d, "manhattan")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res[mode] = ops.rank_eval(t1, t2, d, "manhattan")
        torch.cuda.synchronize()
        print(f"{label} {n} x {n} x {d} {mode}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    same = torch.equal(res["f64"][0], res["grid"][0]) and torch.equal(res["f64"][1], res["grid"][1])
    print(f"   identical ranks and nearest candidates: {same}; mean rank {res['grid'][0].float().mean().item():.1f}", flush=True)
