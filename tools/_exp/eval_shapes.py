"""inner-product evaluation (plain / CSLS 10) at 70,000^2 x {100, 300, 1200}: product path times, records, identical results
against the fp32 sweep (once per d), for ablations by environment variable (OEA_XCD_MAP, OEA_BF16_BREG, ...).
  python tools/_exp/eval_shapes.py "100,300,1200" [noise] [check]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openea_amd import ops                                                            # noqa: E402
from openea_amd.modules.finding.alignment import greedy_alignment_device             # noqa: E402
from openea_amd.modules.finding.similarity import csls_means_device                  # noqa: E402

dims = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "100,300,1200").split(",")]
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
check = len(sys.argv) > 3 and sys.argv[3] == "check"
n = int(os.environ.get("N", "70000"))
tk = [1, 5, 10, 50]


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for d in dims:
    rng = np.random.RandomState(d)
    blocks = (500, 400, 300) if d == 1200 else (d,)
    b1s, b2s = [], []
    for db in blocks:
        b1 = rng.standard_normal((n, db)).astype(np.float32)
        b2 = (b1 + noise * rng.standard_normal((n, db)).astype(np.float32)).astype(np.float32)
        b1s.append(b1 / np.linalg.norm(b1, axis=1, keepdims=True))
        b2s.append(b2 / np.linalg.norm(b2, axis=1, keepdims=True))
    t1, t2 = ops.to_table(np.concatenate(b1s, 1)), ops.to_table(np.concatenate(b2s, 1))
    del b1s, b2s
    out = {}
    for csls in (0, 10):
        out[csls] = greedy_alignment_device(t1, t2, d, tk, "inner", False, csls)
        ms = wall(lambda: greedy_alignment_device(t1, t2, d, tk, "inner", False, csls), 3)
        fl = 2.0 * n * n * d * (2 if csls else 1)
        print("d=%4d noise=%.1f csls=%2d: %8.2f ms  %7.1f TF  hits %s" % (d, noise, csls, ms, fl / ms / 1e9, out[csls][2]), flush=True)
    st = {}
    t0 = time.perf_counter()
    m1 = ops.rank_eval_metrics_bf16(t1, t2, d, tk, stats=st)
    torch.cuda.synchronize()
    print("   plain: records %.2f per row, fallback %s" % (st["records"] / n, st["fallback"]))
    r, c = csls_means_device(t1, t2, d, "inner", 10)
    torch.cuda.synchronize()
    ms = wall(lambda: csls_means_device(t1, t2, d, "inner", 10), 2)
    print("   csls means alone: %.2f ms" % ms)
    st = {}
    ops.rank_eval_metrics_bf16(t1, t2, d, tk, stats=st, csls_r=r, csls_c=c)
    print("   csls rank sweep: records %.2f per row, fallback %s" % (st["records"] / n, st["fallback"]))
    if check:
        os.environ["OEA_EVAL_BF16"] = "0"
        os.environ["OEA_CSLS_BF16"] = "0"
        for csls in (0, 10):
            ref = greedy_alignment_device(t1, t2, d, tk, "inner", False, csls)
            print("   csls=%d identical to the fp32 sweep: ranks %s argmax %s" % (csls, torch.equal(ref[0], out[csls][0]), torch.equal(ref[1], out[csls][1])))
        os.environ.pop("OEA_EVAL_BF16")
        os.environ.pop("OEA_CSLS_BF16")
    del t1, t2
    torch.cuda.empty_cache()
