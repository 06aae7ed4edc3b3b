"""certified bf16 prefilter of the inner-product evaluation: error of the approximate similarities, exactness of rank / argmax
against the fp32 sweep, timings (tools/r04/h.sh)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops  # noqa: E402

ops.lib()
rng = np.random.RandomState(0)


def unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


# ---- 1. error of the approximate values against the exact k-ordered chain ------------------------------------------------
for n1, n2, d in ((1000, 3000, 75), (700, 1300, 100), (513, 900, 300), (300, 500, 37)):
    a, b = unit(rng.standard_normal((n1, d))), unit(rng.standard_normal((n2, d)))
    ta, tb = ops.to_table(a), ops.to_table(b)
    exact = ops.sim_matrix(ta, tb, d, "inner")
    approx = ops.sim_bf16_matrix(ta, tb, d)
    err = float((exact - approx).abs().max())
    kp16 = (d + 15) // 16 * 16
    eps = 1.02 * (3.02 * 2.0 ** -18 + (3 * kp16 + d + 8) * 2.0 ** -23)
    print("approx vs exact %5d x %5d x %3d: max |err| %.3e  (bound %.3e)  %s" % (n1, n2, d, err, eps, "OK" if err <= eps else "VIOLATED"), flush=True)

# ---- 2. rank / argmax identical with the fp32 sweep ---------------------------------------------------------------------------
def check(name, e1, e2, d, off=0):
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    r0, a0 = ops.rank_eval(t1, t2, d, "inner", gold_offset=off)
    st = {}
    r1, a1 = ops.rank_eval_bf16(t1, t2, d, gold_offset=off, stats=st)
    same = bool(torch.equal(r0, r1) and torch.equal(a0, a1))
    print("%-28s %5d x %5d x %3d: identical %s, records %d (%.1f per row), fallback %s, hits@1 %d" %
          (name, len(e1), len(e2), d, same, st["records"], st["records"] / len(e1), st["fallback"], int((r0 == 0).sum())), flush=True)
    if not same:
        bad = torch.nonzero((r0 != r1) | (a0 != a1)).reshape(-1)[:5].cpu().numpy()
        print("   first differing rows", bad, r0[bad].cpu().numpy(), r1[bad].cpu().numpy(), a0[bad].cpu().numpy(), a1[bad].cpu().numpy())
    return same


ok = True
for d in (75, 100, 300, 37):
    n1, n2, off = 1500, 4000, 700
    e2 = unit(rng.standard_normal((n2, d)))
    ok &= check("random gold", unit(rng.standard_normal((n1, d))), e2, d, off)
    ok &= check("trained (gold near top)", unit(e2[off:off + n1] + 0.5 * rng.standard_normal((n1, d)) / np.sqrt(d)), e2, d, off)
    e2t = e2.copy()
    e2t[3000:3400] = e2t[off:off + 400]
    e2t[0:300] = e2t[off + 500:off + 800]
    ok &= check("duplicates of gold columns", unit(e2t[off:off + n1] + 0.1 * rng.standard_normal((n1, d)) / np.sqrt(d)), e2t, d, off)
    e2c = e2.copy()
    e2c[1000:3000] = unit(e2c[1000] + 2e-4 * rng.standard_normal((2000, d)))
    ok &= check("clustered", unit(e2c[off:off + n1] + 1e-4 * rng.standard_normal((n1, d))), e2c, d, off)
big = unit(rng.standard_normal((3, 60))) * np.array([[5.0], [0.01], [1.0]], np.float32)
e2 = (unit(rng.standard_normal((2500, 60))) * rng.uniform(0.05, 4.0, (2500, 1))).astype(np.float32)
ok &= check("unnormalised rows", (e2[:2000] + 0.3 * rng.standard_normal((2000, 60)).astype(np.float32) / 8).astype(np.float32), e2, 60, 0)
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)

# ---- 3. timings ------------------------------------------------------------------------------------------------------------------
def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


TOPK = [1, 5, 10, 50]
for n, d, reps in ((10500, 75, 30), (70000, 100, 4), (70000, 300, 3)):
    for mode in ("gold near the top", "random gold"):
        e1 = unit(rng.standard_normal((n, d)))
        e2 = unit(e1 + 0.4 * rng.standard_normal((n, d)) / np.sqrt(d)) if mode.startswith("gold") else unit(rng.standard_normal((n, d)))
        t1, t2 = ops.to_table(e1), ops.to_table(e2)
        st = {}
        ms32 = wall(lambda: ops.rank_eval_metrics(t1, t2, d, TOPK), reps)
        ms16 = wall(lambda: ops.rank_eval_metrics_bf16(t1, t2, d, TOPK, stats=st), reps)
        a = ops.rank_eval_metrics(t1, t2, d, TOPK)
        b = ops.rank_eval_metrics_bf16(t1, t2, d, TOPK)
        same = (b is not None and bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])) and a[2:4] == b[2:4]
                and abs(a[4] - b[4]) <= 1e-9 * abs(a[4]))            # (the reciprocal-rank sum is a double reduced in another order)
        print("eval %6d^2 x %3d, %-17s: fp32 %.3f ms, bf16 prefilter %.3f ms (%.2fx), records %d (%.1f per row), fallback %s, identical incl. metrics %s" %
              (n, d, mode, ms32, ms16, ms32 / ms16, st["records"], st["records"] / n, st["fallback"], same), flush=True)

# ---- 4. CSLS: means (fp32 one-sweep) + rank sweep with the means, fp32 vs bf16 prefilter ---------------------------------------
from openea_amd.modules.finding.similarity import csls_means_device  # noqa: E402
for n, d, reps in ((10500, 75, 20), (70000, 100, 3)):
    e1 = unit(rng.standard_normal((n, d)))
    e2 = unit(e1 + 0.4 * rng.standard_normal((n, d)) / np.sqrt(d))
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    os.environ["OEA_CSLS_BF16"] = "0"
    r, c = csls_means_device(t1, t2, d, "inner", 10)
    ms_means = wall(lambda: csls_means_device(t1, t2, d, "inner", 10), reps)
    os.environ["OEA_CSLS_BF16"] = "1"
    os.environ["OEA_CSLS_BF16_MIN_PAIRS"] = "1"
    rb, cb = csls_means_device(t1, t2, d, "inner", 10)
    ms_means16 = wall(lambda: csls_means_device(t1, t2, d, "inner", 10), reps)
    del os.environ["OEA_CSLS_BF16_MIN_PAIRS"]
    print("CSLS %6d^2 x %3d: means fp32 sweep %.3f ms, bf16 sweep %.3f ms (%.2fx), identical %s" %
          (n, d, ms_means, ms_means16, ms_means / ms_means16, bool(torch.equal(r, rb) and torch.equal(c, cb))), flush=True)
    st = {}
    ms32 = wall(lambda: ops.rank_eval_metrics(t1, t2, d, TOPK, r, c), reps)
    ms16 = wall(lambda: ops.rank_eval_metrics_bf16(t1, t2, d, TOPK, stats=st, csls_r=r, csls_c=c), reps)
    a = ops.rank_eval_metrics(t1, t2, d, TOPK, r, c)
    b = ops.rank_eval_metrics_bf16(t1, t2, d, TOPK, csls_r=r, csls_c=c)
    same = b is not None and bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])) and a[2:4] == b[2:4]
    print("CSLS %6d^2 x %3d: means %.3f ms; rank sweep fp32 %.3f ms, bf16 %.3f ms (%.2fx); records %d, identical %s" %
          (n, d, ms_means, ms32, ms16, ms32 / ms16, st["records"], same), flush=True)
