"""AliNet's evaluation shape on the GPU: 70,000^2 x 1,200 (three normalised blocks), inner, plain and CSLS 10 -- times of the
product path (certified bf16 prefilter) and of the fp32 sweep, records per row, fallback, identical results, oracle spot rows.
  python tools/_exp/alinet_eval.py [n] [noise]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openea_amd import ops                                                            # noqa: E402
from openea_amd.modules.finding.alignment import greedy_alignment_device             # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
dims = (500, 400, 300)
d = sum(dims)
rng = np.random.RandomState(0)
b1s, b2s = [], []
for db in dims:
    b1 = rng.standard_normal((n, db)).astype(np.float32)
    b2 = (b1 + noise * rng.standard_normal((n, db)).astype(np.float32)).astype(np.float32)
    b1s.append(b1 / np.linalg.norm(b1, axis=1, keepdims=True))
    b2s.append(b2 / np.linalg.norm(b2, axis=1, keepdims=True))
e1, e2 = np.concatenate(b1s, 1), np.concatenate(b2s, 1)
t1, t2 = ops.to_table(e1), ops.to_table(e2)
tk = [1, 5, 10, 50]


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {}
for bf in ("1", "0"):
    os.environ["OEA_EVAL_BF16"] = bf
    os.environ["OEA_CSLS_BF16"] = bf
    for csls in (0, 10):
        out = greedy_alignment_device(t1, t2, d, tk, "inner", False, csls)
        ms = wall(lambda: greedy_alignment_device(t1, t2, d, tk, "inner", False, csls), 2)
        res[(bf, csls)] = out
        fl = 2.0 * n * n * d * (2 if csls else 1)
        print("bf16=%s csls=%2d: %8.2f ms  %7.1f TF  hits %s" % (bf, csls, ms, fl / ms / 1e9, out[2]), flush=True)
for csls in (0, 10):
    a, b = res[("1", csls)], res[("0", csls)]
    print("csls=%d identical ranks %s argmax %s" % (csls, torch.equal(a[0], b[0]), torch.equal(a[1], b[1])))
os.environ["OEA_EVAL_BF16"] = "1"
st = {}
ops.rank_eval_metrics_bf16(t1, t2, d, tk, stats=st)
print("plain: records %d (%.2f per row) fallback %s" % (st["records"], st["records"] / n, st["fallback"]))
from openea_amd.modules.finding.similarity import csls_means_device                  # noqa: E402
r, c = csls_means_device(t1, t2, d, "inner", 10)
st = {}
ops.rank_eval_metrics_bf16(t1, t2, d, tk, stats=st, csls_r=r, csls_c=c)
print("csls: records %d (%.2f per row) fallback %s" % (st["records"], st["records"] / n, st["fallback"]))
# random golds (rank in the bulk: the band around the gold is wide)
perm = torch.randperm(n, device=t2.device)
t2p = t2[perm].contiguous()
st = {}
t0 = time.perf_counter()
m1 = ops.rank_eval_metrics_bf16(t1, t2p, d, tk, stats=st)
torch.cuda.synchronize()
print("random golds: records %d (%.2f per row) fallback %s, %.1f ms" % (st["records"], st["records"] / n, st["fallback"], (time.perf_counter() - t0) * 1e3))
if m1 is not None:
    m0 = ops.rank_eval_metrics(t1, t2p, d, tk)
    print("random golds identical:", torch.equal(m0[0], m1[0]), torch.equal(m0[1], m1[1]))
# oracle spot rows
from oracle import cport                                                              # noqa: E402
rows = rng.choice(n, 48, replace=False)
s = cport.sim_matrix(e1[rows], e2, "inner")
g = s[np.arange(len(rows)), rows]
ref = ((s > g[:, None]) | ((s == g[:, None]) & (np.arange(n)[None, :] < rows[:, None]))).sum(1)
rk = res[("1", 0)][0].cpu().numpy()
am = res[("1", 0)][1].cpu().numpy()
print("oracle spot rows: ranks", np.array_equal(rk[rows], ref), "argmax", np.array_equal(am[rows], s.argmax(1)))
err = float((ops.sim_matrix(t1[:2048], t2[:4096], d, "inner") - ops.sim_bf16_matrix(t1[:2048], t2[:4096], d)).abs().max())
print("max |bf16 - fp32| on 2048 x 4096: %.3e (bound eps*3 = %.3e)" % (err, 3 * 1.02 * (3.02 * 2.0 ** -18 + (3 * 128 + 10) * 2.0 ** -23 + 1208 * 2.0 ** -24)))
