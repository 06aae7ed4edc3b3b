#!/usr/bin/env python
"""Self-check ONLY: writes a file with the layout of tf1.npz whose "TensorFlow outputs" come from THIS REPO'S ORACLE, so that
the code of tests/test_tf1_golden.py (CPU and GPU halves) can be exercised where no TensorFlow exists:

    python tests/golden/make_tf1_standin.py /tmp/tf1_standin.npz
    OEA_TF1_GOLDEN=/tmp/tf1_standin.npz python -m pytest tests/test_tf1_golden.py

It proves nothing about TensorFlow (the oracle is compared with itself / the device with the oracle); the real file is
written by make_tf1_golden.py.  Never commit its output as tests/golden/tf1.npz."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
spec = importlib.util.spec_from_file_location("tf1_tests", os.path.join(os.path.dirname(HERE), "test_tf1_golden.py"))
t = importlib.util.module_from_spec(spec)
spec.loader.exec_module(t)

rng = np.random.RandomState(20260925)                       # the same inputs as make_tf1_golden.py
N_ENT, N_REL, D, B, STEPS, LR = 12, 3, 5, 9, 3, 0.05
ent0 = (rng.standard_normal((N_ENT, D)) * 0.6).astype(np.float32)
rel0 = (rng.standard_normal((N_REL, D)) * 0.6).astype(np.float32)
batches = np.stack([np.stack([rng.randint(0, 6, B), rng.randint(0, N_REL, B), rng.randint(0, 6, B)], 1) for _ in range(STEPS)]).astype(np.int32)
out = dict(tf_version=np.array("STAND-IN (oracle)"), opt_ent0=ent0, opt_rel0=rel0, opt_batches=batches, opt_lr=np.float32(LR))
for norm in (1, 0):
    for name in t.OPTS:
        e, r, l = t.oracle_optimiser_run(out, name, norm)
        out["opt_%s_norm%d_ent" % (name, norm)], out["opt_%s_norm%d_rel" % (name, norm)], out["opt_%s_norm%d_loss" % (name, norm)] = e, r, l
rows = np.array([0, 0, 0, 1, 1, 2, 3, 3, 3, 3], np.int64)
cols = np.array([0, 2, 3, 1, 2, 0, 0, 1, 2, 3], np.int64)
vals = rng.standard_normal(len(rows)).astype(np.float32)
order_cm = np.lexsort((rows, cols))
cases = {"rowmajor": (rows, cols, vals), "colmajor": (rows[order_cm], cols[order_cm], vals[order_cm]),
         "dup": (np.append(rows, 1), np.append(cols, 2), np.append(vals, np.float32(0.7)))}
for name, (rr, cc, vv) in cases.items():
    _, by_runs = t.softmax_groupings(rr, vv)                  # the stand-in ASSUMES the 'runs' grouping
    out["ssm_%s_rows" % name], out["ssm_%s_cols" % name], out["ssm_%s_logits" % name] = rr, cc, vv
    out["ssm_%s_out_indices" % name], out["ssm_%s_out_values" % name] = np.stack([rr, cc], 1), by_runs.astype(np.float32)
x = (rng.standard_normal((7, 4)) * 2 + 1).astype(np.float32)
out["bn_x"], out["bn_y"] = x, (x / np.sqrt(1 + 1e-3)).astype(np.float32)
perm = rng.permutation(len(rows))
rr, cc, vv = np.append(rows[perm], 2), np.append(cols[perm], 0), np.append(vals[perm], np.float32(-1.25))
xd = rng.standard_normal((4, 3)).astype(np.float32)
import scipy.sparse as sp            # noqa: E402
out["spmm_y"] = (sp.coo_matrix((vv, (rr, cc)), shape=(4, 4)).tocsr() @ xd).astype(np.float32)
out.update(spmm_rows=rr, spmm_cols=cc, spmm_vals=vv, spmm_x=xd)
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/tf1_standin.npz"
np.savez(path, **out)
print("wrote", path)
