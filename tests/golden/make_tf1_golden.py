#!/usr/bin/env python
"""TensorFlow-1 golden vectors for the TF-defined pieces of the hot path -> tests/golden/tf1.npz.

The reference's device arithmetic is TensorFlow 1.x (README: tested on 1.8 / 1.12); neither the build container nor the GPU
box has TensorFlow, so the oracle RESTATES these ops from their documentation.  This script is the five-minute way to close
that gap for anyone with a TensorFlow wheel: run it (pure TensorFlow + numpy, nothing from this repo or from the reference
is imported), commit the tf1.npz it writes, and tests/test_tf1_golden.py holds the oracle (CPU) and the HIP path (GPU)
to it.  Without the file those tests SKIP with a message that says so.

    python tests/golden/make_tf1_golden.py            # TF 1.x, or TF 2.x through tensorflow.compat.v1 (same kernels)

What it pins, with the reference call sites:
  opt_*      three steps of tf.train.{GradientDescent,Adagrad,Adam,Adadelta}Optimizer on a TransE-style loss whose lookups go
             through tf.nn.l2_normalize(var, 1) with DUPLICATE ids (modules/base/optimizers.py:4-20,
             modules/base/initializers.py:26, models/basic_model.py:80-98) -- and the same with the normalisation off
             (IndexedSlices gradients: the sparse apply kernels);
  ssm_*      tf.sparse_softmax on a SparseTensor whose indices are in COLUMN-major order (AliNet feeds the coo matrix's
             (row, col) as scipy stores it: approaches/alinet.py:661-676), on a row-major one, and on one with a duplicated
             index -- which entries does it normalise together?  ('runs' vs 'row')
  bn_*       tf.keras.layers.BatchNormalization()(x) called WITHOUT `training=` inside a TF1 graph
             (approaches/alinet.py:575,614,657): inference affine x / sqrt(1 + 1e-3), or batch statistics?  
  spmm_*     tf.sparse_tensor_dense_matmul on UNSORTED COO indices with a duplicated entry
             (approaches/gcn_align.py:79-86) -- duplicates summed?  
"""
import os
import sys

import numpy as np

try:
    import tensorflow.compat.v1 as tf          # TF 2.x (and late 1.x)
    tf.disable_v2_behavior()
except Exception:                              # noqa: BLE001 -- early TF 1.x
    import tensorflow as tf

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf1.npz")
out = {"tf_version": np.array(tf.__version__)}
rng = np.random.RandomState(20260925)

# ---- optimisers through l2_normalize + gather with duplicate ids -----------------------------------------------------------
N_ENT, N_REL, D, B, STEPS, LR = 12, 3, 5, 9, 3, 0.05
ent0 = (rng.standard_normal((N_ENT, D)) * 0.6).astype(np.float32)
rel0 = (rng.standard_normal((N_REL, D)) * 0.6).astype(np.float32)
batches = np.stack([np.stack([rng.randint(0, 6, B), rng.randint(0, N_REL, B), rng.randint(0, 6, B)], 1)       # ids < 6: many duplicates,
                    for _ in range(STEPS)]).astype(np.int32)                                                    # rows 6.. never gathered
out.update(opt_ent0=ent0, opt_rel0=rel0, opt_batches=batches, opt_lr=np.float32(LR))
OPTS = {"SGD": lambda: tf.train.GradientDescentOptimizer(LR), "Adagrad": lambda: tf.train.AdagradOptimizer(LR),
        "Adam": lambda: tf.train.AdamOptimizer(LR), "Adadelta": lambda: tf.train.AdadeltaOptimizer(LR)}
for norm in (1, 0):
    for name, make in OPTS.items():
        tf.reset_default_graph()
        ent, rel = tf.Variable(ent0), tf.Variable(rel0)
        te = tf.nn.l2_normalize(ent, 1) if norm else ent
        tr = tf.nn.l2_normalize(rel, 1) if norm else rel
        ph = tf.placeholder(tf.int32, [None, 3])
        h = tf.nn.embedding_lookup(te, ph[:, 0])
        r = tf.nn.embedding_lookup(tr, ph[:, 1])
        t = tf.nn.embedding_lookup(te, ph[:, 2])
        loss = tf.reduce_sum(tf.reduce_sum(tf.square(h + r - t), 1))            # losses.py:38 (positive_loss, L2)
        step = make().minimize(loss)
        with tf.Session() as s:
            s.run(tf.global_variables_initializer())
            e_hist, r_hist, l_hist = [], [], []
            for b in batches:
                l, _ = s.run([loss, step], {ph: b})
                e, rr = s.run([ent, rel])
                e_hist.append(e), r_hist.append(rr), l_hist.append(l)
        out["opt_%s_norm%d_ent" % (name, norm)] = np.stack(e_hist)
        out["opt_%s_norm%d_rel" % (name, norm)] = np.stack(r_hist)
        out["opt_%s_norm%d_loss" % (name, norm)] = np.asarray(l_hist, np.float32)

# ---- tf.sparse_softmax on non-canonical index orders -----------------------------------------------------------------------
rows = np.array([0, 0, 0, 1, 1, 2, 3, 3, 3, 3], np.int64)
cols = np.array([0, 2, 3, 1, 2, 0, 0, 1, 2, 3], np.int64)
vals = rng.standard_normal(len(rows)).astype(np.float32)
order_cm = np.lexsort((rows, cols))                   # column-major: sorted by (col, row) -- scipy's coo of AliNet's adjacency
cases = {"rowmajor": (rows, cols, vals), "colmajor": (rows[order_cm], cols[order_cm], vals[order_cm]),
         "dup": (np.append(rows, 1), np.append(cols, 2), np.append(vals, np.float32(0.7)))}           # (1, 2) twice
for name, (rr, cc, vv) in cases.items():
    tf.reset_default_graph()
    sp = tf.SparseTensor(np.stack([rr, cc], 1), vv, [4, 4])
    with tf.Session() as s:
        res = s.run(tf.sparse_softmax(sp))
    out["ssm_%s_rows" % name], out["ssm_%s_cols" % name], out["ssm_%s_logits" % name] = rr, cc, vv
    out["ssm_%s_out_indices" % name], out["ssm_%s_out_values" % name] = res.indices, res.values

# ---- keras BatchNormalization without `training=` in a TF1 graph --------------------------------------------------------------
tf.reset_default_graph()
x = (rng.standard_normal((7, 4)) * 2 + 1).astype(np.float32)
y = tf.keras.layers.BatchNormalization()(tf.constant(x))
with tf.Session() as s:
    s.run(tf.global_variables_initializer())
    out["bn_x"], out["bn_y"] = x, s.run(y)

# ---- sparse_tensor_dense_matmul on unsorted COO with a duplicate -------------------------------------------------------------
tf.reset_default_graph()
perm = rng.permutation(len(rows))
rr, cc, vv = np.append(rows[perm], 2), np.append(cols[perm], 0), np.append(vals[perm], np.float32(-1.25))     # (2, 0) twice
xd = rng.standard_normal((4, 3)).astype(np.float32)
with tf.Session() as s:
    out["spmm_y"] = s.run(tf.sparse_tensor_dense_matmul(tf.SparseTensor(np.stack([rr, cc], 1), vv, [4, 4]), tf.constant(xd)))
out["spmm_rows"], out["spmm_cols"], out["spmm_vals"], out["spmm_x"] = rr, cc, vv, xd

np.savez(OUT, **out)
print("wrote %s (TensorFlow %s): %d arrays" % (OUT, tf.__version__, len(out)))
sys.exit(0)
