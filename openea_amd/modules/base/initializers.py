"""Embedding initialisers (mirror of openea/modules/base/initializers.py).

The reference returns a TF tensor that is ``tf.nn.l2_normalize(variable, 1)`` when
``is_l2_norm`` (initializers.py:26,34,41,50); here the returned ``EmbeddingTable`` keeps the
raw variable on the device plus that flag, and every consumer (lookup, the fused step)
applies the normalisation on the fly.  Initial values are drawn on the host (one-off) with a
seeded numpy RNG.
"""
import math

import numpy as np

_rng = np.random.RandomState(20190719)


def seed(s):
    global _rng
    _rng = np.random.RandomState(s)


def truncated_normal_host(rng, shape, stddev):
    """tf.initializers.truncated_normal: N(0, stddev), values beyond 2 sigma re-drawn."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def xavier_host(rng, shape):
    """tf.contrib.layers.xavier_initializer(uniform=False): truncated normal, var = 2/(fan_in+fan_out)
    (initializers.py:22-26)."""
    std = math.sqrt(2.0 / (shape[0] + shape[1]))
    return truncated_normal_host(rng, shape, std / 0.87962566103423978)   # TF rescales the truncated std


def unit_host(rng, shape):
    """initializers.py:44-50: gauss(0,1) rows, sklearn-normalised."""
    v = rng.standard_normal(shape)
    n = np.linalg.norm(v, axis=1, keepdims=True)
    n[n == 0] = 1.0
    return (v / n).astype(np.float32)


def orthogonal_host(rng, shape):
    """tf.initializers.orthogonal (initializers.py:53-56): QR of a normal matrix, sign-fixed."""
    a = rng.standard_normal(shape)
    q, r = np.linalg.qr(a)
    q *= np.sign(np.diag(r))
    return q.astype(np.float32)


def init_embeddings(shape, name, init, is_l2_norm, dtype=None):
    """initializers.py:9-19 -> EmbeddingTable."""
    from ...models.trainer import EmbeddingTable
    if init == 'xavier':
        host = xavier_host(_rng, shape)
    elif init == 'normal':
        host = truncated_normal_host(_rng, shape, 1.0 / math.sqrt(shape[1]))
    elif init == 'uniform':
        host = _rng.uniform(0.0, 1.0, shape).astype(np.float32)      # tf random_uniform(minval=0, maxval=None -> 1)
    elif init == 'unit':
        host = unit_host(_rng, shape)
    else:
        raise ValueError("unknown init %r" % (init,))
    return EmbeddingTable(host, is_l2_norm, name)


# ---- the reference's per-kind entry points (initializers.py:22-56) -> the same tables as init_embeddings -----------------
def xavier_init(shape, name, is_l2_norm, dtype=None):
    return init_embeddings(shape, name, 'xavier', is_l2_norm, dtype)


def truncated_normal_init(shape, name, is_l2_norm, dtype=None):
    return init_embeddings(shape, name, 'normal', is_l2_norm, dtype)


def random_uniform_init(shape, name, is_l2_norm, minval=0, maxval=None, dtype=None):
    from ...models.trainer import EmbeddingTable
    host = _rng.uniform(minval, 1.0 if maxval is None else maxval, shape).astype(np.float32)
    return EmbeddingTable(host, is_l2_norm, name)


def random_unit_init(shape, name, is_l2_norm, dtype=None):
    return init_embeddings(shape, name, 'unit', is_l2_norm, dtype)


def orthogonal_init(shape, name, dtype=None):
    """initializers.py:53-56 -> device fp32 [rows, cols] matrix (the mapping matrix of MTransE)."""
    import torch
    from ... import ops
    return torch.from_numpy(orthogonal_host(_rng, tuple(shape))).to(ops.device())
