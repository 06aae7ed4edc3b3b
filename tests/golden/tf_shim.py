"""A lazily evaluated numpy stand-in for the TensorFlow-1 surface that the reference's translational graph code touches
(placeholders, variables, lookups, l2_normalize, the element-wise / reduction ops of modules/base/losses.py and of the
approach classes' `_define_*_graph` methods).  TEST INFRASTRUCTURE for tests/golden/make_tf_graph_golden.py, which
installs it as `tensorflow`, lets the REFERENCE's own code build its loss graphs and evaluates them in float64: the
forward semantics then come from the reference's source, not from a restatement.  Gradients are NOT provided (optimisers
are inert); the generator takes central finite differences of the evaluated loss instead.

Op semantics supplied here (TF-1 documented behaviour): l2_normalize(x, axis) = x * rsqrt(max(sum(x^2, axis), 1e-12));
embedding_lookup = row gather; reduce_sum(axis, keep_dims); norm(axis) = sqrt(sum(x^2)); everything else is numpy's.
"""
import contextlib
import types

import numpy as np

float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64


class Node:
    def __init__(self, fn, *inputs, name=None):
        self.fn, self.inputs, self.name = fn, inputs, name

    def value(self, env):
        key = id(self)
        if key not in env["cache"]:
            env["cache"][key] = self.fn(*[_val(x, env) for x in self.inputs])
        return env["cache"][key]

    def eval(self, session=None, feed_dict=None):
        return evaluate(self, feed_dict)

    def __add__(self, o): return Node(np.add, self, o)
    def __radd__(self, o): return Node(np.add, o, self)
    def __sub__(self, o): return Node(np.subtract, self, o)
    def __rsub__(self, o): return Node(np.subtract, o, self)
    def __mul__(self, o): return Node(np.multiply, self, o)
    def __rmul__(self, o): return Node(np.multiply, o, self)
    def __truediv__(self, o): return Node(np.divide, self, o)
    def __rtruediv__(self, o): return Node(np.divide, o, self)
    def __neg__(self): return Node(np.negative, self)
    def __getitem__(self, idx): return Node(lambda v: v[idx], self)

    @property
    def shape(self):
        return np.shape(evaluate(self))                 # static shapes: only asked of placeholder-free nodes
    def __pow__(self, o): return Node(np.power, self, o)
    __array_priority__ = 1000
    __array_ufunc__ = None          # numpy operands defer to the reflected methods above


class Variable(Node):
    def __init__(self, data, name=None, **_):
        self.data, self.name = np.array(data, np.float64), name
        VARIABLES.append(self)

    def value(self, env):
        return env["vars"].get(id(self), self.data)


class Placeholder(Node):
    def __init__(self, dtype=None, shape=None, name=None, default=None):
        self.name, self.default = name, default
        if name:
            PLACEHOLDERS[name] = self

    def value(self, env):
        for k, v in env["feed"].items():
            if k is self or (isinstance(k, str) and self.name and k == self.name + ":0"):
                return v if isinstance(v, tuple) else np.asarray(v)
        if self.default is not None:
            return self.default
        raise KeyError("placeholder %r not fed" % (self.name,))


VARIABLES = []
PLACEHOLDERS = {}
_RNG = np.random.RandomState(20190719)


def _val(x, env):
    return x.value(env) if isinstance(x, Node) else x


def evaluate(fetch, feed_dict=None, var_overrides=None):
    env = {"feed": feed_dict or {}, "cache": {}, "vars": var_overrides or {}}
    if isinstance(fetch, dict):
        return {k: _val(v, env) for k, v in fetch.items()}
    if isinstance(fetch, (list, tuple)):
        return [_val(v, env) for v in fetch]
    return _val(fetch, env)


# ---- graph construction API -------------------------------------------------------------------------------------------
def placeholder(dtype=None, shape=None, name=None):
    return Placeholder(dtype, shape, name)


def sparse_placeholder(dtype=None, shape=None, name=None):
    """fed with the (coords [nnz, 2], values [nnz], dense_shape) tuples of sparse_to_tuple"""
    return Placeholder(dtype, shape, name)


def placeholder_with_default(input, shape=None, name=None):      # noqa: A002
    return Placeholder(None, shape, name, default=input)


class _Graph:
    def get_tensor_by_name(self, name):
        return PLACEHOLDERS[name.split(":")[0]]                    # KeyError -> the caller creates the placeholder


def get_default_graph():
    return _Graph()


GraphKeys = types.SimpleNamespace(GLOBAL_VARIABLES="variables")


def get_collection(key, scope=None):
    return list(VARIABLES)


def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return _truncated_normal(stddev)(shape) + mean


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, seed=None, name=None):
    return _RNG.uniform(minval, maxval, shape)


def zeros(shape, dtype=None, name=None):
    return np.zeros(shape)


def ones(shape, dtype=None, name=None):
    return np.ones(shape)


def add_n(inputs, name=None):
    return Node(lambda *v: sum(v[1:], v[0]), *inputs)


def _spmm(a, x):
    import scipy.sparse as sp
    coords, values, shape = a
    coords = np.asarray(coords, np.int64)
    return sp.csr_matrix((np.asarray(values, np.float64), (coords[:, 0], coords[:, 1])), shape=tuple(shape)) @ x


class SparseTensor:
    """indices [nnz, 2] (host), values (host array or Node), dense_shape"""

    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = np.asarray(indices, np.int64).reshape(-1, 2), values, tuple(dense_shape)

    def __mul__(self, dense):
        """sparse * dense with broadcasting of a [n, 1] column or a [1, n] row (alinet.py:665-666)"""
        rows, cols = self.indices[:, 0], self.indices[:, 1]

        def fn(vals, d):
            d = np.asarray(d)
            picked = d[rows, 0] if d.shape[1] == 1 and d.shape[0] != 1 else (d[0, cols] if d.shape[0] == 1 else d[rows, cols])
            return np.asarray(vals, np.float64) * picked
        return SparseTensor(self.indices, Node(fn, self.values, dense), self.dense_shape)


def sparse_tensor_dense_matmul(sp_a, b, name=None):
    if isinstance(sp_a, SparseTensor):
        return Node(lambda vals, x: _spmm((sp_a.indices, vals, sp_a.dense_shape), x), sp_a.values, b)
    return Node(_spmm, sp_a, b)


# What tf.sparse_softmax does with a tensor whose indices are NOT in canonical row-major order cannot be checked here (no
# TensorFlow).  Three candidate readings, selected by SPARSE_SOFTMAX_MODE at evaluation time:
#   'runs'    softmax over each RUN of consecutive entries with the same row (SparseTensor::group on the
#             tensor as fed; on canonical order this is the per-row softmax);
#   'row'     softmax over all entries of a row wherever they stand (the documented meaning of the op);
#   'reorder' the builder's recollection of sparse_softmax_op.cc: the kernel deep-copies the tensor, Reorder()s it to
#             row-major order, normalises each row's group and writes the groups back to back -- i.e. output value p is
#             the softmax value of the p-th entry in SORTED order, and the python wrapper pairs it with the p-th index
#             of the tensor AS FED (for a symmetric sparsity pattern fed column-major, as AliNet does: alpha[r, c]
#             becomes the row-c softmax weight of entry (c, r), the transposed attention).
SPARSE_SOFTMAX_MODE = 'runs'


def _run_softmax(rows, vals):
    out = np.empty_like(vals, dtype=np.float64)
    start = 0
    for i in range(1, len(rows) + 1):
        if i == len(rows) or rows[i] != rows[start]:
            seg = vals[start:i] - vals[start:i].max()
            e = np.exp(seg)
            out[start:i] = e / e.sum()
            start = i
    return out


def _sparse_softmax_values(indices, vals):
    rows = indices[:, 0]
    if SPARSE_SOFTMAX_MODE == 'runs':
        return _run_softmax(rows, vals)
    order = np.lexsort((indices[:, 1], rows))                    # canonical row-major order
    sorted_out = _run_softmax(rows[order], vals[order])
    if SPARSE_SOFTMAX_MODE == 'reorder':
        return sorted_out                                        # left in sorted order, paired with the indices as fed
    assert SPARSE_SOFTMAX_MODE == 'row'
    out = np.empty_like(sorted_out)
    out[order] = sorted_out
    return out


def sparse_softmax(sp_input, name=None):
    idx = np.asarray(sp_input.indices)
    return SparseTensor(sp_input.indices, Node(lambda v: _sparse_softmax_values(idx, np.asarray(v, np.float64)), sp_input.values),
                        sp_input.dense_shape)


def sparse_add(a, b, name=None):
    assert np.array_equal(a.indices, b.indices)
    return SparseTensor(a.indices, Node(np.add, a.values, b.values), a.dense_shape)


def sparse_reshape(sp_input, shape, name=None):
    assert tuple(shape) == tuple(sp_input.dense_shape)
    return sp_input


def tile(x, multiples, name=None):
    return Node(lambda v: np.tile(v, multiples), x)


def glorot_uniform_initializer(**_):
    return lambda shape: _RNG.uniform(-np.sqrt(6.0 / (shape[0] + shape[1])), np.sqrt(6.0 / (shape[0] + shape[1])), shape)


def zeros_initializer(**_):
    return lambda shape: np.zeros(shape)


def expand_dims(x, axis=None, name=None, dim=None):
    return Node(lambda v: np.expand_dims(v, axis if axis is not None else dim), x)


def transpose(x, perm=None, name=None):
    return Node(lambda v: np.transpose(v, perm), x)


def concat(values, axis, name=None):
    return Node(lambda *v: np.concatenate(v, axis=axis), *values)


def cast(x, dtype=None, name=None):
    if isinstance(x, SparseTensor):
        return x
    to = np.int64 if dtype in (np.int32, np.int64) else np.float64
    return Node(lambda v: np.asarray(v, to), x) if isinstance(x, Node) else np.asarray(x, to)


def reset_default_graph():
    PLACEHOLDERS.clear()


def _conv1d(inputs, filters, kernel_size, use_bias=True, **_):
    """tf.layers.conv1d with kernel size 1 = a dense layer over the last axis: glorot-uniform kernel [1, C, F], zero bias.
    The channel count comes from evaluating the (placeholder-free) input once."""
    assert kernel_size == 1
    c_in = np.shape(evaluate(inputs))[-1]
    lim = np.sqrt(6.0 / (c_in + filters))
    kernel = Variable(_RNG.uniform(-lim, lim, (1, c_in, filters)), name="conv1d_kernel_%d" % len(VARIABLES))
    out = Node(lambda x, k: np.matmul(x, k[0]), inputs, kernel)
    if use_bias:
        bias = Variable(np.zeros(filters), name="conv1d_bias_%d" % len(VARIABLES))
        out = out + bias
    return out


layers = types.SimpleNamespace(conv1d=_conv1d)


def constant(value, dtype=None, name=None):
    return np.asarray(value, np.float64) if not isinstance(value, (int, float)) else float(value)


def get_variable(name, shape=None, dtype=None, initializer=None, **_):
    return Variable(initializer(shape), name=name)


def name_scope(*a, **k):
    return contextlib.nullcontext()


variable_scope = name_scope


def global_variables_initializer():
    return types.SimpleNamespace(run=lambda session=None: None)


def _op(fn):
    return lambda *a, name=None, **k: Node((lambda *v: fn(*v, **k)), *a)


def _reduce_sum(x, axis=None, keep_dims=False, keepdims=False):
    return np.sum(x, axis=axis, keepdims=bool(keep_dims or keepdims))


def reduce_sum(x, axis=None, keep_dims=False, keepdims=False, name=None):
    return Node(lambda v: _reduce_sum(v, axis, keep_dims, keepdims), x)


def reduce_mean(x, axis=None, keep_dims=False, keepdims=False, name=None):
    return Node(lambda v: np.mean(v, axis=axis, keepdims=bool(keep_dims or keepdims)), x)


abs = _op(np.abs)                      # noqa: A001 (the names are TensorFlow's)
square = _op(np.square)
exp = _op(np.exp)
log = _op(np.log)
sin = _op(np.sin)
cos = _op(np.cos)
add = _op(np.add)
multiply = _op(np.multiply)
pow = _op(np.power)                    # noqa: A001
sigmoid = _op(lambda x: 1.0 / (1.0 + np.exp(-x)))
tanh = _op(np.tanh)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None, a_is_sparse=False, b_is_sparse=False):
    return Node(lambda x, y: np.matmul(x.T if transpose_a else x, y.T if transpose_b else y), a, b)


def reshape(x, shape, name=None):
    return Node(lambda v: np.reshape(v, shape), x)


def stack(values, axis=0, name=None):
    return Node(lambda *v: np.stack(v, axis=axis), *values)


def norm(x, ord="euclidean", axis=None, name=None, **_):
    return Node(lambda v: np.sqrt(np.sum(v * v, axis=axis)), x)


def _l2_normalize(x, axis=None, epsilon=1e-12, dim=None, name=None):
    ax = axis if axis is not None else dim
    return Node(lambda v: v / np.sqrt(np.maximum(np.sum(v * v, axis=ax, keepdims=True), epsilon)), x)


def _softmax(x, axis=-1):
    e = np.exp(x - np.max(x, axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def _dropout(x, keep_prob=None, **_):
    assert keep_prob in (None, 1, 1.0), "the golden graphs are built with dropout off"
    return x


nn = types.SimpleNamespace(
    embedding_lookup=lambda params, ids, name=None: Node(lambda p, i: np.asarray(p)[np.asarray(i, np.int64)], params, ids),
    l2_normalize=_l2_normalize,
    relu=_op(lambda x: np.maximum(x, 0.0)),
    leaky_relu=lambda x, alpha=0.2, name=None: Node(lambda v: np.where(v > 0, v, alpha * v), x),
    softmax=lambda x, axis=-1, name=None: Node(lambda v: _softmax(v, axis), x),
    sigmoid=_op(lambda x: 1.0 / (1.0 + np.exp(-x))),
    dropout=_dropout,
)
nn.bias_add = lambda value, bias, name=None: Node(np.add, value, bias)


class _BatchNormalization:
    """tf.keras.layers.BatchNormalization called without `training` in a TF-1 graph: inference mode with the initial moving
    statistics (mean 0, variance 1), epsilon 1e-3: y = gamma * x / sqrt(1 + 1e-3) + beta.  gamma / beta are created at the
    first call, like keras builds a layer."""
    count = 0

    def __init__(self, **_):
        self.gamma = self.beta = None
        _BatchNormalization.count += 1
        self.uid = _BatchNormalization.count

    def __call__(self, x, training=None):
        if self.gamma is None:
            dim = np.shape(evaluate(x))[-1]
            self.gamma = Variable(np.ones(dim), name="bn%d_gamma" % self.uid)
            self.beta = Variable(np.zeros(dim), name="bn%d_beta" % self.uid)
        return x * (self.gamma / np.sqrt(1.0 + 1e-3)) + self.beta


keras = types.SimpleNamespace(
    activations=types.SimpleNamespace(get=lambda name: {"relu": nn.relu, "tanh": tanh}[name], relu=nn.relu, tanh=tanh),
    layers=types.SimpleNamespace(BatchNormalization=_BatchNormalization))


def _truncated_normal(stddev=1.0, **_):
    def init(shape):
        v = _RNG.standard_normal(shape)
        while True:
            bad = np.abs(v) > 2.0
            if not bad.any():
                return v * stddev
            v[bad] = _RNG.standard_normal(int(bad.sum()))
    return init


initializers = types.SimpleNamespace(
    truncated_normal=_truncated_normal,
    random_uniform=lambda minval=0, maxval=None, **_: (lambda shape: _RNG.uniform(minval, 1.0 if maxval is None else maxval, shape)),
    orthogonal=lambda **_: (lambda shape: np.linalg.qr(_RNG.standard_normal(shape))[0]),
)
contrib = types.SimpleNamespace(layers=types.SimpleNamespace(
    xavier_initializer=lambda uniform=False, **_: (lambda shape: _RNG.standard_normal(shape) * np.sqrt(2.0 / sum(shape))),
    l2_regularizer=lambda scale=0.0, **_: None))        # regularisation losses are collected by TF but never added to a loss


class _InertOptimizer:
    """compute_gradients / apply_gradients of tf.train.*Optimizer: the golden generator differentiates numerically."""

    def __init__(self, *a, **k):
        pass

    def compute_gradients(self, loss, var_list=None):
        return []

    def apply_gradients(self, grads_and_vars):
        return Node(lambda: None)

    def minimize(self, loss, **_):
        return Node(lambda: None)


train = types.SimpleNamespace(GradientDescentOptimizer=_InertOptimizer, AdagradOptimizer=_InertOptimizer,
                              AdadeltaOptimizer=_InertOptimizer, AdamOptimizer=_InertOptimizer)


class Session:
    def __init__(self, *a, **k):
        pass

    def run(self, fetches=None, feed_dict=None):
        return evaluate(fetches, feed_dict)


def ConfigProto(*a, **k):
    return types.SimpleNamespace(gpu_options=types.SimpleNamespace(allow_growth=False))
