"""Two single-process AliNet runs from the same seeds: are the trained tables identical bit for bit?  (If not, what is left
of run-to-run noise sits in library code -- hipBLASLt split-K reductions, torch index kernels -- not in this repo's kernels.)"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd.approaches import AliNet, GCN_Align  # noqa: E402
from openea_amd.modules.load.synth import make_kgs  # noqa: E402
from openea_amd.run.default_args import get_args  # noqa: E402


def run(cls, name, **kw):
    m = cls()
    m.set_args(get_args(name, output="/tmp/oea_det/", training_data="synthetic/small/", dataset_division="f/", max_epoch=4,
                        start_valid=100, eval_freq=100, **kw))
    m.set_kgs(make_kgs("small", mode="mapping", seed=0))
    with contextlib.redirect_stdout(io.StringIO()):
        m.init()
        m.run()
    return m


a = [run(AliNet, "AliNet", layer_dims=[48, 32, 24], batch_size=600, truncated_epsilon=0.9) for _ in range(2)]
outs = [m._forward()[-1].detach().cpu().numpy() for m in a]
print("AliNet, two runs: identical bits %s, relative L2 %.2e" % (np.array_equal(*outs), np.linalg.norm(outs[0] - outs[1]) / np.linalg.norm(outs[0])))
# which step breaks it: one forward + backward on identical parameters, gradient by gradient
m = a[0]
grads = []
for _ in range(2):
    for p in m._params:
        p.grad = None
    m._rng = np.random.RandomState(5)
    m._neg_step = 77
    pos, neg, valid = m.device_input_batch(600)
    outs_ = m._forward()
    emb = m._concat_train(outs_)
    loss = m.compute_loss(emb, pos, neg, valid)
    loss.backward()
    grads.append([None if p.grad is None else p.grad.detach().cpu().numpy().copy() for p in m._params])
for i, (g0, g1) in enumerate(zip(*grads)):
    if g0 is not None and not np.array_equal(g0, g1):
        print("  gradient of parameter %d %s differs between two backward passes: max abs %.2e" % (i, tuple(g0.shape), np.abs(g0 - g1).max()))
print("gradients compared")
g = [run(GCN_Align, "GCN_Align", se_dim=32, ae_dim=16) for _ in range(2)]
o = [x.model_se.forward()[2].cpu().numpy() for x in g]
print("GCN-Align, two runs: identical bits %s, max abs %.2e" % (np.array_equal(*o), np.abs(o[0] - o[1]).max()))
