"""AliNet / RDGCN epochs at a given shape, a few epochs after warm-up: for rocprofv3 --kernel-trace --stats.
usage: gnn_epochs.py AliNet|RDGCN shape epochs"""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import openea_amd.approaches as approaches  # noqa: E402
from openea_amd.modules.load.synth import make_kgs  # noqa: E402
from openea_amd.run.default_args import get_args  # noqa: E402

name, shape, epochs = sys.argv[1], sys.argv[2], int(sys.argv[3])
kgs = make_kgs(shape, mode="mapping", seed=0)
m = getattr(approaches, name)()
kw = dict(random_name_init=True) if name == "RDGCN" else {}
m.set_args(get_args(name, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=2, start_valid=10**6,
                    eval_freq=10**6, **kw))
m.set_kgs(kgs)
with contextlib.redirect_stdout(io.StringIO()):
    t0 = time.time()
    m.init()
    torch.cuda.synchronize()
    t_init = time.time() - t0
    m.run()
    torch.cuda.synchronize()
    m.args.max_epoch = epochs
    t0 = time.time()
    m.run()
    torch.cuda.synchronize()
    dt = time.time() - t0
print("%s at %s: init %.2f s, %.2f ms per epoch over %d epochs" % (name, shape, t_init, dt / epochs * 1e3, epochs))
