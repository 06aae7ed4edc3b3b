"""host-side profile (cProfile) of an approach's epochs at a synthetic shape: python prof_host.py NAME 15K|100K [epochs]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.profile_models import MODE, SHAPE  # noqa: E402
import openea_amd.approaches as approaches  # noqa: E402
from openea_amd.models import trans  # noqa: E402
from openea_amd.modules.load.synth import make_kgs  # noqa: E402
from openea_amd.run.default_args import get_args  # noqa: E402

name, scale = sys.argv[1], sys.argv[2]
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kgs = make_kgs(SHAPE[scale][name], mode=MODE.get(name, "mapping"), seed=0)
m = (getattr(approaches, name, None) or getattr(trans, name))()
m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1,
                    start_valid=10 ** 6, eval_freq=10 ** 6, sub_epoch=10))
m.set_kgs(kgs)
m.args.random_name_init = True
with contextlib.redirect_stdout(io.StringIO()):
    m.init()
    m.args.max_epoch = 10 if name.startswith("BootEA") else 1
    m.run()
    torch.cuda.synchronize()
    m.args.max_epoch = epochs
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    m.run()
    torch.cuda.synchronize()
    pr.disable()
    wall = time.time() - t0
print("==== %s %s: %.2f ms/epoch over %d epochs (under cProfile)" % (name, scale, wall / epochs * 1e3, epochs))
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(14)
print("\n".join(l for l in out.getvalue().splitlines() if l.strip() and "function calls" not in l and "Ordered by" not in l)[:3500])
