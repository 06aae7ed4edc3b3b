"""Which torch ops (with their Python call sites) cost device time in an approach's epoch: python prof_torch_ops.py NAME 15K|100K"""
import contextlib
import io
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.profile_models import SHAPE  # noqa: E402
import openea_amd.approaches as approaches  # noqa: E402
from openea_amd.modules.load.synth import make_kgs  # noqa: E402
from openea_amd.run.default_args import get_args  # noqa: E402

name, scale = sys.argv[1], sys.argv[2]
kgs = make_kgs(SHAPE[scale][name], mode="mapping", seed=0)
m = getattr(approaches, name)()
m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1,
                    start_valid=10 ** 6, eval_freq=10 ** 6))
m.set_kgs(kgs)
m.args.random_name_init = True
with contextlib.redirect_stdout(io.StringIO()):
    m.init()
    m.run()
    m.args.max_epoch = 2
    m.run()
    torch.cuda.synchronize()
    m.args.max_epoch = 4
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        m.run()
        torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60,
                                                         max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=50,
                                                  max_src_column_width=120))
