import contextlib, io, os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, "/root/repo")
import openea_amd.approaches as approaches
from openea_amd.modules.load.synth import make_kgs
from openea_amd.run.default_args import get_args
name, scale, shape, mode = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
kgs = make_kgs(shape, mode=mode, seed=0)
m = getattr(approaches, name)()
m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=10, start_valid=10**6, eval_freq=10**6, sub_epoch=10))
m.set_kgs(kgs)
with contextlib.redirect_stdout(io.StringIO()):
    m.init(); m.run(); torch.cuda.synchronize()
    m.args.max_epoch = 20
    pr = cProfile.Profile(); pr.enable(); m.run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
