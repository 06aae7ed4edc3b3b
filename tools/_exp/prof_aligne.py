import contextlib, io, os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, "/root/repo")
import openea_amd.approaches as approaches
from openea_amd.modules.load.synth import make_kgs
from openea_amd.run.default_args import get_args
name = sys.argv[1]
kgs = make_kgs("EN-FR-15K-V1", mode="swapping", seed=0)
m = getattr(approaches, name)()
m.set_args(get_args(name, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1, start_valid=10**6, eval_freq=10**6))
m.set_kgs(kgs)
with contextlib.redirect_stdout(io.StringIO()):
    m.init(); m.run(); torch.cuda.synchronize()
    m.args.max_epoch = 20
    pr = cProfile.Profile(); pr.enable(); m.run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
