"""where do AliNet's __amd_rocclr_copyBuffer launches come from?  torch profiler over one epoch at the EN-DE-100K shape: every runtime
memcpy call (hipMemcpy*), grouped by the chain of torch ops above it and the python frame that issued it"""
import collections, contextlib, io, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import openea_amd.approaches as approaches
from openea_amd.modules.load.synth import make_kgs
from openea_amd.run.default_args import get_args
name = sys.argv[1] if len(sys.argv) > 1 else "AliNet"
shape = {"AliNet": "EN-DE-100K-V1", "GCN_Align": "EN-FR-100K-V1"}[name]
kgs = make_kgs(shape, mode="mapping", seed=0)
m = getattr(approaches, name)()
m.set_args(get_args(name, scale="100K", output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1, start_valid=10**6, eval_freq=10**6))
m.set_kgs(kgs)
with contextlib.redirect_stdout(io.StringIO()):
    m.init(); m.run(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m.run(); torch.cuda.synchronize()
groups, dev = collections.Counter(), collections.Counter()
for e in prof.events():
    nm = e.name or ""
    if "emcpy" in nm and e.device_type.name == "CPU":
        chain, p = [], e.cpu_parent
        stack = None
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            if stack is None and p.stack:
                st = [s for s in p.stack if "openea_amd" in s]
                stack = st[0].split("/")[-1][:80] if st else None
            p = p.cpu_parent
        groups[(nm, " <- ".join(chain), stack)] += 1
    if e.device_type.name != "CPU" and ("emcpy" in nm or "copyBuffer" in nm):
        dev[nm] += 1
print("device-side copy events:", dict(dev))
print("runtime memcpy calls in one epoch: %d" % sum(groups.values()))
for (nm, chain, stack), c in groups.most_common(25):
    print("  %4d  %s | %s | %s" % (c, nm, chain, stack))
