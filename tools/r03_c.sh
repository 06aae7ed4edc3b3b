#!/bin/bash
# Run ON the GPU box: the whole GPU suite (no -x) + a longer RDGCN / GCN-Align epoch measurement.
set -u
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|drift|per-epoch exchange" $OUT/pytest_gpu.log | tail -30
python - <<PY > $OUT/rdgcn_epochs.txt 2>&1
import contextlib, io, sys, time, torch
sys.path.insert(0, "$R")
sys.argv = ["x"]
from tools.profile_models import SHAPE
import openea_amd.approaches as approaches
from openea_amd.modules.load.synth import make_kgs
from openea_amd.run.default_args import get_args
for name in ("RDGCN", "GCN_Align"):
    kgs = make_kgs(SHAPE["100K"][name], mode="mapping", seed=0)
    m = getattr(approaches, name)()
    m.set_args(get_args(name, scale="100K", output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1,
                        start_valid=10 ** 6, eval_freq=10 ** 6))
    m.set_kgs(kgs)
    m.args.random_name_init = True
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        m.init(); m.run(); torch.cuda.synchronize()
        for ep in (9, 20):
            m.args.max_epoch = ep
            t0 = time.time(); m.run(); torch.cuda.synchronize()
            print("%s %d epochs: %.2f ms/epoch" % (name, ep, (time.time() - t0) / ep * 1e3), file=sys.stderr)
PY
cat $OUT/rdgcn_epochs.txt | grep -v amdgpu
