#!/bin/bash
set -u
TAG=${1:-r03m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gnn_gpu.py tests/test_graph_golden.py tests/test_models_100k_gpu.py tests/test_models_gpu.py -m gpu -q -k "rdgcn or RDGCN or hard_neg" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error" $OUT/pytest.log | tail -20
python - <<PY > $OUT/rdgcn_epochs.txt 2>&1
import contextlib, io, sys, time, torch
sys.path.insert(0, "$R")
sys.argv = ["x"]
from tools.profile_models import SHAPE
import openea_amd.approaches as approaches
from openea_amd.modules.load.synth import make_kgs
from openea_amd.run.default_args import get_args
for scale in ("15K", "100K"):
    name = "RDGCN"
    kgs = make_kgs(SHAPE[scale][name], mode="mapping", seed=0)
    m = getattr(approaches, name)()
    m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1,
                        start_valid=10 ** 6, eval_freq=10 ** 6))
    m.set_kgs(kgs)
    m.args.random_name_init = True
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        m.init(); m.run(); torch.cuda.synchronize()
        for ep in (20,):
            m.args.max_epoch = ep
            t0 = time.time(); m.run(); torch.cuda.synchronize()
            print("%s %s %d epochs: %.2f ms/epoch" % (name, scale, ep, (time.time() - t0) / ep * 1e3), file=sys.stderr)
        from openea_amd.approaches.rdgcn import get_neg
        out = m._output()
        ill = m.gcn_model.ill_dev[:, 0].contiguous()
        for exact in (True, False):
            get_neg(ill, out, m.args.dim, m.args.neg_triple_num, exact_strip=exact); torch.cuda.synchronize()
            t0 = time.time(); a = get_neg(ill, out, m.args.dim, m.args.neg_triple_num, exact_strip=exact); torch.cuda.synchronize()
            print("   get_neg exact_strip=%s: %.2f ms" % (exact, (time.time() - t0) * 1e3), file=sys.stderr)
            if exact: ref = a
        same = (ref.view(-1, m.args.neg_triple_num) == a.view(-1, m.args.neg_triple_num)).all(1).float().mean().item()
        print("   rows with the same negative set: %.4f" % same, file=sys.stderr)
PY
grep -v "amdgpu\|^/tmp\|results output" $OUT/rdgcn_epochs.txt | tail -12
