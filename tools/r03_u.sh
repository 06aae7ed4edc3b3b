#!/bin/bash
# round 3, call u: MTransE mapping kernels (load clustering), RDGCN (u16 grid pre-filter, chunked segment sums, fused GCN block),
# library GEMM vs the NT tile pipeline on AliNet's dense layers
set -u
TAG=${1:-r03u}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gnn_gpu.py tests/test_graph_golden.py tests/test_models_gpu.py tests/test_models_100k_gpu.py -m gpu -q -x \
    -k "rdgcn or RDGCN or hard_neg or gather_few or mapping or MTransE or mtranse" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest.log | tail -20
timeout 300 python tools/_exp/gemm_nt.py 200000 > $OUT/gemm_nt.txt 2>&1; tail -6 $OUT/gemm_nt.txt
python - <<PY > $OUT/rdgcn_epochs.txt 2>&1
import contextlib, io, os, sys, time, torch
sys.path.insert(0, "$R")
sys.argv = ["x"]
from tools.profile_models import SHAPE
import openea_amd.approaches as approaches
from openea_amd.modules.load.synth import make_kgs
from openea_amd.run.default_args import get_args
for scale in ("15K", "100K"):
    name = "RDGCN"
    kgs = make_kgs(SHAPE[scale][name], mode="mapping", seed=0)
    for fused in ("1", "0"):
        os.environ["OEA_RDGCN_FUSED"] = fused
        m = getattr(approaches, name)()
        m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1,
                            start_valid=10 ** 6, eval_freq=10 ** 6))
        m.set_kgs(kgs)
        m.args.random_name_init = True
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            m.init(); m.run(); torch.cuda.synchronize()
            m.args.max_epoch = 20
            t0 = time.time(); m.run(); torch.cuda.synchronize()
            print("%s %s fused=%s 20 epochs: %.2f ms/epoch" % (name, scale, fused, (time.time() - t0) / 20 * 1e3), file=sys.stderr)
    from openea_amd.approaches.rdgcn import get_neg
    out = m._output()
    ill = m.gcn_model.ill_dev[:, 0].contiguous()
    ref = None
    for kw in (dict(exact_strip=True), dict(prefilter="f32"), dict(prefilter="u16")):
        st = {}
        get_neg(ill, out, m.args.dim, m.args.neg_triple_num, stats=st, **kw); torch.cuda.synchronize()
        t0 = time.time(); a = get_neg(ill, out, m.args.dim, m.args.neg_triple_num, **kw); torch.cuda.synchronize()
        print("   get_neg %s: %.2f ms %s" % (kw, (time.time() - t0) * 1e3, st), file=sys.stderr)
        if ref is None: ref = a
        same = (ref.view(-1, m.args.neg_triple_num) == a.view(-1, m.args.neg_triple_num)).all(1).float().mean().item()
        print("   rows with the same negative set as the all-pairs fp64 path: %.4f" % same, file=sys.stderr)
PY
grep -v "amdgpu\|^/tmp\|results output" $OUT/rdgcn_epochs.txt | tail -24
python - <<PY > $OUT/mtranse_epochs.txt 2>&1
import contextlib, io, sys, time, torch
sys.path.insert(0, "$R")
sys.argv = ["x"]
from tools.profile_models import SHAPE
import openea_amd.approaches as approaches
from openea_amd.modules.load.synth import make_kgs
from openea_amd.run.default_args import get_args
for scale in ("15K", "100K"):
    name = "MTransE"
    kgs = make_kgs(SHAPE[scale][name], mode="mapping", seed=0)
    m = getattr(approaches, name)()
    m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1,
                        start_valid=10 ** 6, eval_freq=10 ** 6))
    m.set_kgs(kgs)
    with contextlib.redirect_stdout(io.StringIO()):
        m.init(); m.run(); torch.cuda.synchronize()
        m.args.max_epoch = 10
        t0 = time.time(); m.run(); torch.cuda.synchronize()
    print("%s %s 10 epochs: %.2f ms/epoch" % (name, scale, (time.time() - t0) / 10 * 1e3))
PY
tail -4 $OUT/mtranse_epochs.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python $R/tools/profile_models.py 15K MTransE > $OUT/log.txt 2>&1
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/MTransE_15k_kernel_stats.csv && head -8 $f | cut -c1-150
rm -rf $OUT/tr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python $R/tools/profile_models.py 100K RDGCN > $OUT/log2.txt 2>&1
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/RDGCN_100k_kernel_stats.csv && head -12 $f | cut -c1-150
rm -rf $OUT/tr
