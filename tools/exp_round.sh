#!/bin/bash
# Run ON the GPU box: quick A/B of step-kernel variants (env switches) with the default bench, no cpu / extra legs.
#   gpurun --timeout 900 -- 'bash tools/exp_round.sh r02x "OEA_STEP_RUNTIME_KIND=1" "OEA_STEP_RUNTIME_KIND=0"'
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_graph_golden.py tests/test_fullsize_gpu.py -m gpu -q -x -k "step or triple or graph" > $OUT/pytest_step.log 2>&1
tail -3 $OUT/pytest_step.log
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout 300 python bench.py --no-cpu --no-extra > $OUT/bench_v$i.json 2> $OUT/bench_v$i.err
  env $v timeout 300 python bench.py --shape EN-FR-100K-V1 --dim 100 --batch 20000 --eps 0.98 --steps 80 --warmup 10 --repeats 10 --no-cpu --no-extra > $OUT/bench100k_v$i.json 2> $OUT/bench100k_v$i.err
  python - "$v" $OUT/bench_v$i.json $OUT/bench100k_v$i.json <<'PY'
import json, sys
for f in sys.argv[2:]:
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = j["roofline"]
        print(sys.argv[1], f.split("/")[-1], "value %.1f M/s  ms/step %.4f  fwd %.2f us  apply %.2f us" % (j["value"] / 1e6, j["ms_per_step"], r["avg_kernel_us"], r["apply_rows_avg_us"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
done
