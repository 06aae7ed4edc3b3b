#!/bin/bash
# round 6, call c: the gathered-sum plan of an epoch -- plan == oracle restatement, planned epochs == atomic epochs, model tests, step time
set -u
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "step_plan or planned_epochs or wave_per_positive" 2>&1 | tail -15 ) > $O/pytest_plan.log 2>&1
tail -6 $O/pytest_plan.log
( timeout 1500 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py -q -x -k "not manhattan_rdgcn and not alinet_width" 2>&1 | tail -8 ) > $O/pytest_models.log 2>&1
tail -4 $O/pytest_models.log
B="python bench.py --steps 56 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic"
for P in 1 0; do
  ( OEA_STEP_PLAN=$P timeout 600 $B 2>&1 | tail -1 ) > $O/bench100k_plan$P.log 2>&1
  ( OEA_STEP_PLAN=$P timeout 600 $B --shape EN-FR-15K-V1 2>&1 | tail -1 ) > $O/bench15k_plan$P.log 2>&1
done
for f in $O/bench*.log; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print("value %.1f M/s  ms/step %.4f  grad %.2f us  apply %.2f us  frac %.3f step_frac %.3f" % (j["value"] / 1e6, j["ms_per_step"], r["avg_kernel_us"], r.get("apply_rows_avg_us", 0), r["frac"], r.get("step_frac", 0)))
except Exception as e:
    print("parse failed", e, open(sys.argv[1]).read()[-1500:])
PY
done
tools/prof.sh trace r06c_trace -- timeout 600 python bench.py --steps 56 --warmup 5 --repeats 10 --no-extra --no-gnn --no-cpu --no-traffic
head -12 gpurun_out/r06c_trace/trace_stats.csv | cut -c1-150,300-400
