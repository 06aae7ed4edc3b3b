#!/bin/bash
# round 6, call a: one-wave-per-positive step kernel (triple_wave) -- step parity tests, step time at both shapes vs triple_grouped
set -u
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "step or triple or loss or optimis or sampler" 2>&1 | tail -15 ) > $O/pytest_step.log 2>&1
tail -5 $O/pytest_step.log
for W in 1 0; do
  ( OEA_STEP_WAVE=$W timeout 600 python bench.py --steps 56 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic 2>&1 | tail -1 ) > $O/bench100k_wave$W.log 2>&1
  ( OEA_STEP_WAVE=$W timeout 600 python bench.py --shape EN-FR-15K-V1 --steps 50 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic 2>&1 | tail -1 ) > $O/bench15k_wave$W.log 2>&1
done
for f in $O/bench*.log; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print("value %.1f M/s  ms/step %.4f  grad %.2f us  apply %.2f us  frac %.3f" % (j["value"] / 1e6, j["ms_per_step"], r["avg_kernel_us"], r.get("apply_rows_avg_us", 0), r["frac"]))
except Exception as e:
    print("parse failed", e, open(sys.argv[1]).read()[-600:])
PY
done
( timeout 900 python -m pytest tests/test_models_gpu.py tests/test_partition_gpu.py -q -x 2>&1 | tail -8 ) > $O/pytest_models.log 2>&1
tail -4 $O/pytest_models.log
