#!/bin/bash
# round 5, call g: the software-pipelined step kernel (LDS-DMA gathers) -- bit equality with the one-positive kernel, step time at both shapes
set -u
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "software_pipelined" 2>&1 | tail -15 ) > $O/pytest_pipe.log 2>&1
for PIPE in 0 1; do
  ( OEA_STEP_PIPE=$PIPE timeout 600 python bench.py --steps 56 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic 2>&1 | tail -1 ) > $O/bench100k_pipe$PIPE.log 2>&1
done
for NP in 3 5 6; do
  ( OEA_STEP_PIPE=1 OEA_STEP_PIPE_NP=$NP timeout 600 python bench.py --steps 56 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic 2>&1 | tail -1 ) > $O/bench100k_np$NP.log 2>&1
done
( OEA_STEP_PIPE=1 timeout 600 python bench.py --shape EN-FR-15K-V1 --steps 50 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic 2>&1 | tail -1 ) > $O/bench15k_pipe1.log 2>&1
( OEA_STEP_PIPE=0 timeout 600 python bench.py --shape EN-FR-15K-V1 --steps 50 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic 2>&1 | tail -1 ) > $O/bench15k_pipe0.log 2>&1
tail -4 $O/pytest_pipe.log; for f in $O/bench*.log; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print("value %.1f M/s  ms/step %.4f  grouped %.2f us  apply %.2f us  frac %.3f" % (j["value"] / 1e6, j["ms_per_step"], r["avg_kernel_us"], r.get("apply_rows_avg_us", 0), r["frac"]))
except Exception as e:
    print("parse failed", e, open(sys.argv[1]).read()[-600:])
PY
done
