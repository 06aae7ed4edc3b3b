#!/bin/bash
# round 6, call d: one-sided negatives vs the oracle (triple_wave's fast paths), plan tests, plan statistics + step time
set -u
O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -k "one_sided or step_plan or planned_epochs or wave_per_positive" 2>&1 | tail -25 ) > $O/pytest_plan.log 2>&1
tail -12 $O/pytest_plan.log
B="python bench.py --steps 56 --warmup 5 --repeats 20 --no-extra --no-gnn --no-cpu --no-traffic"
( timeout 600 $B --shape EN-FR-15K-V1 2>&1 | tail -1 ) > $O/bench15k.log 2>&1
cp bench_detail.json $O/bench_detail_15k.json
( timeout 600 $B 2>&1 | tail -1 ) > $O/bench100k.log 2>&1
cp bench_detail.json $O/bench_detail_100k.json
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
python - <<'PY'
import json
for t in ("15k", "100k"):
    d = json.load(open("gpurun_out/r06d/bench_detail_%s.json" % t))
    print(t, "plan:", d.get("roofline", {}).get("step_plan"))
PY
tools/prof.sh trace r06d_trace -- timeout 600 python bench.py --steps 56 --warmup 5 --repeats 10 --no-extra --no-gnn --no-cpu --no-traffic
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r06d_trace/trace_stats.csv')))[:12]:
    print(r['Name'][:60].replace('void (anonymous namespace)::',''), r['Calls'], int(float(r['AverageNs'])), r['MinNs'], r['MaxNs'])
PY
