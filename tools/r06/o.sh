#!/bin/bash
# round 6, call o: the planned optimiser kernel with float4 rows (apply_step_plan_v4) against dword fragments: pipelined durations (kernel trace)
set -u
for V in 1 0; do
  OEA_APPLY_V4=$V tools/prof.sh trace r06o_v4_$V -- timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-traffic --no-gnn --no-extra
  python - $V <<'PY'
import csv, sys
rows = list(csv.DictReader(open("gpurun_out/r06o_v4_%s/trace_stats.csv" % sys.argv[1])))
for r in rows[:3]: print("V4=%s %6s calls %8.1f us avg  min %6.1f max %6.1f  %s" % (sys.argv[1], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:60]))
PY
  tail -1 gpurun_out/r06o_v4_$V/trace_stdout.log | cut -c1-200
done
