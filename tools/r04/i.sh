#!/bin/bash
export TMPDIR=/tmp
for cfg in "near 70000 100 3" "random 70000 100 3" "near 10500 75 10" "random 10500 75 10"; do
  tag=r04i_$(echo $cfg | tr ' ' '_')
  tools/prof.sh trace $tag -- python tools/_exp/bf16_trace.py $cfg
  echo "== $cfg"; grep "^[0-9]" gpurun_out/$tag/trace_stdout.log | tail -1
  python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/$tag/trace_stats.csv")))
for r in rows[:8]:
    print("   %-64s calls %4s avg %10.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
