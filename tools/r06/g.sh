#!/bin/bash
# round 6, call g: kernel traces of whole model epochs at the 100K shapes (AliNet, RDGCN, GCN-Align, BootEA) -- what an epoch consists of
set -u
export TMPDIR=/tmp
for M in AliNet RDGCN GCN_Align BootEA; do
  tools/prof.sh trace r06g_$M -- timeout 800 python tools/profile_models.py 100K $M
  tail -2 gpurun_out/r06g_$M/trace_stdout.log
  python - $M <<'PY'
import csv, sys
rows = list(csv.DictReader(open("gpurun_out/r06g_%s/trace_stats.csv" % sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("  %s: %d kernels, %.1f ms of kernel time in the trace" % (sys.argv[1], len(rows), tot / 1e6))
for r in rows[:8]:
    print("    %5.1f %%  %6s calls  %9.1f us  %s" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
cb = [r for r in rows if "copyBuffer" in r["Name"] or "fillBuffer" in r["Name"]]
for r in cb: print("    (%s: %s calls, %.2f %% of kernel time)" % (r["Name"], r["Calls"], float(r["Percentage"])))
PY
done
