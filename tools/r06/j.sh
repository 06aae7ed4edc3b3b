#!/bin/bash
# round 6, call j: kernel trace of greedy_alignment at 10,500^2 x 75 with CSLS 10 (the 15K datasets' evaluation)
set -u
tools/prof.sh trace r06j_csls10k -- python tools/_exp/csls10k_trace.py
tail -3 gpurun_out/r06j_csls10k/trace_stdout.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r06j_csls10k/trace_stats.csv")))
for r in rows: print("%5.1f %% %5s calls %9.1f us  %s" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
