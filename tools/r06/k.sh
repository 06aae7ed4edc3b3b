#!/bin/bash
# round 6, call k: kernel trace of the neighbour refresh at 100,000^2 x 100, k = 2,000 (random rows and a trained-like table)
set -u
KNN_QUICK=1 tools/prof.sh trace r06k_knn -- python tools/_exp/knn_time.py
tail -2 gpurun_out/r06k_knn/trace_stdout.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r06k_knn/trace_stats.csv")))
for r in rows[:16]: print("%5.1f %% %5s calls %9.1f us  %s" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
