#!/bin/bash
# round 4, call g: balanced K-chunk schedule of the packed tile pipeline: exactness + timings
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_reference_fullsize.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests.log 2>&1
( timeout 300 python bench.py --no-gnn --no-cpu --no-traffic --steps 20 --warmup 5 2>&1 | tail -1 ) > $O/bench.log 2>&1
KNN_QUICK=1 tools/prof.sh trace r04g -- python tools/_exp/knn_time.py
tail -3 $O/tests.log
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r04g/bench.log") if l.startswith("{")][-1])
x=j["extra"]; s=x["shape_100k"]
print("15K eval inner %.1f csls %.1f knn %.1f" % (x["eval_pairs_per_s_inner"], x["eval_pairs_per_s_inner_csls10"], x["neighbour_rows_per_s"]))
print("100K eval inner %.1f csls %.1f knn %.1f" % (s["eval_pairs_per_s_inner"], s["eval_pairs_per_s_inner_csls10"], s["neighbour_rows_per_s"]))
print("roofline_eval", j["roofline_eval"]["frac"])
PY
head -4 gpurun_out/r04g/trace_stats.csv | cut -c1-60,300-420
