#!/bin/bash
# Run ON the GPU box: full suite + bench + kNN segment-capacity experiment.
set -u
TAG=${1:-r03l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|per-epoch exchange|AliNet two|end-to-end forward" $OUT/pytest_gpu.log | tail -30
for C in ; do
  OEA_TOPK_CCAP=$C timeout 300 python tools/_exp/knn_time.py > $OUT/knn_ccap$C.log 2>&1
  echo "ccap $C: $(grep -h -i "knn\|ms" $OUT/knn_ccap$C.log | tail -3 | tr '\n' ' ')"
done
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err ) 2> $OUT/bench_time.txt
python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_driver_like.json") if l.startswith("{")][-1])
e = j["extra"]
print("value %.1f M/s, eval inner %.1f M pairs/s (mfma frac %.3f), csls %.1f, knn %.2f M rows/s" % (j["value"] / 1e6, e["eval_pairs_per_s_inner"] / 1e6, e["eval_inner_mfma_frac"], e["eval_pairs_per_s_inner_csls10"] / 1e6, e["neighbour_rows_per_s"] / 1e6))
s = e["shape_100k"]
print("100k: value %.1f M/s, eval %.2f M, csls %.2f M, knn %.2f M rows/s" % (s["value"] / 1e6, s["eval_pairs_per_s_inner"] / 1e6, s["eval_pairs_per_s_inner_csls10"] / 1e6, s["neighbour_rows_per_s"] / 1e6))
g = e["gnn"]
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "ms" in kk} for k, v in g.items()}) if "error" not in g else g)
PY
cat $OUT/bench_time.txt
