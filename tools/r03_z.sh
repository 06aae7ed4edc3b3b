#!/bin/bash
set -u
TAG=${1:-r03z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x -k "gcn or GCN or align_loss or golden" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest.log | tail -10
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "ms" in kk} for k, v in j["extra"]["gnn"].items()}))
print("value %.1f M/s frac %.3f" % (j["value"] / 1e6, j["roofline"]["frac"]))
PY
