#!/bin/bash
# final records of round 3: the whole GPU suite, then tools/collect_r03.sh
set -u
TAG=${1:-r03final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|two ranks vs single|per-epoch exchange" $OUT/pytest_gpu.log | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_r03.sh $TAG
