#!/bin/bash
set -u
TAG=${1:-r03ae}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x -k "csls or CSLS or fullsize or alignment" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest.log | tail -10
timeout 600 python tools/_exp/csls_time.py 2>&1 | grep eval | tee $OUT/csls_time.txt
