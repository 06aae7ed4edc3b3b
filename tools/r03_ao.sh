#!/bin/bash
set -u
TAG=${1:-r03ao}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_partition_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert|EPOCH" $OUT/pytest.log | tail -12
