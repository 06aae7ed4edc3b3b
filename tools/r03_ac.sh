#!/bin/bash
set -u
TAG=${1:-r03ac}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_dist_cpu.py -m gpu -q -x -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert|exchange|partition" $OUT/pytest.log | tail -14
