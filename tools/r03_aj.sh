#!/bin/bash
set -u
TAG=${1:-r03aj}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_reference_fullsize.py -m gpu -q -x -k "manhattan or fullsize" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest.log | tail -10
timeout 600 python tools/_exp/l1_eval_time.py 2>&1 | grep -v amdgpu | tee $OUT/l1_eval.txt
