#!/bin/bash
set -u
TAG=${1:-r03ah}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_models_gpu.py -m gpu -q -x -s -k "two_ranks_reproduce or mapping or MTransE or mtranse or transh or TransH" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert|two ranks vs single" $OUT/pytest.log | tail -12
{
python tools/_exp/epoch_time.py MTransE 15K 20
python tools/_exp/epoch_time.py MTransE 100K 20
python tools/profile_models.py 15K MTransE
python tools/profile_models.py 100K MTransE
} 2>&1 | grep "ms/epoch\|epoch " | tee $OUT/mtranse.txt
