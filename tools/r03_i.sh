#!/bin/bash
set -u
TAG=${1:-r03i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -s -k "two_ranks_reproduce" > $OUT/pytest_dist.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_dist.log
grep -E "passed|failed|FAILED|ERROR|AliNet two|gradient alinet" $OUT/pytest_dist.log | tail -40
python tools/profile_models.py 15K GCN_Align,AliNet,RDGCN > $OUT/models.txt 2>&1
python tools/profile_models.py 100K RDGCN >> $OUT/models.txt 2>&1
grep epoch $OUT/models.txt
