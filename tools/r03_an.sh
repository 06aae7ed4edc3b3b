#!/bin/bash
set -u
TAG=${1:-r03an}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for V in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_dist_gpu.py -m gpu -q -x -s -k "two_ranks_reproduce" > $OUT/pytest_$V.log 2>&1
  echo "run $V: $(grep -h 'AliNet two ranks' $OUT/pytest_$V.log | cut -c50-150) $(grep -h -E 'passed|failed' $OUT/pytest_$V.log | tail -1)"
  grep -h "first batch\|gradient alinet_g00" $OUT/pytest_$V.log | cut -c1-160
done
