#!/bin/bash
set -u
TAG=${1:-r03am}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for V in 0 1 0 1; do
  OEA_TOPK_SELECT_REGS=$V timeout 600 python -m pytest tests/test_dist_gpu.py -m gpu -q -x -s -k "two_ranks_reproduce" > $OUT/pytest_regs$V.log 2>&1
  echo "REGS=$V $(grep -h 'AliNet two ranks' $OUT/pytest_regs$V.log | cut -c1-170) $(grep -h -E 'passed|failed' $OUT/pytest_regs$V.log | tail -1)"
done
