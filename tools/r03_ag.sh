#!/bin/bash
set -u
TAG=${1:-r03ag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "topk or neighbour or neighbor or knn or fullsize" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest.log | tail -10
timeout 300 python tools/_exp/knn_time.py 2>&1 | grep kNN | tee $OUT/knn.txt
timeout 300 python tools/_exp/knn_trained.py 2>&1 | grep -i "knn\|ms" | tail -6 | tee -a $OUT/knn.txt
