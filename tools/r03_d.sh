#!/bin/bash
# Run ON the GPU box: fused AliNet glue -- tests + epoch times + kernel trace.
set -u
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gnn_gpu.py tests/test_graph_golden.py tests/test_models_100k_gpu.py tests/test_models_gpu.py -m gpu -q > $OUT/pytest_gnn.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gnn.log
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gnn.log | tail -20
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -s > $OUT/pytest_dist.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_dist.log
grep -E "passed|failed|FAILED|ERROR|per-epoch exchange|AliNet two" $OUT/pytest_dist.log | tail -20
python tools/profile_models.py 100K AliNet > $OUT/models_100k.txt 2>&1
python tools/profile_models.py 15K AliNet >> $OUT/models_100k.txt 2>&1
grep "epoch" $OUT/models_100k.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_AliNet -- python $R/tools/profile_models.py 100K AliNet > $OUT/models_AliNet.log 2>&1
f=$(ls $OUT/trace_AliNet/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/AliNet_kernel_stats.csv
rm -rf $OUT/trace_AliNet
