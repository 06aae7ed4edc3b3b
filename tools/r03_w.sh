#!/bin/bash
set -u
TAG=${1:-r03w}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gnn_gpu.py tests/test_graph_golden.py tests/test_models_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest.log | tail -20
{
OEA_RDGCN_LOSS=rows python tools/_exp/epoch_time.py RDGCN 100K 20
OEA_RDGCN_LOSS=atomic python tools/_exp/epoch_time.py RDGCN 100K 20
OEA_RDGCN_LOSS=rows python tools/_exp/epoch_time.py RDGCN 15K 20
OEA_RDGCN_LOSS=atomic python tools/_exp/epoch_time.py RDGCN 15K 20
python tools/_exp/epoch_time.py AliNet 100K 10
python tools/_exp/epoch_time.py AliNet 15K 10
python tools/_exp/epoch_time.py GCN_Align 15K 200
} 2>&1 | grep "ms/epoch" | tee $OUT/epochs.txt
cd /tmp && export TMPDIR=/tmp
OEA_RDGCN_LOSS=rows timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python $R/tools/profile_models.py 100K RDGCN > $OUT/log2.txt 2>&1
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/RDGCN_100k_kernel_stats.csv && head -6 $f | cut -c1-150
rm -rf $OUT/tr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python $R/tools/profile_models.py 15K MTransE > $OUT/log.txt 2>&1
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/MTransE_15k_kernel_stats.csv && head -6 $f | cut -c1-150
rm -rf $OUT/tr
