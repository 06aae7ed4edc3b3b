#!/bin/bash
set -u
TAG=${1:-r03y}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_gnn_gpu.py tests/test_graph_golden.py -m gpu -q -x -k "gemm_tn or side_loss or pair_loss or alinet or AliNet or rdgcn or RDGCN or fused" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error|assert" $OUT/pytest.log | tail -20
timeout 300 python tools/_exp/gemm_tn_time.py 2>&1 | grep "M=" | tee $OUT/gemm_tn.txt
{
python tools/_exp/epoch_time.py AliNet 100K 10
python tools/_exp/epoch_time.py AliNet 15K 10
python tools/_exp/epoch_time.py RDGCN 100K 20
} 2>&1 | grep "ms/epoch" | tee $OUT/epochs.txt
