#!/bin/bash
set -u
TAG=${1:-knn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_reference_fullsize.py tests/test_gnn_gpu.py -m gpu -q -s -k "topk or neighbours or augmentation" > $OUT/pytest_knn.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_knn.log
tail -6 $OUT/pytest_knn.log
timeout 300 python tools/_exp/knn_time.py 2>&1 | grep kNN | tee $OUT/knn_lists.txt
OEA_TOPK_LISTS=0 timeout 300 python tools/_exp/knn_time.py 2>&1 | grep kNN | tee $OUT/knn_strip.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/tools/_exp/knn_time.py > $OUT/prof.log 2>&1
f=$(ls $OUT/prof/*/*_kernel_stats.csv | head -1); head -14 $f | cut -c1-200
