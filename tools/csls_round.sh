#!/bin/bash
set -u
TAG=${1:-csls}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_reference_fullsize.py -m gpu -q -s -k "csls or greedy_alignment" > $OUT/pytest_csls.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_csls.log
grep -E "differ|passed|failed|FAILED|Error" $OUT/pytest_csls.log | head -30
timeout 600 python tools/_exp/csls_time.py 2>&1 | grep -E "eval|csls" | tee $OUT/csls_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/tools/_exp/csls_time.py small > $OUT/prof.log 2>&1
f=$(ls $OUT/prof/*/*_kernel_stats.csv | head -1); head -16 $f | cut -c1-160
