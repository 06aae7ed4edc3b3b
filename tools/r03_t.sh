#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python $R/tools/profile_models.py 15K MTransE > $OUT/log.txt 2>&1
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/MTransE_15k_kernel_stats.csv && head -14 $f | cut -c1-170
rm -rf $OUT/tr
