#!/bin/bash
set -u
TAG=${1:-r03ad}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python $R/tools/_exp/csls70k_trace.py > $OUT/log.txt 2>&1
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/csls70k_kernel_stats.csv && head -30 $f | cut -c1-160
rm -rf $OUT/tr
sed -i 's/^sys.path.insert(0, .\.\.)/sys.path.insert(0, "'$R'")/' $R/tools/_exp/l1_eval_time.py
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python $R/tools/_exp/l1_eval_time.py > $OUT/log2.txt 2>&1
f=$(ls $OUT/tr/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/l1_eval_kernel_stats.csv && head -12 $f | cut -c1-160
rm -rf $OUT/tr
