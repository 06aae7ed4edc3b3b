#!/bin/bash
set -u
TAG=${1:-r03j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_gnn_gpu.py -m gpu -q -k "unwritten or gather_few" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|Error" $OUT/pytest.log | tail -20
cd /tmp && export TMPDIR=/tmp
LEGS=gnn timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gnn -- python $R/tools/profile_legs.py > $OUT/legs_gnn.log 2>&1
f=$(ls $OUT/trace_gnn/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/gnn_kernel_stats.csv && head -16 $f | cut -c1-160
rm -rf $OUT/trace_gnn
grep -v amdgpu $OUT/legs_gnn.log | grep -v "^W2\|^E2" | tail -6
