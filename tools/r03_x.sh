#!/bin/bash
set -u
TAG=${1:-r03x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python tools/_exp/prof_torch_ops.py AliNet 100K > $OUT/alinet_ops.txt 2>&1
timeout 600 python tools/_exp/prof_torch_ops.py RDGCN 100K > $OUT/rdgcn_ops.txt 2>&1
tail -5 $OUT/alinet_ops.txt
