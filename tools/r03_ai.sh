#!/bin/bash
set -u
TAG=${1:-r03ai}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for M in AlignE BootEA AliNet RDGCN GCN_Align; do
  timeout 300 python tools/_exp/prof_host.py $M 15K 20 2>&1 | grep -v amdgpu > $OUT/host_$M.txt
  head -22 $OUT/host_$M.txt | cut -c1-150
done
