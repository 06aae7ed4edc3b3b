#!/bin/bash
set -u
TAG=${1:-r03k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for G in 32 16 8; do
  OEA_SPMM_G=$G python - <<PY 2>&1 | grep -v amdgpu
import sys, json
sys.argv = ["bench.py"]
import torch, bench
from openea_amd import ops
ops.lib()
g = bench.gnn_legs(torch, ops, torch.device("cuda", 0))
e = g["gcn_align_se_epoch_DW15K"]
print("G=$G: GCN-Align SE epoch %.4f ms, one aggregate %.4f ms (frac %.3f); AliNet epoch %.2f ms" % (e["ms_per_epoch"], e["roofline"]["ms"], e["roofline"]["frac"], g["alinet_EN-DE-100K"]["ms_per_epoch"]))
PY
done
