#!/bin/bash
# round 6, call s: instruction counters of the neighbour sweep with the branch-free record epilogue and with the branching one, and of the
# planned optimiser kernel with float4 / dword rows
set -u
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
OEA_TOPK_STREAM_FAST=1 KNN_QUICK=1 tools/prof.sh pmc r06s_knn_fast "$C" -- python tools/_exp/knn_time.py
OEA_TOPK_STREAM_FAST=0 KNN_QUICK=1 tools/prof.sh pmc r06s_knn_slow "$C" -- python tools/_exp/knn_time.py
OEA_APPLY_V4=1 tools/prof.sh pmc r06s_apply_v4 "$C" -- python bench.py --steps 20 --warmup 5 --repeats 4 --no-cpu --no-traffic --no-gnn --no-extra
OEA_APPLY_V4=0 tools/prof.sh pmc r06s_apply_dw "$C" -- python bench.py --steps 20 --warmup 5 --repeats 4 --no-cpu --no-traffic --no-gnn --no-extra
for t in knn_fast knn_slow apply_v4 apply_dw; do echo "== $t"; head -4 gpurun_out/r06s_$t/pmc_1.csv | cut -c1-260; done
