#!/bin/bash
# round 6, call f: counters of the planned step's two kernels (instructions per wave, waits, fabric requests) + headline-only kernel trace
set -u
O=gpurun_out/r06f_pmc; mkdir -p $O
export TMPDIR=/tmp
C="timeout 300 python bench.py --steps 30 --warmup 5 --repeats 2 --no-extra --no-gnn --no-cpu --no-traffic"
tools/prof.sh pmc r06f_pmc "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" -- $C
tools/prof.sh pmc r06f_pmc "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" -- $C
tools/prof.sh pmc r06f_pmc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- $C
tools/prof.sh pmc r06f_pmc "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_EA0_WRREQ_64B_sum" -- $C
tools/prof.sh pmc r06f_pmc "FETCH_SIZE" -- $C
tools/prof.sh pmc r06f_pmc "WRITE_SIZE" -- $C
for f in $O/pmc_*.csv; do head -1 $f; grep "triple_wave\|apply_step_plan" $f | cut -c1-40,110-400; done
tools/prof.sh trace r06f_trace -- timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-gnn --no-cpu --no-traffic
head -8 gpurun_out/r06f_trace/trace_stats.csv | cut -c1-90,250-330
