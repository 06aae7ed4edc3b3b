#!/bin/bash
# round 5, call d: where the 70,000^2 x 1,200 evaluation spends its time (kernel trace) and what the big sweep's SIMDs do (counters)
set -u
export TMPDIR=/tmp
T=r05d
tools/prof.sh trace $T -- python tools/_exp/eval1200_trace.py 1200 2
tools/prof.sh pmc $T "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" -- python tools/_exp/eval1200_trace.py 1200 1
tools/prof.sh pmc $T "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" -- python tools/_exp/eval1200_trace.py 1200 1
tools/prof.sh pmc $T "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" -- python tools/_exp/eval1200_trace.py 1200 1
tools/prof.sh pmc $T "TCC_HIT_sum TCC_MISS_sum" -- python tools/_exp/eval1200_trace.py 1200 1
tools/prof.sh pmc $T "FETCH_SIZE" -- python tools/_exp/eval1200_trace.py 1200 1
tools/prof.sh pmc $T "SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" -- python tools/_exp/eval1200_trace.py 1200 1
head -25 gpurun_out/$T/trace_stats.csv | cut -c1-220
for f in gpurun_out/$T/pmc_*.csv; do echo == $f; head -6 $f | cut -c1-250; done
