#!/bin/bash
# (every pass under its own timeout: the first version of this script had none and one counter group ran for 20+ minutes)
# round 5, call l: which unit the big sweeps wait for -- vector-memory issue, texture addresser / cache stalls (modes 2 and 1)
set -u
export TMPDIR=/tmp
T=r05l
tools/prof.sh list $T
grep -i -o "\b\(SQ_[A-Z_0-9]*VMEM[A-Z_0-9]*\|SQ_WAIT_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*\|TA_[A-Za-z_0-9]*\|TCP_[A-Za-z_0-9]*STALL[A-Za-z_0-9]*\|TCP_[A-Za-z_0-9]*BUSY[A-Za-z_0-9]*\)\b" gpurun_out/$T/counters.txt | sort -u | head -80 > gpurun_out/$T/names.txt
for M in 2 1; do
  export OEA_BF16_BIG_MODE=$M
  tools/prof.sh pmc ${T}_m$M "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" -- timeout 120 python tools/_exp/eval1200_trace.py 1200 1
  tools/prof.sh pmc ${T}_m$M "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY" -- timeout 120 python tools/_exp/eval1200_trace.py 1200 1
  tools/prof.sh pmc ${T}_m$M "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" -- timeout 120 python tools/_exp/eval1200_trace.py 1200 1
  tools/prof.sh pmc ${T}_m$M "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" -- timeout 120 python tools/_exp/eval1200_trace.py 1200 1
  tools/prof.sh pmc ${T}_m$M "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" -- timeout 120 python tools/_exp/eval1200_trace.py 1200 1
done
cat gpurun_out/$T/names.txt | tr '\n' ' '; echo
for M in 2 1; do for f in gpurun_out/${T}_m$M/pmc_*.csv; do echo "== $f"; head -3 $f | cut -c1-40,110-300; done; done
