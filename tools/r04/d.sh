#!/bin/bash
# round 4, call d: where does the symmetric neighbour sweep spend its cycles?  (counters of the 100,000^2 search)
export KNN_QUICK=1
tools/prof.sh list r04d
tools/prof.sh trace r04d -- python tools/_exp/knn_time.py
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  tools/prof.sh pmc r04d "$set" -- python tools/_exp/knn_time.py
done
cd gpurun_out/r04d
grep -i "SQ_WAIT\|SQ_ACTIVE\|SQ_INST\|MFMA\|SQ_BUSY\|SQ_WAVE" counters.txt | head -80 > counters_sq.txt
head -5 trace_stats.csv
for f in pmc_*.csv; do echo "== $f"; head -4 $f | cut -c1-400; done
tail -3 pmc_*_stdout.log | cut -c1-300
