#!/bin/bash
# round 4, call zb: counters of the stream neighbour search at 100,000 rows (sweep, bucketing, select)
# (a fourth pass with "FETCH_SIZE WRITE_SIZE" ran into the call's 20-minute limit on this workload -- 7 searches of ~30 kernels with
#  multi-GB working sets -- and cost the round 20 GPU-minutes: not repeated here; bench.py's pmc_gnn leg measures that traffic on a
#  single call behind marker launches)
export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_WR"; do
  timeout 240 tools/prof.sh pmc r04zb "$set" -- python tools/_exp/knn_abl.py
done
cd gpurun_out/r04zb; for f in pmc_*.csv; do echo "== $f"; head -4 $f | cut -c1-60,110-400; done
