#!/bin/bash
# round 4, call r: counters of the bf16 evaluation sweep at 70,000^2 x 100 (streaming and resident-B pipelines)
export TMPDIR=/tmp
for res in 1 0; do
  export OEA_BF16_RES=$res
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
    tools/prof.sh pmc r04r_res$res "$set" -- python tools/_exp/bf16_trace.py near 70000 100 3
  done
done
cd gpurun_out
for d in r04r_res1 r04r_res0; do echo "== $d"; for f in $d/pmc_*.csv; do grep -E "^kernel|rank_bf16_(res_)?kernel<false" $f | cut -c1-60,120-400; done; done
