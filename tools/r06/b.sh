#!/bin/bash
# round 6, call b: full-size manhattan parity (70,000^2 x 300), where triple_wave's time goes (atomics / loads dropped by empty
# buffer resources), its counters
set -u
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -k "manhattan_rdgcn or bf16_csls_row_blocks" 2>&1 | tail -15 ) > $O/pytest_l1.log 2>&1
tail -5 $O/pytest_l1.log
B="python bench.py --steps 56 --warmup 5 --repeats 10 --no-extra --no-gnn --no-cpu --no-traffic"
for D in 0 1 2 3 4 7; do
  ( OEA_STEP_WAVE_DBG=$D timeout 600 $B 2>&1 | tail -1 ) > $O/bench100k_dbg$D.log 2>&1
  ( OEA_STEP_WAVE_DBG=$D timeout 600 $B --shape EN-FR-15K-V1 2>&1 | tail -1 ) > $O/bench15k_dbg$D.log 2>&1
done
for f in $O/bench*.log; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print("value %.1f M/s  ms/step %.4f  grad %.2f us  apply %.2f us  frac %.3f" % (j["value"] / 1e6, j["ms_per_step"], r["avg_kernel_us"], r.get("apply_rows_avg_us", 0), r["frac"]))
except Exception as e:
    print("parse failed", e, open(sys.argv[1]).read()[-600:])
PY
done
tools/prof.sh list r06b_pmc
C="timeout 300 python bench.py --steps 30 --warmup 5 --repeats 2 --no-extra --no-gnn --no-cpu --no-traffic"
tools/prof.sh pmc r06b_pmc "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" -- $C
tools/prof.sh pmc r06b_pmc "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" -- $C
tools/prof.sh pmc r06b_pmc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- $C
tools/prof.sh pmc r06b_pmc "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_EA0_WRREQ_64B_sum" -- $C
head -4 $O/../r06b_pmc/pmc_*.csv | cut -c1-400
