#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace stats of the default bench + the two HBM-traffic PMC
# passes (FETCH_SIZE / WRITE_SIZE in their own runs, no other trace domains), into gpurun_out/$1.
#   gpurun -- 'bash tools/collect_profiles.sh r01d'
# then on the build host: python tools/summarize_profiles.py gpurun_out/r01d r01d   (writes profiles/)
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 60 --warmup 10"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_stats.log 2>&1
PMCBENCH="$BENCH --no-cpu --no-extra"     # the counter passes need the step kernels only
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMCBENCH > $OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMCBENCH > $OUT/bench_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/legs -- python $R/tools/profile_legs.py > $OUT/legs.log 2>&1
# matrix-core utilisation of the evaluation sweep (70,000^2 x 100), counters in their own pass
LEGS=eval70k timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python $R/tools/profile_legs.py > $OUT/legs_mfma.log 2>&1
# the other model families (TransE / TransH / TransD, BootEA_RotatE): kernel stats of a few epochs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/models -- python $R/tools/profile_models.py 15K TransE,TransH,TransD,BootEA_RotatE > $OUT/models.log 2>&1
python $R/bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
tail -1 $OUT/bench_stats.log
