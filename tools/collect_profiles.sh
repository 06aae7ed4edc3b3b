#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace stats of the default bench + the HBM-traffic PMC passes (FETCH_SIZE /
# WRITE_SIZE in their own runs, no other trace domains) for both bench shapes and for the neighbour search, into gpurun_out/$1.
#   gpurun -- 'bash tools/collect_profiles.sh r02'
# then on the build host: python tools/summarize_profiles.py gpurun_out/r02 r02   (writes profiles/)
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 60 --warmup 10 --repeats 10"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_stats.log 2>&1
PMC15="$BENCH --no-cpu --no-extra"     # the counter passes need the step kernels only
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMC15 > $OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMC15 > $OUT/bench_write.log 2>&1
PMC100="python $R/bench.py --shape EN-FR-100K-V1 --dim 100 --batch 20000 --eps 0.98 --steps 40 --warmup 10 --repeats 5 --no-cpu --no-extra"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats100k -- $PMC100 > $OUT/bench100k_stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch100k -- $PMC100 > $OUT/bench100k_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write100k -- $PMC100 > $OUT/bench100k_write.log 2>&1
python $R/bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err
if [ "${COLLECT:-all}" = "bench" ]; then exit 0; fi     # COLLECT=bench: the step kernels' passes only
# neighbour search at 100,000 x 100,000, k = 2,000: kernel stats + traffic of the strip-free path and of the strip path
KNN="python $R/tools/_exp/knn_time.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/knn_stats -- $KNN > $OUT/knn_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/knn_fetch -- $KNN > $OUT/knn_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/knn_write -- $KNN > $OUT/knn_write.log 2>&1
# the general (queries != candidates) list path: OEA_TOPK_SYM=0; the strip path's traffic is in profiles/r02_pmc_hbm_traffic_knn_strip.csv
OEA_TOPK_SYM=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/knngen_fetch -- $KNN > $OUT/knngen_fetch.log 2>&1
OEA_TOPK_SYM=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/knngen_write -- $KNN > $OUT/knngen_write.log 2>&1
# evaluation sweep at 70,000^2 x 100: traffic of the rank kernel
LEGS=eval70k timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/eval70k_fetch -- python $R/tools/profile_legs.py > $OUT/eval70k_fetch.log 2>&1
LEGS=eval70k timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/eval70k_write -- python $R/tools/profile_legs.py > $OUT/eval70k_write.log 2>&1
# evaluation (with / without CSLS at 10,500^2 and 70,000^2), GNN legs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/csls -- python $R/tools/_exp/csls_time.py > $OUT/csls.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/legs -- python $R/tools/profile_legs.py > $OUT/legs.log 2>&1
# matrix-core utilisation of the evaluation sweep (70,000^2 x 100), counters in their own pass
LEGS=eval70k timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python $R/tools/profile_legs.py > $OUT/legs_mfma.log 2>&1
grep -h kNN $OUT/knn_stats.log | head -4
tail -3 $OUT/csls.log
