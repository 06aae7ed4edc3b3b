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
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/bench_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/legs -- python $R/tools/profile_legs.py > $OUT/legs.log 2>&1
tail -1 $OUT/bench_stats.log
