#!/bin/bash
# Run ON the GPU box: the round-3 records -- kernel-trace stats of the default bench command, the bench lines themselves (with the
# in-run FETCH_SIZE / WRITE_SIZE passes), kernel-trace stats of the GNN epochs.   gpurun -- 'bash tools/collect_r03.sh r03z'
set -u
TAG=${1:-r03z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the traced run leaves out the in-run counter passes (a rocprofv3 inside a rocprofv3) and the CPU leg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 60 --warmup 10 --repeats 10 --no-traffic --no-cpu > $OUT/bench_stats.log 2>&1
f=$(ls $OUT/stats/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/bench_kernel_stats.csv
rm -rf $OUT/stats
for M in AliNet RDGCN GCN_Align; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$M -- python $R/tools/profile_models.py 100K $M > $OUT/models_$M.log 2>&1
  f=$(ls $OUT/trace_$M/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${M}_100k_kernel_stats.csv
  rm -rf $OUT/trace_$M
done
cd $R
python bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err
python tools/profile_models.py 100K GCN_Align,AliNet,RDGCN,AlignE,BootEA,MTransE > $OUT/models_epochs.txt 2>&1
python tools/profile_models.py 15K GCN_Align,AliNet,RDGCN,AlignE,BootEA,MTransE >> $OUT/models_epochs.txt 2>&1
grep epoch $OUT/models_epochs.txt
tail -c 600 $OUT/bench_driver_like.json
