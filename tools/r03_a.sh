#!/bin/bash
# Run ON the GPU box: round-3 first check -- deterministic attention + sharded 'runs' grouping + the new bench line.
set -u
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gnn_gpu.py tests/test_graph_golden.py tests/test_models_100k_gpu.py -m gpu -q -x > $OUT/pytest_attn.log 2>&1
echo "pytest attn rc=$?" >> $OUT/pytest_attn.log
tail -15 $OUT/pytest_attn.log
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -x > $OUT/pytest_dist.log 2>&1
echo "pytest dist rc=$?" >> $OUT/pytest_dist.log
tail -15 $OUT/pytest_dist.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err ) 2> $OUT/bench_time.txt
tail -c 6000 $OUT/bench_driver_like.json
tail -5 $OUT/bench_driver_like.err
cat $OUT/bench_time.txt
