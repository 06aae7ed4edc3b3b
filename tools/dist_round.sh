#!/bin/bash
set -u
TAG=${1:-dist}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_partition_gpu.py tests/test_dist_gpu.py -m gpu -q -s > $OUT/pytest_dist.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_dist.log
grep -E "deviation|exchange bytes|passed|failed|FAILED|Error" $OUT/pytest_dist.log | head -40
# 2 ranks on the one GPU through the bench (gloo): the partitioned exchange end to end
OEA_BENCH_ONE_GPU=1 OEA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --repeats 5 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err
tail -c 1500 $OUT/bench_2ranks_gloo.json
