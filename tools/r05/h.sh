#!/bin/bash
# round 5, call h: the driver-like bench line of the current tree + its kernel trace
set -u
O=gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
( S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err > $O/bench.out; echo "rc $? bench wall $(( $(date +%s) - S )) s" ) > $O/bench.log 2>&1
cp bench_detail.json $O/bench_detail.json 2>/dev/null
tools/prof.sh trace r05h -- python bench.py --steps 20 --warmup 5 --no-cpu --no-traffic --no-gnn
cat $O/bench.log; tail -1 $O/bench.out
