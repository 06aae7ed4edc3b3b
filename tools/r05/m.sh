#!/bin/bash
# round 5, call m: the driver-like bench line of the final tree with its bench_detail.json kept beside it
set -u
O=gpurun_out/r05m; mkdir -p $O
export TMPDIR=/tmp
( S=$(date +%s); timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err > $O/bench.out; echo "rc $? bench wall $(( $(date +%s) - S )) s" ) > $O/bench.log 2>&1
cp bench_detail.json $O/bench_detail.json 2>/dev/null
cat $O/bench.log; tail -1 $O/bench.out | cut -c1-400
