#!/bin/bash
# round 4, call k: driver-like bench line with the bf16 evaluation prefilter as the product path
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
( S=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1; echo "bench wall $(( $(date +%s) - S )) s" ) > $O/bench_driver_like.log 2>&1
tail -c 1500 $O/bench_driver_like.log; tail -5 $O/bench.err
