#!/bin/bash
# round 5, call i: bytes of the halo exchange at the 100K shape with 8 ranks (all on this box's GPU, host-staged collectives over gloo:
# the VOLUME is what is measured here, not the time), headline-only kernel trace
set -u
O=gpurun_out/r05i; mkdir -p $O
export TMPDIR=/tmp
( OEA_BENCH_ONE_GPU=1 OEA_BENCH_BACKEND=gloo OEA_BENCH_NO_SIDE=1 timeout 900 python bench.py --gpus 8 --steps 6 --warmup 2 --repeats 2 --no-extra --no-cpu --no-gnn --no-traffic > $O/halo8.out 2> $O/halo8.err; echo rc $? ) > $O/halo8.log 2>&1
cp bench_detail.json $O/halo8_detail.json 2>/dev/null
( OEA_BENCH_ONE_GPU=1 OEA_BENCH_BACKEND=gloo OEA_BENCH_NO_SIDE=1 timeout 900 python bench.py --gpus 8 --exchange step --steps 3 --warmup 1 --repeats 1 --no-extra --no-cpu --no-gnn --no-traffic > $O/dense8.out 2> $O/dense8.err; echo rc $? ) > $O/dense8.log 2>&1
tools/prof.sh trace r05i -- python bench.py --steps 20 --warmup 5 --no-cpu --no-traffic --no-gnn --no-extra
cat $O/halo8.log; tail -1 $O/halo8.out | cut -c1-1800; tail -3 $O/halo8.err | cut -c1-300; cat $O/dense8.log; tail -1 $O/dense8.out | cut -c1-1500; head -4 $O/trace_stats.csv | cut -c1-60,180-300
