#!/bin/bash
# Run ON the GPU box (through gpurun): the round-end sequence -- GPU test suite, smoke(), the driver-like bench line, the
# kernel trace of the same command -- into gpurun_out/$1 (default tag "round").
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r04_final'
# then on the build host copy what should be judged into profiles/ (profiles are committed, gpurun_out/ is scratch).
set -u
TAG=${1:-round}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
( S=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1; echo "bench wall $(( $(date +%s) - S )) s" ) > $O/bench_driver_like.log 2>&1
cp bench_detail.json $O/bench_detail.json 2>/dev/null      # (the trace run below writes its own bench_detail.json)
tools/prof.sh trace ${TAG}_trace -- timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-traffic --no-gnn --no-extra
tail -3 $O/pytest_gpu.log; cat $O/smoke.log; tail -1 $O/bench_driver_like.log
