#!/bin/bash
# round 5, call a: the driver-like bench line (compact line + bench_detail.json), d = 500 / 1,200 parity cases, AliNet evaluation shape
set -u
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
( S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err > $O/bench.out; echo "rc $? bench wall $(( $(date +%s) - S )) s" ; tail -c 5000 $O/bench.out ) > $O/bench.log 2>&1
cp bench_detail.json $O/bench_detail.json 2>/dev/null
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16_prefilter or csls_means_one_sweep or rank_eval_bit_exact" 2>&1 | tail -15 ) > $O/pytest_d1200.log 2>&1
( timeout 600 python tools/_exp/alinet_eval.py 2>&1 | tail -40 ) > $O/alinet_eval.log 2>&1
tail -3 $O/bench.log | cut -c1-3000; cat $O/pytest_d1200.log | tail -8; cat $O/alinet_eval.log
