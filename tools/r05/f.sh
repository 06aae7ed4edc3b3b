#!/bin/bash
# round 5, call f: boundary-row (halo) exchange -- 2 / 4 ranks through callbacks vs the single-process job, bench wiring with 2 ranks
set -u
O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_partition_gpu.py -q -x -s -k "boundary_row or one_call_partitioned" 2>&1 | tail -40 ) > $O/pytest_halo.log 2>&1
( timeout 1200 python -m pytest tests/test_dist_gpu.py -q -x -k "bench_two_ranks" 2>&1 | tail -15 ) > $O/pytest_bench2.log 2>&1
cat $O/pytest_halo.log; cat $O/pytest_bench2.log
