#!/bin/bash
# round 5, call e: big pipeline with lean bookkeeping + early fragment reads, bf16 sample strips; whole GPU suite
set -u
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
( OEA_BF16_BIG=1 timeout 600 python tools/_exp/eval_shapes.py "100,300,1200" 8 check 2>&1 | grep -v amdgpu.ids ) > $O/shapes.log 2>&1
tools/prof.sh trace r05e -- python tools/_exp/eval1200_trace.py 1200 2
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_gpu.log 2>&1
cat $O/shapes.log; head -8 $O/trace_stats.csv | cut -c1-70,200-330; tail -6 $O/pytest_gpu.log
