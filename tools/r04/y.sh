#!/bin/bash
# round 4, call y: B in registers for the CSLS means sweep and the stream neighbour sweep: tests + timings with / without
O=gpurun_out/r04y; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "csls or neighbour or topk" 2>&1 | tail -3 ) > $O/pytest.log 2>&1
for b in 1 0; do
  OEA_BF16_BREG=$b timeout 300 python tools/_exp/knn_bf16.py 2>&1 | grep "n=" | sed "s/^/breg=$b /"
  OEA_BF16_BREG=$b timeout 300 python tools/_exp/csls_trace.py 2>&1 | grep greedy | sed "s/^/breg=$b /"
done > $O/times.log 2>&1
tail -2 $O/pytest.log; cat $O/times.log
