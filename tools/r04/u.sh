#!/bin/bash
# round 4, call u: stream neighbour search with the redo list + overflow pool: tests, random / clustered timing, trained tables
O=gpurun_out/r04u; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_reference_fullsize.py -m gpu -q -x -k "neighbour or topk or knn" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
timeout 300 python tools/_exp/knn_bf16.py > $O/knn_stream_on.log 2>&1
OEA_TOPK_DEBUG=1 timeout 300 python tools/_exp/knn_bf16.py 2>&1 | grep "redone\|overflow" | sort | uniq -c | sort -rn | head -8 > $O/knn_stream_dbg.log
for st in 400 3000; do
  STEPS=$st timeout 300 python tools/_exp/knn_trained.py 2>&1 | grep -E "refresh" | tail -1
  STEPS=$st OEA_TOPK_DEBUG=1 timeout 300 python tools/_exp/knn_trained.py 2>&1 | grep -E "redone|overflow" | tail -2
  STEPS=$st OEA_TOPK_STREAM=0 timeout 300 python tools/_exp/knn_trained.py 2>&1 | grep -E "refresh" | tail -1 | sed 's/^/   stream off: /'
done > $O/trained.log 2>&1
tail -3 $O/pytest.log; cat $O/knn_stream_on.log $O/knn_stream_dbg.log $O/trained.log
