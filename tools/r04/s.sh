#!/bin/bash
# round 4, call s: stream form of the symmetric neighbour search: tests, timing with / without, kernel trace
O=gpurun_out/r04s; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_reference_fullsize.py -m gpu -q -x -k "neighbour or topk or knn" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
OEA_TOPK_DEBUG=1 timeout 300 python tools/_exp/knn_bf16.py > $O/knn_stream_on.log 2>&1
OEA_TOPK_STREAM=0 timeout 300 python tools/_exp/knn_bf16.py > $O/knn_stream_off.log 2>&1
tools/prof.sh trace r04s_trace -- python tools/_exp/knn_abl.py > $O/trace.log 2>&1
tail -6 $O/pytest.log; grep -v "rows redone" $O/knn_stream_on.log; grep "rows redone" $O/knn_stream_on.log | sort | uniq -c | sort -rn | head -5; cat $O/knn_stream_off.log
