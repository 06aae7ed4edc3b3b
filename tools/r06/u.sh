#!/bin/bash
# round 6, call u: the query operand in registers for the general search too
for N in 1 0; do OEA_TOPK_STREAM_NCH=$N python tools/_exp/knn_asym.py 2>&1 | grep kNN | sed "s/^/NCH=$N  /"; done
python -m pytest tests/test_fullsize_gpu.py tests/test_reference_fullsize.py tests/test_kernels_gpu.py -x -q -m gpu -k "neighbour or knn or topk" 2>&1 | tail -3
