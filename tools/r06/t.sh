#!/bin/bash
# round 6, call t: list_select_kernel at five workgroups per CU (LDS arrays aliased, 96 registers): tests
python -m pytest tests/test_fullsize_gpu.py tests/test_reference_fullsize.py tests/test_kernels_gpu.py -x -q -m gpu -k "neighbour or knn or topk or csls" 2>&1 | tail -3
