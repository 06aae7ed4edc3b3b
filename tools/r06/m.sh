#!/bin/bash
# round 6, call m: the general (queries != candidates) neighbour search on the bf16 split
set -u
python -m pytest tests/test_kernels_gpu.py tests/test_dist_gpu.py tests/test_partition_gpu.py -x -q -m gpu -k "general_path or two_ranks or refresh or partition" 2>&1 | tail -3
