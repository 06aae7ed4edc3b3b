#!/bin/bash
# round 6, call i: attention backward with 16 lanes per edge; 'reorder' on the segment kernels
set -u
python -m pytest tests/test_gnn_gpu.py tests/test_graph_golden.py tests/test_models_100k_gpu.py -q -m gpu 2>&1 | tail -5
