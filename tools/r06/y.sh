#!/bin/bash
# round 6, call y: where the epoch sampler's time goes; one 12-byte store per negative
python tools/_exp/sampler_where.py 2>&1 | tail -4
python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "sampler or sample or negat" 2>&1 | tail -2
