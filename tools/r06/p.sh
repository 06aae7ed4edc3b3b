#!/bin/bash
# round 6, call p: one Philox call per lane in the sampler (a spare lane computes the round's side block)
set -u
python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "sampler or sample or negat" 2>&1 | tail -2
python tools/_exp/side_work.py 2>&1 | tail -4
