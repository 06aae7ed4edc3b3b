#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_models_gpu.py -m gpu -q -k "mapping or MTransE or translational" 2>&1 | tail -4
python tools/profile_models.py 15K MTransE,AlignE 2>&1 | grep epoch
python tools/profile_models.py 100K MTransE 2>&1 | grep epoch
