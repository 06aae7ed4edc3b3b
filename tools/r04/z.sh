#!/bin/bash
# round 4, call z: kernel trace of the stream form at the 15K shape
export TMPDIR=/tmp
OEA_TOPK_SYM_MIN=8192 tools/prof.sh trace r04z15k_trace -- python tools/_exp/knn_15k.py
