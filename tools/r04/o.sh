#!/bin/bash
# round 4, call o: kernel trace of the CSLS evaluation at 70,000^2 (product path)
O=gpurun_out/r04o; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/_exp/csls_trace.py > $O/csls.log 2>&1
tools/prof.sh trace r04o_trace -- python tools/_exp/csls_trace.py
cat $O/csls.log
