#!/bin/bash
# round 5, call o: the full-size parity test at AliNet's evaluation width
set -u
O=gpurun_out/r05o; mkdir -p $O
( timeout 500 python -m pytest tests/test_fullsize_gpu.py -q -x -k "alinet_width" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
cat $O/pytest.log
