#!/bin/bash
# round 5, call p: the halo plan against its oracle restatement
set -u
O=gpurun_out/r05p; mkdir -p $O
( timeout 300 python -m pytest tests/test_partition_gpu.py -q -x -k "halo_plan" 2>&1 | tail -12 ) > $O/pytest.log 2>&1
cat $O/pytest.log
