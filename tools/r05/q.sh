#!/bin/bash
# round 5, call q: the step / partition tests on the last build of triple_step.hip (oea_halo_plan added)
set -u
O=gpurun_out/r05q; mkdir -p $O
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
( timeout 100 python -m pytest tests/test_kernels_gpu.py -q -x -k "triple_step or pipelined or transh or transd or empty_shard" 2>&1 | tail -3 ) > $O/pytest_step.log 2>&1
( timeout 230 python -m pytest tests/test_partition_gpu.py -q -x 2>&1 | tail -3 ) > $O/pytest_partition.log 2>&1
cat $O/smoke.log $O/pytest_step.log $O/pytest_partition.log
