#!/bin/bash
# round 6, call h: the side stream's per-epoch work after the keyed-permutation layout and the scalar batch lookup in the sampler;
# sampler / layout / plan tests; the driver's protocol twice
set -u
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "sampler or epoch_layout or step_plan or planned" 2>&1 | tail -3
python tools/_exp/side_work.py 2>&1 | tail -4
for i in 1 2; do python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('extra',{}).get('regions'))"; done
