#!/bin/bash
# round 6, call z: the "certainly absent" bit array in front of the sampler's membership probes
python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "sampler or sample or negat" 2>&1 | tail -3
for F in 1 0; do echo "OEA_SAMPLER_FILTER=$F"; OEA_SAMPLER_FILTER=$F python tools/_exp/side_work.py 2>&1 | grep sampler; done
for F in 1 0 1 0; do OEA_SAMPLER_FILTER=$F python bench.py --steps 20 --warmup 5 --no-cpu --no-traffic --no-gnn --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('FILTER=$F', d['value'], d['ms_per_step'])"; done
