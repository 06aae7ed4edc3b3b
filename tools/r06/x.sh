#!/bin/bash
# round 6, call x: bench with the kernel-timing regions in the middle of the repeats; the 2-rank bench test
python bench.py --steps 20 --warmup 5 --no-cpu --no-gnn --no-extra 2>/dev/null | tail -1 | cut -c1-1200
python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "bench_two_ranks" 2>&1 | tail -2
