#!/bin/bash
# Run ON the GPU box (through gpurun): GPU test suite + bench lines + kernel variants, logs into gpurun_out/$1.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a'
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 2500 $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err
for r in ; do
  OEA_APPLY_ROWS=$r timeout 300 python bench.py --no-cpu --no-extra > $OUT/bench_applyrows_$r.json 2> $OUT/bench_applyrows_$r.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = j["roofline"]
        print(f.split("/")[-1], "value %.1f M/s  ms/step %.4f  fwd %.2f us  apply %.2f us" % (j["value"] / 1e6, j["ms_per_step"], r["avg_kernel_us"], r["apply_rows_avg_us"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
