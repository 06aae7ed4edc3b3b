#!/bin/bash
# round 4, call f: 64-lane groups for the grouped step at ld <= 128 (experiment)
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python bench.py --no-gnn --no-cpu --no-traffic --steps 20 --warmup 5 2>&1 | tail -1 ) > $O/bench_g32.log 2>&1
( OEA_STEP_G64=1 timeout 300 python bench.py --no-gnn --no-cpu --no-traffic --steps 20 --warmup 5 2>&1 | tail -1 ) > $O/bench_g64.log 2>&1
( OEA_STEP_G64=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "triple_step" 2>&1 | tail -3 ) > $O/g64_tests.log 2>&1
python - <<'PY'
import json
for f in ("g32","g64"):
    try:
        j=json.loads([l for l in open("gpurun_out/r04f/bench_%s.log"%f) if l.startswith("{")][-1])
        r=j["roofline"]; x=j["extra"].get("shape_100k",{})
        print(f, "15K: value %.1f ms/step %.4f kernel %.2f apply %.2f"%(j["value"],j["ms_per_step"],r["avg_kernel_us"],r["apply_rows_avg_us"]))
        r=x["roofline"]; print(f, "100K: value %.1f ms/step %.4f kernel %.2f apply %.2f"%(x["value"],x["ms_per_step"],r["avg_kernel_us"],r["apply_rows_avg_us"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/g64_tests.log
