#!/bin/bash
# round 4, call a: deterministic build + callback communicator + bench defaults
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_partition_gpu.py -x -q -s 2>&1 | tail -40 ) > $O/partition.log 2>&1
( OEA_STEP_DETERMINISTIC=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -30 ) > $O/det_kernels_models.log 2>&1
( timeout 300 python bench.py --no-gnn --no-cpu --steps 20 --warmup 5 2>&1 | tail -3 ) > $O/bench_fp32.log 2>&1
( OEA_STEP_DETERMINISTIC=1 timeout 300 python bench.py --no-gnn --no-cpu --steps 20 --warmup 5 2>&1 | tail -3 ) > $O/bench_det.log 2>&1
( timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -30 ) > $O/dist.log 2>&1
( OEA_STEP_DETERMINISTIC=1 timeout 600 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "not bench" 2>&1 | tail -30 ) > $O/dist_det.log 2>&1
tail -5 $O/partition.log $O/det_kernels_models.log $O/dist.log $O/dist_det.log
python - <<'PY'
import json
for f in ("fp32","det"):
    try:
        j=json.loads([l for l in open("gpurun_out/r04a/bench_%s.log"%f) if l.startswith("{")][-1])
        r=j["roofline"]; x=j["extra"].get("shape_100k",{})
        print(f, "15K: value %.1f ms/step %.4f kernel %.2f apply %.2f traffic %s"%(j["value"],j["ms_per_step"],r["avg_kernel_us"],r["apply_rows_avg_us"],r["traffic"]))
        if x: 
            r=x["roofline"]; print(f, "100K: value %.1f ms/step %.4f kernel %.2f apply %.2f traffic %s"%(x["value"],x["ms_per_step"],r["avg_kernel_us"],r["apply_rows_avg_us"],r["traffic"]))
    except Exception as e: print(f, "ERR", e)
PY
