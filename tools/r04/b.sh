#!/bin/bash
# round 4, call b: partition tests (fixed), manhattan + CSLS on the grid, bench with the GNN counter passes, TF1 stand-in
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_partition_gpu.py -x -q -s 2>&1 | tail -30 ) > $O/partition.log 2>&1
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "manhattan" 2>&1 | tail -30 ) > $O/manhattan.log 2>&1
python tests/golden/make_tf1_standin.py /tmp/tf1_standin.npz > $O/tf1.log 2>&1
( OEA_TF1_GOLDEN=/tmp/tf1_standin.npz timeout 300 python -m pytest tests/test_tf1_golden.py -q 2>&1 | tail -15 ) >> $O/tf1.log 2>&1
( OEA_STEP_DETERMINISTIC=1 timeout 300 python bench.py --no-gnn --no-cpu --steps 20 --warmup 5 2>&1 | tail -3 ) > $O/bench_det.log 2>&1
( /usr/bin/time -v timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_full.err | tail -3 ) > $O/bench_full.log 2>&1
tail -8 $O/bench_full.err >> $O/bench_full.log
for f in partition manhattan tf1; do echo "== $f"; tail -6 $O/$f.log; done
python - <<'PY'
import json
def last(f):
    return json.loads([l for l in open(f) if l.startswith("{")][-1])
try:
    j=last("gpurun_out/r04b/bench_det.log"); r=j["roofline"]; x=j["extra"]["shape_100k"]
    print("det 15K: value %.1f ms/step %.4f kernel %.2f apply %.2f"%(j["value"],j["ms_per_step"],r["avg_kernel_us"],r["apply_rows_avg_us"]))
    r=x["roofline"]; print("det 100K: value %.1f ms/step %.4f kernel %.2f apply %.2f"%(x["value"],x["ms_per_step"],r["avg_kernel_us"],r["apply_rows_avg_us"]))
except Exception as e: print("det ERR", e)
try:
    j=last("gpurun_out/r04b/bench_full.log")
    print("full: value %.1f ms/step %.4f"%(j["value"],j["ms_per_step"]), "roofline_eval", j.get("roofline_eval"))
    g=j["extra"]["gnn"]
    if "error" in g: print("GNN ERROR", g)
    else:
        print("gcn spmm", {k:g["gcn_align_se_epoch_DW15K"]["roofline"].get(k) for k in ("frac","hbm_frac","traffic","traffic_source")})
        a=g["alinet_EN-DE-100K"]
        print("alinet epoch", a["ms_per_epoch"], "row", a["ms_per_epoch_grouping_row"])
        print("attn runs", {k:a["roofline"].get(k) for k in ("ms","frac","hbm_frac","traffic")})
        print("attn row", a["grouping_row"]["attention_fwd_ms"], a["grouping_row"]["attention_bwd_ms"], {k:a["grouping_row"]["roofline"].get(k) for k in ("ms","frac","hbm_frac","traffic")})
        print("1hop", {k:a["roofline_1hop_aggregate"].get(k) for k in ("ms","frac","hbm_frac","traffic")})
        r=g["rdgcn_eval_70000x300"]; print("rdgcn eval", {k:r[k] for k in r if k.endswith("_ms")})
    s=j["extra"]["shape_100k"]
    print("csls hbm", {k:v for k,v in s.get("csls_eval_hbm",{}).items() if k!="kernels"})
    print("knn hbm", {k:v for k,v in s.get("neighbour_search_hbm",{}).items() if k!="kernels"})
    print("knn kernels", s.get("neighbour_search_hbm",{}).get("kernels"))
except Exception as e: print("full ERR", repr(e))
PY
