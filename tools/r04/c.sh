#!/bin/bash
# round 4, call c: partition tests (fp32: 2 epochs / fixed point: 5 epochs bitwise), sharded dW, bench with the GNN counter passes
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_partition_gpu.py -q -s 2>&1 | grep -v "^$" | tail -30 ) > $O/partition.log 2>&1
( timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -s -k "not bench" 2>&1 | grep -v "^$" | tail -20 ) > $O/dist.log 2>&1
( S=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_full.err | tail -3; echo "bench wall $(( $(date +%s) - S )) s" ) > $O/bench_full.log 2>&1
tail -8 $O/bench_full.err >> $O/bench_full.log
for f in partition dist; do echo "== $f"; tail -12 $O/$f.log; done
tail -3 $O/bench_full.log | cut -c1-300
python - <<'PY'
import json
def last(f):
    return json.loads([l for l in open(f) if l.startswith("{")][-1])
try:
    j=last("gpurun_out/r04c/bench_full.log")
    print("full: value %.1f ms/step %.4f"%(j["value"],j["ms_per_step"]), "roofline_eval", {k:j["roofline_eval"][k] for k in ("achieved","frac","csls10_frac_4N1N2d")})
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
