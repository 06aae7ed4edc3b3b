#!/usr/bin/env python
"""gpurun_out/<runs of round 3> -> profiles/r03_* (the records DESIGN.md cites).  python tools/summarize_r03.py [collect tag]"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03z"


def copy(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))
        return True
    return False


def json_line(src, dst):
    if os.path.exists(src):
        lines = [l for l in open(src) if l.startswith("{")]
        if lines:
            open(os.path.join(P, dst), "w").write(lines[-1])


def text(lines, dst, header):
    keep = [l for l in lines if "amdgpu" not in l and not l.startswith(("/tmp/oea", "results output", "W2", "E2"))]
    open(os.path.join(P, dst), "w").write("# " + header + "\n" + "".join(keep))


z = os.path.join(G, tag)
copy(os.path.join(z, "bench_kernel_stats.csv"), "r03_bench_kernel_stats.csv")
for m in ("AliNet", "RDGCN", "GCN_Align"):
    copy(os.path.join(z, "%s_100k_kernel_stats.csv" % m), "r03_%s_100k_kernel_stats.csv" % m)
json_line(os.path.join(z, "bench_unprofiled.json"), "r03_bench_unprofiled.json")
json_line(os.path.join(z, "bench_driver_like.json"), "r03_bench_driver_like.json")
if os.path.exists(os.path.join(z, "models_epochs.txt")):
    text([l for l in open(os.path.join(z, "models_epochs.txt")) if " epoch " in l], "r03_models_epochs.txt",
         "tools/profile_models.py, device-synchronised wall clock per epoch (5 or 20 timed epochs after a warm-up run)")
if os.path.exists(os.path.join(z, "bench_stats.log")):
    text(open(os.path.join(z, "bench_stats.log")).readlines()[-6:], "r03_bench_stats_stdout.txt",
         "stdout of `bench.py --steps 60 --warmup 10 --repeats 10 --no-traffic --no-cpu` UNDER rocprofv3 --kernel-trace: wall-clock figures are "
         "inflated by the tracer; the kernel durations in r03_bench_kernel_stats.csv are what DESIGN.md cites")
copy(os.path.join(G, "r03j", "gnn_kernel_stats.csv"), "r03_gnn_legs_kernel_stats.csv")
for src, dst, hdr in (("r03e/knn_ccap16.log", None, None),):
    pass
lines = []
for c in (16, 8, 6):
    f = os.path.join(G, "r03e", "knn_ccap%d.log" % c)
    if os.path.exists(f):
        lines += ["OEA_TOPK_CCAP=%d  %s" % (c, l) for l in open(f) if l.startswith("kNN")]
if lines:
    text(lines, "r03_knn_candidate_segment_capacity.txt", "tools/_exp/knn_time.py: symmetric neighbour search, candidate-side segment capacity 16 / 8 / 6 entries")
for name, dst, hdr in (("/tmp/r03n.out", "r03_eval_workgroup_target.txt", "greedy_alignment_device end to end (ms per call) against the workgroup target of the rank sweep (OEA_RANK_WGS)"),
                       ("/tmp/r03k2.out", "r03_spmm_group_width.txt", "bench.gnn_legs with OEA_SPMM_G = lanes per row of the aggregate at 64 < ld <= 128")):
    if os.path.exists(name):
        text([l for l in open(name) if l.startswith(("WGS=", "G="))], dst, hdr)
f = os.path.join(G, "r03u", "rdgcn_epochs.txt")
if os.path.exists(f):
    text([l for l in open(f) if "epochs:" in l or "get_neg" in l or "negative set" in l], "r03_rdgcn_mining.txt",
         "RDGCN: ms per epoch over 20 epochs (fused GCN block on / off), and one hard-negative mining call (20,000 x 200,000 x 300 at the "
         "100K shape) through the all-pairs fp64 strip, the fp32 pre-filter and the certified 16-bit grid pre-filter (default), each "
         "followed by the exact fp64 re-rank; 'uncertified' = seeds the certificate sent to the all-pairs path")
for src, dst, hdr, keep in (
        ("r03y3/gemm_tn.txt", "r03_gemm_tn.txt", "tools/_exp/gemm_tn_time.py: dW = X^T dY at E = 200,000, library (torch.mm) vs oea_gemm_tn_f32", "M="),
        ("r03ab/l1_eval.txt", "r03_l1_grid_eval.txt", "tools/_exp/l1_eval_time.py: manhattan evaluation 70,000^2 x 300, all-pairs fp64 vs 16-bit grid distances + "
         "exact decisions (ranks and nearest candidates compared)", ""),
        ("r03af/select_phases.txt", "r03_select_phases.txt", "tools/r03_af.sh: symmetric neighbour search 100,000^2, k = 2,000, ms per call UNDER the tracer with "
         "list_select_kernel leaving after phase N (OEA_TOPK_SELECT_STOP; 0 = complete)", "STOP"),
        ("r03ag4/knn.txt", "r03_knn_times.txt", "tools/_exp/knn_time.py (random unit rows) and knn_trained.py (tables after 400 training steps)", ""),
        ("r03ae/csls_time.txt", "r03_csls_eval.txt", "tools/_exp/csls_time.py: greedy alignment with and without CSLS", "eval")):
    f = os.path.join(G, src)
    if os.path.exists(f):
        text([l for l in open(f) if keep in l], dst, hdr)
copy(os.path.join(G, "r03w", "MTransE_15k_kernel_stats.csv"), "r03_MTransE_15k_kernel_stats.csv")
copy(os.path.join(G, "r03ak", "knn15_kernel_stats.csv"), "r03_knn15k_kernel_stats.csv")
f = os.path.join(G, "r03al", "row_select_phases.txt")
if os.path.exists(f):
    text(open(f).readlines(), "r03_row_select_phases.txt", "15,000^2, k = 1,499 neighbour search, ms per call with row_select_kernel leaving after "
         "phase N (OEA_TOPK_SELECT_STOP: 1 = histogram + bucket, 2 = candidate pass, 3 = ranking; 0 = complete)")
lines = []
for i in (1, 2, 3, 4):
    for tag in ("r03an", "r03an2"):
        f = os.path.join(G, tag, "pytest_%d.log" % i)
        if os.path.exists(f):
            lines += ["%s run %d: %s" % ("list(set) order" if tag == "r03an" else "sorted order  ", i, l) for l in open(f) if "AliNet two ranks" in l or "first batch" in l]
if lines:
    text(lines, "r03_alinet_two_ranks.txt", "tests/test_dist_gpu.py::test_two_ranks_reproduce_single_process, four runs each with AliNet's triple list as "
         "list(set) (hash-seed dependent) and sorted")
copy(os.path.join(G, "r03ad", "l1_eval_kernel_stats.csv"), "r03_l1_grid_eval_kernel_stats.csv")
copy(os.path.join(G, "r03ad", "csls70k_kernel_stats.csv"), "r03_csls70k_kernel_stats.csv")
f = os.path.join(G, "r03g", "determinism.txt")
if os.path.exists(f):
    text(open(f).readlines(), "r03_determinism.txt", "tools/_exp/alinet_determinism.py: two single-process runs from the same seeds")
print(sorted(x for x in os.listdir(P) if x.startswith("r03")))
