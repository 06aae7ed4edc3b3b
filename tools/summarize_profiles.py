#!/usr/bin/env python
"""gpurun_out/<tag> (written by tools/collect_profiles.sh) -> profiles/<name>_*.csv + profiles/traffic_triple_fwd_bwd.json.

HBM traffic per launch follows MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE are collected in
separate passes, both count KB, and on gfx950 FETCH_SIZE reports half of the bytes read: bytes = (2*FETCH + WRITE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def pmc_avg(d, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return {k: acc[k] / n[k] for k in acc}, n


def main():
    src, name = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    for leg, out in (("stats", "bench_kernel_stats"), ("stats100k", "bench100k_kernel_stats"), ("legs", "legs_kernel_stats"),
                     ("models", "models_kernel_stats"), ("knn_stats", "knn_kernel_stats"), ("csls", "eval_csls_kernel_stats")):
        fs = glob.glob(os.path.join(src, leg, "*", "*_kernel_stats.csv"))
        if fs:
            shutil.copy(fs[0], os.path.join(prof, "%s_%s.csv" % (name, out)))
    for fn in ("bench_unprofiled", "bench_driver_like"):
        up = os.path.join(src, fn + ".json")
        if os.path.exists(up):
            lines = [l for l in open(up) if l.startswith("{")]
            if lines:
                open(os.path.join(prof, "%s_%s.json" % (name, fn)), "w").write(lines[-1])
    # matrix-core counters of the evaluation sweep: raw per-kernel averages
    rows = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(src, "pmc_mfma", "*", "*_counter_collection.csv")):
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[(r["Kernel_Name"], r["Counter_Name"])] += 1
        for (k, c), v in acc.items():
            rows[k][c] = v / cnt[(k, c)]
            rows[k]["calls"] = cnt[(k, c)]
    if rows:
        names = sorted({c for d in rows.values() for c in d if c != "calls"})
        with open(os.path.join(prof, "%s_pmc_mfma_eval70k.csv" % name), "w") as f:
            f.write("kernel,calls," + ",".join(n + "_avg" for n in names) + "\n")
            for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
                f.write('"%s",%d,' % (k[:100], d["calls"]) + ",".join("%.0f" % d.get(n, 0.0) for n in names) + "\n")
    for log in ("bench_stats.log", "legs.log", "models.log", "knn_stats.log", "csls.log"):
        p = os.path.join(src, log)
        if os.path.exists(p):
            lines = [l for l in open(p) if not l.startswith("/opt/amdgpu")]
            note = "# stdout of the run UNDER rocprofv3 --kernel-trace: wall-clock figures are inflated by the tracer;\n" \
                   "# the kernel durations in the *_kernel_stats.csv next to this file are what DESIGN.md cites.\n"
            open(os.path.join(prof, "%s_%s" % (name, log.replace(".log", "_stdout.txt"))), "w").writelines([note] + lines[-40:])
    workloads = {"": ("EN-FR-15K-V1", 75, 5000, 10), "100k": ("EN-FR-100K-V1", 100, 20000, 10)}
    for suffix, tag in (("", "pmc_hbm_traffic"), ("100k", "pmc_hbm_traffic_100k"), ("knn", "pmc_hbm_traffic_knn_lists"),
                        ("knnstrip", "pmc_hbm_traffic_knn_strip"), ("knngen", "pmc_hbm_traffic_knn_general"),
                        ("eval70k", "pmc_hbm_traffic_eval70k")):
        fdir = os.path.join(src, "pmc_fetch" + suffix if suffix in ("", "100k") else suffix + "_fetch")
        wdir = os.path.join(src, "pmc_write" + suffix if suffix in ("", "100k") else suffix + "_write")
        if not os.path.isdir(fdir):
            continue
        fetch, n = pmc_avg(fdir, "FETCH_SIZE")
        write, _ = pmc_avg(wdir, "WRITE_SIZE")
        rows = []
        for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0)) * n[k]):
            rows.append((k[:90], n[k], fetch[k], write.get(k, 0.0), int((2 * fetch[k] + write.get(k, 0.0)) * 1024)))
        with open(os.path.join(prof, "%s_%s.csv" % (name, tag)), "w") as f:
            f.write("kernel,calls,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
            for r in rows:
                f.write('"%s",%d,%.1f,%.1f,%d\n' % r)
        if suffix in workloads:
            dom = [r for r in rows if "triple_grouped" in r[0]]
            if dom:
                r = dom[0]
                shape = workloads[suffix][0]
                # average durations of the two step kernels in the kernel trace of the same bench command
                dur = {}
                leg = "stats" if suffix == "" else "stats100k"
                for fcsv in glob.glob(os.path.join(src, leg, "*", "*_kernel_stats.csv")):
                    for row in csv.DictReader(open(fcsv)):
                        nm = row["Name"]
                        if r[0][:60] in nm:
                            dur["rocprof_avg_kernel_us"] = round(float(row["AverageNs"]) / 1e3, 2)
                            dur["rocprof_launches"] = int(row["Calls"])
                        want = "apply_rows<32, 3" if suffix == "" else "apply_rows<"
                        if want in nm and ("apply_rows<32, 3" in nm) == (suffix == ""):
                            dur["rocprof_apply_rows_avg_us"] = round(float(row["AverageNs"]) / 1e3, 2)
                json.dump({**dur, "kernel": r[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0],
                           "workload": list(workloads[suffix]),
                           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/%s_%s.csv" % (name, tag),
                           "FETCH_SIZE_KB": r[2], "WRITE_SIZE_KB": r[3],
                           "correction": "gfx950: FETCH_SIZE reads 1/2 of the bytes (MI355X_MICROARCH.md HBM section) -> 2*FETCH + WRITE",
                           "hbm_bytes_per_launch": r[4]}, open(os.path.join(prof, "traffic_%s.json" % shape), "w"), indent=1)
    print("wrote", sorted(os.listdir(prof)))


if __name__ == "__main__":
    main()
