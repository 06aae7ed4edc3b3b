#!/bin/bash
# round 6, call e: who overlaps the step kernels?  full kernel trace of the headline workload, durations of triple_wave / apply_step_plan
# by what runs beside them on the side stream
set -u
O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
D=$(mktemp -d /tmp/oea_trace_XXXX)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --steps 80 --warmup 5 --repeats 6 --no-extra --no-gnn --no-cpu --no-traffic > $O/trace_stdout.log 2>&1
f=$(find $D -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/overlap.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = "wave" if "triple_wave" in name else "apply" if "apply_step_plan" in name else "sampler" if "sample_negatives" in name else \
        "plan" if ("plan_" in name or "ROCPRIM_400200" in name or "rocprim" in name.lower() and "400200" in name) else "shuffle" if ("rocprim" in name.lower() or "at::native" in name or "randperm" in name) else "other"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, name[:60]))
ev.sort()
side = [e for e in ev if e[2] in ("sampler", "plan", "shuffle", "other")]
import bisect
starts = [e[0] for e in side]
def beside(s, e):
    out = collections.Counter()
    i = bisect.bisect_left(starts, s - 2000000)
    while i < len(side) and side[i][0] < e:
        if side[i][1] > s:
            out[side[i][2]] += min(e, side[i][1]) - max(s, side[i][0])
        i += 1
    return out
for kind in ("wave", "apply"):
    alone, with_ = [], collections.defaultdict(list)
    for s, e, k, n in ev:
        if k != kind: continue
        b = beside(s, e)
        if not b: alone.append(e - s)
        else: with_[max(b, key=b.get)].append(e - s)
    import statistics as st
    print(kind, "alone: n=%d median %.1f us mean %.1f" % (len(alone), st.median(alone) / 1e3 if alone else 0, st.mean(alone) / 1e3 if alone else 0))
    for k2, v in with_.items():
        print(kind, "beside %-8s n=%d median %.1f us mean %.1f" % (k2, len(v), st.median(v) / 1e3, st.mean(v) / 1e3))
# gaps between consecutive main-stream kernels
main = [e for e in ev if e[2] in ("wave", "apply")]
gaps = [main[i + 1][0] - main[i][1] for i in range(len(main) - 1) if main[i + 1][0] - main[i][1] < 200000]
print("gaps between step kernels: median %.2f us mean %.2f us (n=%d)" % (st.median(gaps) / 1e3, st.mean(gaps) / 1e3, len(gaps)))
# main-stream idle time inside the timed regions: runs of step kernels (gaps < 150 us), span against the sum of durations
runs, cur = [], [main[0]]
for a, b in zip(main, main[1:]):
    if b[0] - a[1] < 150000: cur.append(b)
    else: runs.append(cur); cur = [b]
runs.append(cur)
for rr in runs:
    if len(rr) < 40: continue
    span, busy = rr[-1][1] - rr[0][0], sum(e - s for s, e, _, _ in rr)
    big = sorted(((b[0] - a[1]) / 1e3, a[2], b[2]) for a, b in zip(rr, rr[1:]) if b[0] - a[1] > 3000)[-6:]
    print("run of %d kernels: span %.2f ms, busy %.2f ms, idle %.1f us per step; largest gaps (us, after, before): %s" %
          (len(rr), span / 1e6, busy / 1e6, (span - busy) / 1e3 / (len(rr) / 2), [(round(g, 1), x, y) for g, x, y in big]))
    # the run in slices of 10 steps: mean kernel time per step, and the side-stream kernels active in the slice
    for i in range(0, len(rr), 20):
        sl = rr[i:i + 20]
        t0, t1 = sl[0][0], sl[-1][1]
        act = collections.Counter()
        for s_, e_, k_, n_ in side:
            if s_ < t1 and e_ > t0: act[k_] += (min(e_, t1) - max(s_, t0)) / 1e3
        print("    steps %3d-%3d: %.1f us per step (wave %.1f, apply %.1f)   side stream active (us): %s" %
              (i // 2, i // 2 + len(sl) // 2, (t1 - t0) / 1e3 / (len(sl) / 2), sum(e - s for s, e, k, _ in sl if k == "wave") / 1e3 / (len(sl) / 2),
               sum(e - s for s, e, k, _ in sl if k == "apply") / 1e3 / (len(sl) / 2), {k: round(v) for k, v in act.items()}))
tot = collections.Counter()
for s, e, k, n in ev: tot[k] += e - s
print("total kernel time by kind (ms):", {k: round(v / 1e6, 2) for k, v in tot.items()})
oth, cnt = collections.Counter(), collections.Counter()
for s, e, k, n in ev:
    if k in ("other", "shuffle", "plan", "sampler"): oth[n] += e - s; cnt[n] += 1
for n, v in oth.most_common(16): print("  %-62s %.2f ms  %d calls  %.1f us each" % (n, v / 1e6, cnt[n], v / cnt[n] / 1e3))
PY
cat $O/overlap.txt
rm -rf $D
