#!/bin/bash
# One parameterised profiling runner for the GPU box (replaces the one-off r03_*.sh launchers).
#   tools/prof.sh trace <tag> -- <command...>              kernel trace + per-kernel stats  -> gpurun_out/<tag>/trace_stats.csv
#   tools/prof.sh pmc <tag> "<CTR1 CTR2 ...>" -- <command...>   one counter pass (ONLY --kernel-trace beside --pmc) -> gpurun_out/<tag>/pmc_<n>.csv
#   tools/prof.sh list <tag>                               available counters -> gpurun_out/<tag>/counters.txt
# Several pmc calls with the same tag append pmc_1.csv, pmc_2.csv ...; every csv is "kernel,launches,<avg per launch of each counter>".
set -u
mode=$1; tag=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
ROCPROF=${ROCPROF:-$(command -v rocprofv3 || echo /opt/rocm/bin/rocprofv3)}
case $mode in
  list)
    $ROCPROF -L > $O/counters.txt 2>&1 || $ROCPROF --list-avail > $O/counters.txt 2>&1
    ;;
  trace)
    [ "$1" = "--" ] && shift
    D=$(mktemp -d /tmp/oea_trace_XXXX)
    timeout 900 $ROCPROF --kernel-trace --stats --output-format csv -d $D -- "$@" > $O/trace_stdout.log 2>&1
    f=$(find $D -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $O/trace_stats.csv
    rm -rf $D
    ;;
  pmc)
    ctrs=$1; shift
    [ "$1" = "--" ] && shift
    D=$(mktemp -d /tmp/oea_pmc_XXXX)
    n=$(ls $O/pmc_*.csv 2>/dev/null | wc -l); n=$((n + 1))
    timeout 600 $ROCPROF --kernel-trace --pmc $ctrs --output-format csv -d $D -- "$@" > $O/pmc_${n}_stdout.log 2>&1
    python - "$D" "$O/pmc_$n.csv" <<'PY'
import collections, csv, glob, os, sys
acc, cnt, names = collections.defaultdict(float), collections.Counter(), set()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        cnt[(r["Kernel_Name"], r["Counter_Name"])] += 1
        names.add(r["Counter_Name"])
names = sorted(names)
kernels = sorted({k for k, _ in acc}, key=lambda k: -max(acc[(k, c)] for c in names))
with open(sys.argv[2], "w") as out:
    out.write("kernel,launches," + ",".join(names) + "\n")
    for k in kernels:
        n = max(cnt[(k, c)] for c in names)
        out.write('"%s",%d,' % (k[:110], n) + ",".join("%.0f" % (acc[(k, c)] / max(cnt[(k, c)], 1)) for c in names) + "\n")
PY
    rm -rf $D
    ;;
esac
