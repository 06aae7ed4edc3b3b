#!/bin/bash
# where the list select spends its time: the kernel leaves after phase N (OEA_TOPK_SELECT_STOP), kernel-trace averages
set -u
TAG=${1:-r03af}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in 0 1 2 3 4 5; do
  KNN_QUICK=1 OEA_TOPK_SELECT_STOP=$S timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr$S -- python $R/tools/_exp/knn_time.py > $OUT/log$S.txt 2>&1
  f=$(ls $OUT/tr$S/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 -c "
import csv,sys
rows={r['Name'].split('(')[0].split('::')[-1]:r for r in csv.DictReader(open('$f'))}
def g(n):
    r=[v for k,v in rows.items() if n in k]
    return '%s %.1f us x %s'%(n, float(r[0]['AverageNs'])/1e3, r[0]['Calls']) if r else n+' -'
print('STOP=$S', g('list_select_kernel'), '|', g('topk_append_sym_kernel'), '|', g('kth_value_kernel'), '|', open('$OUT/log$S.txt').read().split('kNN')[1].split('ms')[0].split(':')[-1].strip(), 'ms per call')
" | tee -a $OUT/select_phases.txt
  rm -rf $OUT/tr$S
done
