#!/bin/bash
# Run ON the GPU box: attention / ADVICE tests + kernel traces of the GNN epochs at the 100K shapes.
set -u
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gnn_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "attention or adadelta or dense_sgd or alinet or dense_optim" > $OUT/pytest_attn.log 2>&1
echo "pytest attn rc=$?" >> $OUT/pytest_attn.log
tail -8 $OUT/pytest_attn.log
cd /tmp && export TMPDIR=/tmp
for M in AliNet RDGCN; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$M -- python $R/tools/profile_models.py 100K $M > $OUT/models_$M.log 2>&1
  grep -h "epoch" $OUT/models_$M.log | tail -2
  f=$(ls $OUT/trace_$M/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${M}_kernel_stats.csv && head -30 $f | cut -c1-200
  rm -rf $OUT/trace_$M
done
python $R/tools/profile_models.py 100K AliNet,RDGCN,GCN_Align > $OUT/models_100k.txt 2>&1
python $R/tools/profile_models.py 15K AliNet,RDGCN,GCN_Align > $OUT/models_15k.txt 2>&1
cat $OUT/models_100k.txt $OUT/models_15k.txt | grep -v amdgpu
