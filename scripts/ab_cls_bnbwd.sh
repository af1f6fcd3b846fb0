#!/bin/bash
# A/B kernel traces of the bf16 classifier with / without BatchNorm's backward first stage in the data-gradient epilogues
#   gpurun --timeout 400 -- 'bash scripts/ab_cls_bnbwd.sh'
set -e -o pipefail
OUT=gpurun_out/ab_bnbwd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in True False; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$f -- python3 bench.py --model resnet50_cls --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --engine-set side_stream=False --engine-set fuse_bn_bwd_amp=$f > $OUT/kt_$f.log 2>&1
  cp $(find $OUT/kt_$f -name "*kernel_stats.csv" | head -1) $OUT/stats_$f.csv
  rm -rf $OUT/kt_$f
done
