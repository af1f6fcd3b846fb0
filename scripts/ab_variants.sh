#!/bin/bash
# A/B of conv kernel scheduling variants (tuning aid): runs the conv micro-benchmark once per build.
for v in base; do
  if [ "$v" = base ]; then unset SACV_HIP_LIB; else export SACV_HIP_LIB=$PWD/gpurun_variants/libsacv_$v.so; fi
  echo "== $v"
  python scripts/bench_conv.py balanced_3x3_256 head3x3_5lvl l3_3x3_256 l3_1x1_256_1024 l2_1x1_512_128 2>&1 | grep -v amdgpu | cut -c1-96
done
