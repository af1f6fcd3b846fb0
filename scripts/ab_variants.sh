#!/bin/bash
# A/B of conv kernel scheduling variants (tuning aid): runs the conv micro-benchmark once per build.
# usage: scripts/ab_variants.sh "<variant names>" <bench_conv case names...>
vs=$1; shift
for v in base $vs; do
  if [ "$v" = base ]; then unset SACV_HIP_LIB; else export SACV_HIP_LIB=$PWD/gpurun_variants/libsacv_$v.so; fi
  echo "== $v"
  python scripts/bench_conv.py "$@" 2>&1 | grep -v amdgpu | cut -c1-100
done
