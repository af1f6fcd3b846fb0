#!/bin/bash
# Same-session A/B of two builds of libsacv_hip.so: gpurun_variants/libsacv_prev.so (previous commit) vs the tree's.
#   bash scripts/ab_lib.sh "retina fcos clsbf16 mae" 3
set -o pipefail
out=gpurun_out/ab_lib
rm -rf $out; mkdir -p $out
PREV=$PWD/gpurun_variants/libsacv_prev.so
declare -A ARGS=( [retina]="" [fcos]="--model fcos" [clsbf16]="--model resnet50_cls --dtype bf16" [clsf32]="--model resnet50_cls" [mae]="--model vit_mae" )
for tag in ${1:-retina}; do
  for rep in $(seq 1 ${2:-2}); do
    SACV_HIP_LIB=$PREV python bench.py ${ARGS[$tag]} --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $out/${tag}_prev_$rep.log 2>&1
    python bench.py ${ARGS[$tag]} --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $out/${tag}_new_$rep.log 2>&1
  done
done
for f in $out/*.log; do echo "== $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1)"; done
