#!/bin/bash
# Same-session A/B of an environment switch: bash scripts/ab_env.sh "VAR=value" "retina fcos" 3
set -o pipefail
out=gpurun_out/ab_env
rm -rf $out; mkdir -p $out
declare -A ARGS=( [retina]="" [fcos]="--model fcos" [clsbf16]="--model resnet50_cls --dtype bf16" [clsf32]="--model resnet50_cls" [mae]="--model vit_mae" )
for tag in ${2:-retina}; do
  for rep in $(seq 1 ${3:-2}); do
    env $1 python bench.py ${ARGS[$tag]} --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $out/${tag}_A_$rep.log 2>&1
    python bench.py ${ARGS[$tag]} --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $out/${tag}_B_$rep.log 2>&1
  done
done
for f in $out/*.log; do echo "== $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1)"; done
