#!/bin/bash
# A/B of the weight gradients' side stream (engine._on_side): same box, same session.  SACV_WGRAD_STREAM=0 = launch stream.
set -o pipefail
out=gpurun_out/ab_wgrad
mkdir -p $out
for rep in 1 2; do
  SACV_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $out/base_$rep.log 2>&1
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $out/side_$rep.log 2>&1
done
for m in fcos "resnet50_cls --dtype bf16" "resnet50_cls --dtype f32"; do
  tag=$(echo $m | tr -d ' -')
  SACV_WGRAD_STREAM=0 python bench.py --model $m --steps 20 --warmup 5 --no-kernel-timer > $out/${tag}_base.log 2>&1
  python bench.py --model $m --steps 20 --warmup 5 --no-kernel-timer > $out/${tag}_side.log 2>&1
done
for f in $out/*.log; do echo "== $f"; grep -o '"ms_per_step": [0-9.]*' $f | head -1; done
