#!/bin/bash
# sensitivity of the step to collective kernels sharing the chip, and what a CU reserve buys, on ONE GPU.
# A stand-in for RCCL (bench.py --emulate-rccl) runs R resident workgroups per exchange stage for the time an 8-rank ring
# all-reduce of that stage takes at 200 GB/s; the persistent grids plan for num_cus - reserve CUs during the backward pass.
set -o pipefail
OUT=gpurun_out/r4_cu_reserve
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # label, bench args
  python3 bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(label='$1', ms_per_step=d['ms_per_step'], ms_median=d['ms_per_step_median'])))"
}
for rep in 1 2; do
  run "no collective" ""
  for R in 8 16 32 64; do
    run "R=$R lds=16K reserve=0" "--emulate-rccl $R:16384"
    run "R=$R lds=16K reserve=$R" "--emulate-rccl $R:16384 --engine-set BWD_SHARE=(3,2,3,2,$R)"
  done
  run "R=32 lds=64K reserve=0" "--emulate-rccl 32:65536"
  run "R=32 lds=64K reserve=32" "--emulate-rccl 32:65536 --engine-set BWD_SHARE=(3,2,3,2,32)"
  run "R=0 reserve=16" "--engine-set BWD_SHARE=(3,2,3,2,16)"
  run "R=0 reserve=32" "--engine-set BWD_SHARE=(3,2,3,2,32)"
  run "R=32 lds=16K 100GB/s reserve=0" "--emulate-rccl 32:16384:100"
  run "R=32 lds=16K 100GB/s reserve=32" "--emulate-rccl 32:16384:100 --engine-set BWD_SHARE=(3,2,3,2,32)"
done | tee $OUT/sweep.jsonl
