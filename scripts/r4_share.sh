#!/bin/bash
# Same-session sweep of the backward pass's launch share (slots per CU: nt64,nt128,wino,wgrad)
set -o pipefail
OUT=gpurun_out/r4_share
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "" "2,1,2,1" "2,2,2,1" "3,2,2,1" "2,1,3,1" "2,1,2,2" "3,2,3,1"; do
    SACV_BWD_SHARE=$cfg python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share [$cfg]', d['ms_per_step'], d['ms_per_step_median'])"
  done
done | tee $OUT/sweep.txt
