#!/bin/bash
# Same-session A/B of library builds: bash scripts/ab_lib2.sh "<bench args>" reps libA.so libB.so ...  ("tree" = the tree's)
args=$1; reps=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    if [ "$lib" = "tree" ]; then unset SACV_HIP_LIB; else export SACV_HIP_LIB=$PWD/gpurun_variants/$lib; fi
    python3 bench.py $args --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['ms_per_step_median'])"
  done
done
