#!/bin/bash
# same-session A/B of library builds on the headline step: bash scripts/ab_lib3.sh reps tree libsacv_X.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
reps=$1; shift
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    if [ "$lib" = "tree" ]; then unset SACV_HIP_LIB; else export SACV_HIP_LIB=$PWD/gpurun_variants/$lib; fi
    python3 bench.py $AB_ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['ms_per_step_median'])"
  done
done
