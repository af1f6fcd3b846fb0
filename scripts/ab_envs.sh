#!/bin/bash
# Same-session sweep of environment settings: bash scripts/ab_envs.sh "<bench args>" reps "ENV1=a" "ENV1=b ENV2=c" ...
args=$1; reps=$2; shift 2
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    env $cfg python bench.py $args --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['ms_per_step_median'])"
  done
done
