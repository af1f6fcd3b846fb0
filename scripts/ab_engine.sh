#!/bin/bash
# Same-session A/B of executor settings: bash scripts/ab_engine.sh reps "<bench args A>" "<bench args B>" ...
reps=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    python3 bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['ms_per_step'], d['ms_per_step_median'])"
  done
done
