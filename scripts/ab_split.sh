#!/bin/bash
# same-session A/B of the Winograd operand form on the headline step: bash scripts/ab_split.sh reps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${1:-2}); do
  for s in False True; do
    python3 bench.py --engine-set wino_split=$s --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('wino_split=$s', d['ms_per_step'], d['ms_per_step_median'])"
  done
done
