#!/bin/bash
# same-session A/B of operand forms on the headline step: bash scripts/ab_split.sh reps "wino_split=False conv_split=False" "wino_split=True conv_split=True" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
reps=$1; shift
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    sets=""; for kv in $cfg; do sets="$sets --engine-set $kv"; done
    python3 bench.py $sets $AB_ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['ms_per_step_median'])"
  done
done
