#!/bin/bash
# repeated runs of one workload: mean / median step time (a host stall shows as mean >> median)
for i in $(seq 1 ${2:-6}); do
  python bench.py $1 --steps 20 --warmup 5 --no-kernel-timer --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run',$i,'mean',d['ms_per_step'],'median',d['ms_per_step_median'])"
done
cat /proc/loadavg
