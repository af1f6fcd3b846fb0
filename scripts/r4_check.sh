#!/bin/bash
# Quick GPU check of a scheduling / kernel change: selected tests, bench twice, time-ordered kernel trace.
#   gpurun -- 'bash scripts/r4_check.sh <tag> "<pytest selection>"'
set -e -o pipefail
TAG=${1:-chk}
SEL=${2:-tests/test_gpu_streams.py tests/test_gpu_dp.py tests/test_gpu_trainer.py}
OUT=gpurun_out/r4_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$SEL" != "none" ]; then
  timeout -k 10 900 python3 -m pytest $SEL -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
  tail -3 $OUT/tests.log
fi
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $OUT/b1.json 2> $OUT/b1.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $OUT/b2.json 2> $OUT/b2.err
python3 -c "
import json,sys
for f in ('$OUT/b1.json','$OUT/b2.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['ms_per_step_median'], d['loss'])
"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kto -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > $OUT/kto.log 2>&1
cp $(find $OUT/kto -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace_overlapped.csv
cp $(find $OUT/kto -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_overlapped.csv
rm -rf $OUT/kto
