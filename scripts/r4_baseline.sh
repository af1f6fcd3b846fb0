#!/bin/bash
# Round-4 baseline session: bench twice + a time-ordered kernel trace of the step as it runs (two queues).
set -e -o pipefail
OUT=gpurun_out/r4_base
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $OUT/b1.json 2> $OUT/b1.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $OUT/b2.json 2> $OUT/b2.err
tail -c 400 $OUT/b2.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kto -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > $OUT/kto.log 2>&1
cp $(find $OUT/kto -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace_overlapped.csv
cp $(find $OUT/kto -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_overlapped.csv
rm -rf $OUT/kto
ls -la $OUT
