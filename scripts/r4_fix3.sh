#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r4_fix3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 scripts/conv_bits.py > $OUT/bits_new.txt 2> $OUT/bits_new.err || { tail -20 $OUT/bits_new.err; exit 1; }
diff $OUT/bits_new.txt gpurun_out/r4_fixup/bits_r3.txt > $OUT/bits.diff && echo "conv bits identical to round 3" || { echo "conv bits DIFFER"; head $OUT/bits.diff; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_splitk.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for rep in 1 2 3; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['ms_per_step_median'])"
SACV_HIP_LIB=$PWD/gpurun_variants/libsacv_r3.so python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('r3lib', d['ms_per_step'], d['ms_per_step_median'])"
done
