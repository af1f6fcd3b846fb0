#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r4_fixup
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 scripts/conv_bits.py > $OUT/bits_new.txt 2> $OUT/bits_new.err || { tail -20 $OUT/bits_new.err; exit 1; }
SACV_HIP_LIB=$PWD/gpurun_variants/libsacv_r3.so timeout -k 10 600 python3 scripts/conv_bits.py > $OUT/bits_r3.txt 2> $OUT/bits_r3.err
if diff $OUT/bits_new.txt $OUT/bits_r3.txt > $OUT/bits.diff; then echo "conv bits: identical to round 3 ($(wc -l < $OUT/bits_new.txt) lines)"; else echo "conv bits DIFFER"; head -20 $OUT/bits.diff; fi
timeout -k 10 900 python3 -m pytest tests/test_gpu_splitk.py tests/test_gpu_kernels.py tests/test_gpu_blocks.py -x -q -m gpu -k "split or conv or blocks or stats or bn" > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
