#!/bin/bash
# One profiling session on the GPU box (run through gpurun); raw outputs land in gpurun_out/prof_$TAG/ and are turned
# into profiles/$TAG_* by scripts/make_profile_summary.py in the build container.
#   gpurun --timeout 1200 -- 'bash scripts/profile_session.sh r04'
# (the decode + NMS latency lines and the evaluation-pass lines come from scripts/decode_session.sh and
#  `bench.py --model retinanet_infer`; scripts/make_decode_summary.py appends their section to the summary -- run it
#  AFTER make_profile_summary.py, which rewrites the file)
set -e -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SERIAL="--engine-set side_stream=False"
echo "[1] headline bench (un-profiled, CPU baseline, per-class table)"; date
python3 bench.py --steps 20 --warmup 5 --class-table $OUT/classes_retina.json > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
date; echo "[2] kernel trace of the headline bench: in stream order (side_stream=False: every kernel alone on the chip, the durations the roofline object is computed from) and as it runs (weight gradients overlapped on the side stream; time-ordered trace kept)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $SERIAL > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kto -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $OUT/kto.log 2>&1
cp $(find $OUT/kto -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_overlapped.csv
cp $(find $OUT/kto -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace_overlapped.csv
rm -rf $OUT/kto $OUT/kt
date; echo "[3] PMC passes (separate runs, stream order): FETCH_SIZE, WRITE_SIZE, MFMA busy"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer $SERIAL > $OUT/pmc_fetch.log 2>&1
cp $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $OUT/pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer $SERIAL > $OUT/pmc_write.log 2>&1
cp $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/pmc_write.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer $SERIAL > $OUT/pmc_mfma.log 2>&1
cp $(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/pmc_mfma.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
date; echo "[4] secondary lines: FCOS, S2, classification fp32 / bf16, MAE"
python3 bench.py --model fcos --steps 20 --warmup 5 --no-cpu-baseline --class-table $OUT/classes_fcos.json | tail -1 > $OUT/fcos.json
python3 bench.py --image-size 1333x1333 --steps 10 --warmup 3 --no-cpu-baseline | tail -1 > $OUT/s2.json
python3 bench.py --model resnet50_cls --steps 10 --warmup 3 | tail -1 > $OUT/cls_f32.json
python3 bench.py --model resnet50_cls --dtype bf16 --steps 10 --warmup 3 --class-table $OUT/classes_cls_bf16.json | tail -1 > $OUT/cls_bf16.json
python3 bench.py --model vit_mae --steps 10 --warmup 3 --class-table $OUT/classes_mae.json | tail -1 > $OUT/mae.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktc -- python3 bench.py --model resnet50_cls --dtype bf16 --steps 5 --warmup 2 --no-kernel-timer $SERIAL > $OUT/ktc.log 2>&1
cp $(find $OUT/ktc -name "*kernel_stats.csv" | head -1) $OUT/cls_bf16_kernel_stats.csv
rm -rf $OUT/ktc
ls -la $OUT
