#!/bin/bash
# Decode + NMS latency (SURVEY.md 8d secondary metric): bench lines + kernel trace; outputs under gpurun_out/decode_$TAG/
#   gpurun --timeout 600 -- 'bash scripts/decode_session.sh r03'
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/decode_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nt in python_nms diou_python_nms torch_nms soft_nms soft_nms_gaussian; do
  python3 bench.py --model retinanet_decode --nms-type $nt --steps 200 --warmup 20 | tail -1 > $OUT/retina_$nt.json
  python3 bench.py --model fcos_decode --nms-type $nt --steps 200 --warmup 20 | tail -1 > $OUT/fcos_$nt.json
done
python3 bench.py --model retinanet_decode --image-size 1333x1333 --steps 200 --warmup 20 | tail -1 > $OUT/retina_s2_python_nms.json
# dense lists: every prediction above the threshold (min_score_threshold 0)
SACV_DECODE_MIN_SCORE=0 python3 bench.py --model retinanet_decode --steps 50 --warmup 5 --no-cpu-baseline | tail -1 > $OUT/retina_dense_python_nms.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --model retinanet_decode --steps 50 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/retina_python_nms_kernel_stats.csv
rm -rf $OUT/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --model fcos_decode --nms-type soft_nms --steps 50 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/fcos_soft_nms_kernel_stats.csv
rm -rf $OUT/kt
for f in $OUT/*.json; do echo $f; python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['unit'], d['roofline']['achieved'], d.get('cpu_baseline'))"; done
head -8 $OUT/retina_python_nms_kernel_stats.csv; head -8 $OUT/fcos_soft_nms_kernel_stats.csv
