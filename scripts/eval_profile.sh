#!/bin/bash
# Kernel trace of the evaluation path (eval-mode forward + decoder) at 2 x 800x1333:  gpurun -- 'bash scripts/eval_profile.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/evalfwd.py <<'PY'
import sys, time; sys.path.insert(0, '.')
import torch
from sacv_amd import synthetic as synth
from sacv_amd.detection import models, decode
net = models.__dict__['resnet50_retinanet']().cuda().eval()
dec = decode.__dict__['RetinaDecoder']()
x = torch.tensor(synth.images(0, 2, 800, 1333)).cuda()
with torch.no_grad():
    for _ in range(3):
        dec.device_call(net(x))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = dec.device_call(net(x))
    torch.cuda.synchronize()
print('eval forward + decode: %.3f ms per batch of 2' % ((time.perf_counter() - t0) / 10 * 1e3))
PY
python3 /tmp/evalfwd.py 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_eval -- python3 /tmp/evalfwd.py > gpurun_out/kt_eval.log 2>&1
f=$(find gpurun_out/kt_eval -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 13 / 1e6
print('kernel time per batch %.3f ms' % tot)
for r in rows[:22]:
    print('%-70s calls/batch %6.1f  ms/batch %.3f' % (r['Name'][:70], int(r['Calls']) / 13, float(r['TotalDurationNs']) / 13 / 1e6))
PY
rm -rf gpurun_out/kt_eval
