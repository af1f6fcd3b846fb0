#!/usr/bin/env python3
"""Append (or refresh) the decode + NMS latency section of profiles/<tag>_summary.md from the files
scripts/decode_session.sh produced (copied to profiles/<tag>_decode_*):  python scripts/make_decode_summary.py r03"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = '## Decode + NMS latency'


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    prof = os.path.join(ROOT, 'profiles')
    md = ['', MARK + ' (SURVEY 8d secondary metric)', '',
          '`bash scripts/decode_session.sh`: `bench.py --model {retinanet,fcos}_decode --nms-type T --steps 200 --warmup 20`, one '
          'step = one decoder call on the head outputs of a batch of 2 (800x1333; 200 700 / 22 400 predictions x 80 classes '
          'per image, 1 in 29 above the threshold), results left on the device.  `GB/s` = every class score / regression read '
          'once, over the WHOLE call (scan + selection + NMS; the selection and NMS stages are latency-bound).', '',
          '| workload | ms / image | call us (median) | algorithmic MB | GB/s over the call | CPU oracle ms / image (1 thread) | identical |',
          '|---|---|---|---|---|---|---|']
    for path in sorted(glob.glob(os.path.join(prof, f'{tag}_decode_*.json'))):
        d = json.load(open(path))
        name = os.path.basename(path)[len(tag) + 8:-5]
        r, c = d['roofline'], d.get('cpu_baseline')
        md.append(f'| {name} | {d["value"]} | {d["ms_per_step_median"] * 1e3:.1f} | {r["algorithmic_mb_per_launch"]} | '
                  f'{r["achieved"]} | {c["value"] if c else "-"} | '
                  f'{("yes" if "identical to it: True" in c["sample"] else "1e-5 (gaussian: expf)") if c else "-"} |')
    for stem, title in (('retina_python_nms', 'RetinaDecoder python_nms'), ('fcos_soft_nms', 'FCOSDecoder soft_nms')):
        path = os.path.join(prof, f'{tag}_decode_{stem}_kernel_stats.csv')
        if not os.path.exists(path):
            continue
        md += ['', f'Kernel split, {title} (`rocprofv3 --kernel-trace --stats`, `{os.path.basename(path)}`):', '',
               '| kernel | calls | avg us | bound |', '|---|---|---|---|']
        for r in csv.DictReader(open(path)):
            k = r['Name'].split('(')[0].replace('void ', '')
            if not any(s in k for s in ('decode_scan', 'select_sort', 'nms')):
                continue
            avg = float(r['AverageNs']) / 1e3
            note = 'latency (one workgroup per image)'
            if 'decode_scan' in k:
                mb = 128.4 if 'retina' in stem else 14.5  # the class scores (+ FCOS centerness) of the batch
                note = f'HBM: {mb} MB read once = {mb / avg:.2f} TB/s = {mb / avg / 8:.2f} of 8 TB/s'
            if 'nms_mask' in k:
                note = 'latency / VALU (1000 x 16 waves over the chip)'
            md.append(f'| `{k}` | {r["Calls"]} | {avg:.1f} | {note} |')
    infer = sorted(glob.glob(os.path.join(prof, f'{tag}_infer_*.json')))
    if infer:
        md += ['', 'Evaluation passes (`bench.py --model {retinanet,fcos}_infer`: `model.eval()` forward with BatchNorm folded into the '
               'convolution epilogues and the im2col-free stem + the decoder on the device, batch of 2):', '',
               '| workload | ms / batch (median) | images/s |', '|---|---|---|']
        for path in infer:
            d = json.load(open(path))
            md.append(f'| {os.path.basename(path)[len(tag) + 7:-5]} ({d["metric"].split(" evaluation")[0].split("node ")[1]}, '
                      f'{d["metric"].split("NMS ")[1].split(" bs")[0]}) | {d["ms_per_step_median"]} | {d["value"]} |')
    spath = os.path.join(prof, f'{tag}_summary.md')
    text = open(spath).read()
    if MARK in text:
        text = text[:text.index(MARK)].rstrip('\n') + '\n'
    open(spath, 'w').write(text + '\n'.join(md) + '\n')
    print('\n'.join(md))


if __name__ == '__main__':
    main()
