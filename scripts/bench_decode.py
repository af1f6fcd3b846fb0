"""Secondary measurement: decode + top-1000 + NMS latency per batch of 2 images at 800x1333,
device path vs the CPU oracle (= the reference's NumPy postprocess restated)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import golden_inputs as gi
from oracle import decode as odecode
from sacv_amd.detection import decode

H, W = gi.S1
for name, fcos in (('RetinaDecoder', False), ('FCOSDecoder', True)):
    preds = gi.synth_preds_decode(21, 2, H, W, fcos=fcos)
    dev = [[torch.tensor(x).cuda() for x in grp] for grp in preds]
    dec = decode.__dict__[name]()
    dec.device_call(dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dec.device_call(dev)
    e1.record()
    torch.cuda.synchronize()
    t_dev = e0.elapsed_time(e1) / 20
    t0 = time.perf_counter()
    out = dec(dev)
    t_e2e = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ref = odecode.fcos_decode(*preds) if fcos else odecode.retina_decode(*preds)
    t_cpu = (time.perf_counter() - t0) * 1e3
    print(f'{name}: device kernels {t_dev:.3f} ms/batch(2 imgs), incl. D2H of the [2,100] results {t_e2e:.3f} ms, '
          f'CPU oracle (NumPy, reference algorithm) {t_cpu:.1f} ms')
