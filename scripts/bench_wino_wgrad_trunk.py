"""Weight gradient of the trunk's stride-1 3x3 convolutions at 2 x 800 x 1333: conv_tn (direct) against the Winograd
weight gradient fed by a transform of x made on the spot (wino_input + wino_dy + wino_wgrad + reduce)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import ops

def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for n, h, w, c in ((2, 200, 334, 64), (2, 100, 167, 128), (2, 50, 84, 256), (2, 25, 42, 512)):
    x = torch.randn(n, h, w, c, device='cuda')
    dy = torch.randn(n, h, w, c, device='cuda')
    dw = torch.empty(c, c, 3, 3, device='cuda')
    dw2 = torch.empty(c, c, 3, 3, device='cuda')
    direct = t(lambda: ops.conv_wgrad([dy], [x], dw, c, 3, 3, 1, 1))
    tin = t(lambda: ops.wino_input([x], tag='bench_v'))
    def wino():
        v = ops.wino_input([x], tag='bench_v')
        ops.wino_wgrad([dy], v, dw2, c)
    tw = t(wino)
    err = float((dw2 - dw).abs().max() / dw.abs().max())
    fl = 2.0 * n * h * w * c * c * 9
    print(f'{n}x{h}x{w} {c}->{c}: conv_tn {direct:6.1f} us ({fl / direct / 1e6:5.1f} TF)  winograd {tw:6.1f} us (input transform {tin:5.1f}) '
          f'rel diff {err:.1e}', flush=True)
