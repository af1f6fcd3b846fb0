"""conv_tn tile shape / split count on the trunk's 1x1 (and 3x3) weight gradients at 2 x 800 x 1333 (experiment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import ops
from sacv_amd._lib import lib, check, iarr, parr

def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def wgrad(dy, x, dw, cin, k, chunk, hint):
    nb, ho, wo, cout = dy.shape
    nbytes = lib.sacv_conv2d_wgrad_workspace_bytes(1, iarr([nb]), iarr([ho]), iarr([wo]), cin, cout, k, k, chunk)
    ws = ops.workspace(nbytes, dy.device, 'wgrad')
    check(lib.sacv_conv2d_wgrad(1, parr([dy.data_ptr()]), parr([x.data_ptr()]), iarr([nb]), iarr([x.shape[1]]),
                                iarr([x.shape[2]]), iarr([ho]), iarr([wo]), dw.data_ptr(), ws.data_ptr(), cin, cin, cout,
                                k, k, 1, k // 2, chunk, 0, 0, (hint << 8), ops._stream()), 'wgrad')

CASES = [(2, 50, 84, 256, 1024, 1), (2, 50, 84, 1024, 256, 1), (2, 25, 42, 512, 2048, 1), (2, 25, 42, 2048, 512, 1),
         (2, 100, 167, 128, 512, 1), (2, 100, 167, 512, 128, 1), (2, 200, 334, 64, 256, 1), (2, 200, 334, 256, 64, 1),
         (2, 50, 84, 256, 256, 3), (2, 25, 42, 512, 512, 3), (2, 100, 167, 128, 128, 3)]
for n, h, w, cin, cout, k in CASES:
    x = torch.randn(n, h, w, cin, device='cuda')
    dy = torch.randn(n, h, w, cout, device='cuda')
    dw = torch.empty(cout, cin, k, k, device='cuda')
    npix = n * h * w
    flops = 2.0 * npix * cin * cout * k * k
    base = t(lambda: ops.conv_wgrad([dy], [x], dw, cin, k, k, 1, k // 2))
    ref = dw.clone()
    line = f'{npix:6d}px {cin:4d}->{cout:4d} k{k}: default {base:6.1f} us ({flops / base / 1e6:5.1f} TF)'
    n2 = k * k * cin
    for hint, bm in ((0, 128), (1, 64)):
        tiles = -(-cout // bm) * -(-n2 // bm)
        for target in (256, 512, 768, 1024, 1536, 2048):
            chunks = max(1, min(target // tiles, -(-npix // 128)))
            chunk = max(32, -(-(-(-npix // chunks)) // 32) * 32)
            try:
                us = t(lambda: wgrad(dy, x, dw, cin, k, chunk, hint))
            except Exception as e:
                us = float('nan')
            err = float((dw - ref).abs().max() / ref.abs().max())
            line += f' | t{bm} b{target}:{us:5.0f}' + ('' if err < 1e-4 else f'(ERR {err:.1e})')
    print(line, flush=True)
