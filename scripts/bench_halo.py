"""Microbenchmark of the layer1 3x3 convolution of the bf16 classifier (64 -> 64 at 56 x 56, batch 256): forward and
data gradient through sacv_conv2d_gemm_bf16 (the halo-resident kernel when it applies), HIP-event time over 20 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import ops
n, h, w, c = 256, 56, 56, 64
x = torch.randn(n, h, w, c, device='cuda').to(torch.bfloat16)
wp = (torch.randn(c, 9 * c, device='cuda') * 0.05).to(torch.bfloat16)
for mode in (0, 1):
    for stats in ((True, False) if mode == 0 else (False,)):
        f = lambda: ops.conv_gemm([x], wp, c, 3, 3, 1, 1, mode, [(h, w)], stats=stats)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * n * h * w * c * 9 * c
        by = 2.0 * n * h * w * c * 2
        print(f'mode {mode} stats {stats}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s  {by / us / 1e3:.0f} GB/s = {by / us / 1e3 / 8000:.3f} of 8 TB/s')
