"""sacv_conv2d_gemm with fp32 operands on the fp32 matrix pipe against the split-operand form (three bf16 terms at the
LDS store, six products on the bf16 pipe): us per call on the trunk shapes of the headline workload, forward and data
gradient, and the error of both against fp64 on a sample of outputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sacv_amd  # noqa
from sacv_amd import ops


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator(device='cuda').manual_seed(3)
SHAPES = [  # (h, w, cin, cout, k)
    (200, 334, 64, 256, 1), (200, 334, 256, 64, 1), (200, 334, 64, 64, 3),
    (100, 167, 128, 512, 1), (100, 167, 512, 128, 1), (100, 167, 128, 128, 3),
    (50, 84, 256, 1024, 1), (50, 84, 1024, 256, 1), (50, 84, 256, 256, 3),
    (25, 42, 512, 2048, 1), (25, 42, 2048, 512, 1), (25, 42, 512, 512, 3),
]
nb = 2
for h, w, cin, cout, k in SHAPES:
    x = torch.randn((nb, h, w, cin), device='cuda', generator=g)
    wt = torch.randn((cout, k, k, cin), device='cuda', generator=g) * (2.0 / (k * k * cin)) ** 0.5
    line = f'{h}x{w} {cin}->{cout} k{k}:'
    outs = {}
    for sp in (0, 1):
        prev = ops.launch_share(nt_split=sp)
        f = lambda: ops.conv_gemm([x], wt, cout, k, k, 1, k // 2, 0, [(h, w)], flags=ops.CF_RELU)
        t = timeit(f)
        outs[sp] = f()[0]
        ops.launch_share(*prev)
        line += f'  {"split" if sp else "fp32"} {t:7.1f} us'
    # error vs fp64 on a crop
    xs = x[:, :16, :16].permute(0, 3, 1, 2).double().cpu()
    ref = torch.nn.functional.conv2d(xs, wt.permute(0, 3, 1, 2).double().cpu(), padding=k // 2).relu()
    c = slice(2, 14)
    for sp in (0, 1):
        got = outs[sp][:, :16, :16].permute(0, 3, 1, 2).double().cpu()
        e = float((got[:, :, c, c] - ref[:, :, c, c]).norm() / ref[:, :, c, c].norm())
        line += f'  err {e:.2e}'
    print(line, flush=True)
    # data gradient (mode 1) of the same layer: dy [nb, h, w, cout] -> dx, small-magnitude dy as in a real backward pass
    dy = torch.randn((nb, h, w, cout), device='cuda', generator=g) * 1e-4
    wd = ops.pack_weight(wt.permute(0, 3, 1, 2).contiguous(), ops.PACK_DGRAD)
    line = f'    data gradient:'
    outs = {}
    for sp in (0, 1):
        prev = ops.launch_share(nt_split=sp)
        f = lambda: ops.conv_gemm([dy], wd, cin, k, k, 1, k // 2, 1, [(h, w)])
        t = timeit(f)
        outs[sp] = f()[0]
        ops.launch_share(*prev)
        line += f'  {"split" if sp else "fp32"} {t:7.1f} us'
    dys = dy[:, :16, :16].permute(0, 3, 1, 2).double().cpu()
    ref = torch.nn.functional.conv_transpose2d(dys, wt.permute(0, 3, 1, 2).double().cpu(), padding=k // 2)
    for sp in (0, 1):
        got = outs[sp][:, :16, :16].permute(0, 3, 1, 2).double().cpu()
        e = float((got[:, :, c, c] - ref[:, :, c, c]).norm() / ref[:, :, c, c].norm())
        line += f'  err {e:.2e}'
    line += f'  split-vs-fp32 {float((outs[1] - outs[0]).norm() / outs[0].norm()):.2e}'
    print(line, flush=True)
    # weight gradient
    line = f'    weight gradient:'
    outs = {}
    for sp in (0, 1):
        prev = ops.launch_share(nt_split=sp)
        dw = torch.empty((cout, k, k, cin), device='cuda').permute(0, 3, 1, 2)
        f = lambda: ops.conv_wgrad([dy], [x], dw, cin, k, k, 1, k // 2)
        t = timeit(f)
        f()
        outs[sp] = dw.clone()
        ops.launch_share(*prev)
        line += f'  {"split" if sp else "fp32"} {t:7.1f} us'
    line += f'  split-vs-fp32 {float((outs[1] - outs[0]).norm() / outs[0].norm()):.2e}'
    print(line, flush=True)
