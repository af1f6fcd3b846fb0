"""Winograd GEMM on the headline head shapes: fp32 operands (fp32 matrix pipe) against the three-bf16-plane form
(bf16 pipe, six products).  us per call, transforms excluded and included; error of both against fp64 on one level."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sacv_amd  # noqa
from sacv_amd import ops


def timeit(fn, n=20, warm=5):
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


nb, c = 2, 256
hw = [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]
if '--small' in sys.argv:  # V of a tower fits the Infinity Cache
    hw = hw[1:]
FORMS = (True,) if '--split-only' in sys.argv else (False, True)
g = torch.Generator(device='cuda').manual_seed(5)
srcs = [torch.randn((nb, h, w, c), device='cuda', generator=g) for h, w in hw]
srcs2 = [torch.randn((nb, h, w, c), device='cuda', generator=g) for h, w in hw]
w = torch.randn((c, 3, 3, c), device='cuda', generator=g) * (2.0 / (9 * c)) ** 0.5
w720 = torch.randn((720, 3, 3, c), device='cuda', generator=g) * (2.0 / (9 * c)) ** 0.5
b = torch.randn(c, device='cuda', generator=g) * 0.1
masks = [torch.randn((nb, h, ww, c), device='cuda', generator=g) for h, ww in hw]
for split in FORMS:
    u = ops.wino_weights(w, c, c, split=split)
    u720 = ops.wino_weights(w720, 720, c, split=split)
    v = ops.wino_input(srcs, tag='keep', split=split)
    v2 = ops.wino_input(srcs2, tag='keep2', split=split)
    t_in = timeit(lambda: ops.wino_input(srcs, tag='scr', split=split))
    t_1 = timeit(lambda: ops.conv3x3_wino([srcs], [u], [c], biases=[b], flags=[ops.CF_RELU], vs=[v]))
    t_2 = timeit(lambda: ops.conv3x3_wino([srcs, srcs2], [u, u], [c, c], biases=[b, b], flags=[ops.CF_RELU] * 2, vs=[v, v2]))
    t_2m = timeit(lambda: ops.conv3x3_wino([srcs, srcs2], [u, u], [c, c], flags=[ops.CF_MASK] * 2, vs=[v, v2],
                                           masks_list=[masks, masks], mode=1))
    t_720 = timeit(lambda: ops.conv3x3_wino([srcs], [u720], [720], vs=[v]))
    dw = torch.empty((c, 3, 3, c), device='cuda').permute(0, 3, 1, 2)
    t_wg = timeit(lambda: ops.wino_wgrad(srcs2, v, dw, c))
    print(f'split={split}: input transform {t_in:.1f} us | GEMM one tower {t_1:.1f} | two towers {t_2:.1f} | '
          f'two towers mask epilogue {t_2m:.1f} | 720-channel output conv {t_720:.1f} | weight gradient {t_wg:.1f}',
          flush=True)
    o = ops.conv3x3_wino([srcs], [u], [c], biases=[b], flags=[0], vs=[v])[0]
    x0 = srcs[1].permute(0, 3, 1, 2).double().cpu()
    ref = torch.nn.functional.conv2d(x0, w.permute(0, 3, 1, 2).double().cpu(), b.double().cpu(), padding=1)
    got = o[1].permute(0, 3, 1, 2).double().cpu()
    print(f'   level-1 error vs fp64: rel L2 {float((got - ref).norm() / ref.norm()):.3e}  max abs {float((got - ref).abs().max()):.3e}')
