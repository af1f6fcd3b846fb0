"""Winograd F(2x2,3x3) vs the direct implicit-GEMM kernel on the head-tower shapes of the headline workload
(5 levels of 2 x 800x1333).  Usage: python scripts/bench_winograd.py [H W]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from sacv_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    quick = 'quick' in sys.argv
    argv = [a for a in sys.argv if a != 'quick']
    H, W = (int(argv[1]), int(argv[2])) if len(argv) > 2 else (800, 1333)
    nb = 2
    hw = []
    h, w = -(-H // 8), -(-W // 8)
    for _ in range(5):
        hw.append((h, w))
        h, w = -(-h // 2), -(-w // 2)
    pix = sum(nb * h * w for h, w in hw)
    g = torch.Generator(device='cuda').manual_seed(1)
    for cin, cout in [(256, 256)] if quick else [(256, 256), (256, 720), (720, 256)]:
        srcs = [torch.randn((nb, h, w, cin), device='cuda', generator=g) for h, w in hw]
        wt = torch.randn((cout, 3, 3, cin), device='cuda', generator=g) * 0.02
        b = torch.randn(cout, device='cuda', generator=g)
        u = ops.wino_weights(wt, cout, cin)
        fl = 2.0 * pix * cout * cin * 9
        t_d = timeit(lambda: ops.conv_gemm(srcs, wt, cout, 3, 3, 1, 1, 0, hw, bias=b, flags=ops.CF_RELU))
        t_in = timeit(lambda: ops.wino_input(srcs))
        v = ops.wino_input(srcs, tag='keep')
        t_g = timeit(lambda: ops.conv3x3_wino([srcs], [u], [cout], biases=[b], flags=[ops.CF_RELU], vs=[v]))
        t_w = timeit(lambda: ops.conv3x3_wino([srcs], [u], [cout], biases=[b], flags=[ops.CF_RELU]))
        t_u = timeit(lambda: ops.wino_weights(wt, cout, cin, out=u))
        torch.cuda.synchronize()
        print(f'{cin}->{cout} pix {pix}: direct {t_d*1e3:.0f} us ({fl/t_d/1e9:.1f} TF)  wino {t_w*1e3:.0f} us '
              f'({fl/t_w/1e9:.1f} TF-eq) = input {t_in*1e3:.0f} + gemm {t_g*1e3:.0f}; weights {t_u*1e3:.0f} us',
              flush=True)
        if cin == 256:
            gys = [torch.randn((nb, h, w, cout), device='cuda', generator=g) for h, w in hw]
            dw = torch.empty((cout, 3, 3, cin), device='cuda').permute(0, 3, 1, 2)
            t_wd = timeit(lambda: ops.conv_wgrad(gys, srcs, dw, cin, 3, 3, 1, 1))
            t_ww = timeit(lambda: ops.wino_wgrad(gys, v, dw, cin))
            print(f'   weight gradient: direct {t_wd*1e3:.0f} us ({fl/t_wd/1e9:.1f} TF)  wino {t_ww*1e3:.0f} us '
                  f'({fl/t_ww/1e9:.1f} TF-eq)', flush=True)
        if cin == cout:
            srcs2 = [torch.randn((nb, h, w, cin), device='cuda', generator=g) for h, w in hw]
            t_2 = timeit(lambda: ops.conv3x3_wino([srcs, srcs2], [u, u], [cout, cout], biases=[b, b],
                                                  flags=[ops.CF_RELU, ops.CF_RELU]))
            print(f'   two towers in one launch: {t_2*1e3:.0f} us = {t_2*5e2:.0f} us per layer '
                  f'({2*fl/t_2/1e9:.1f} TF-eq)', flush=True)


if __name__ == '__main__':
    main()
