"""Stem convolution at 2 x 800 x 1333: im2col + GEMM / conv_tn against the image-reading forward and weight gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import ops, synthetic as synth
x = torch.tensor(synth.images(0, 2, 800, 1333)).cuda()
wt = (torch.randn(64, 3, 7, 7) * 0.05).cuda()
wp = ops.pack_weight(wt, ops.PACK_STEM)
ho, wo = 400, 667
dy = torch.randn(2, ho, wo, 64, device='cuda')
dw = torch.empty_like(wt)
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
col = ops.im2col_stem(x, ho, wo)
print('im2col            %7.1f us' % t(lambda: ops.im2col_stem(x, ho, wo)))
print('gemm on col       %7.1f us' % t(lambda: ops.conv_gemm([col], wp, 64, 1, 1, 1, 0, 0, [(ho, wo)], stats=True)))
print('stem_conv fwd     %7.1f us' % t(lambda: ops.stem_conv(x, wp, ho, wo, stats=True)))
print('conv_tn on col    %7.1f us' % t(lambda: ops.conv_wgrad([dy], [col], dw, 3, 1, 1, 1, 0, cin_pad=ops.STEM_KPAD, dw_taps=49)))
print('stem_conv wgrad   %7.1f us' % t(lambda: ops.stem_conv_wgrad(x, dy, dw)))
