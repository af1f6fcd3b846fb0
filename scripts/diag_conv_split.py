"""Where do the two operand forms of sacv_conv2d_gemm differ?  Small shapes (few tiles: every tile K-split over many
workgroups + fix-up), strides, grouped launches, epilogue options; error of each form against fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import sacv_amd  # noqa
from sacv_amd import ops

g = torch.Generator().manual_seed(0)
CASES = [(2, 64, 25, 42, 64, 3, 1, 1), (2, 64, 19, 23, 256, 1, 1, 0), (2, 256, 13, 21, 64, 1, 1, 0),
         (2, 128, 26, 31, 128, 3, 2, 1), (2, 256, 25, 42, 512, 1, 2, 0), (2, 256, 13, 21, 256, 3, 1, 1),
         (1, 256, 7, 11, 720, 3, 1, 1), (2, 256, 7, 11, 36, 3, 1, 1), (2, 512, 9, 10, 256, 3, 2, 1),
         (2, 32, 8, 9, 96, 3, 1, 1), (2, 64, 13, 21, 64, 1, 1, 0), (2, 16, 13, 21, 64, 1, 1, 0),
         (2, 2048, 7, 11, 512, 1, 1, 0), (2, 1024, 13, 21, 256, 1, 1, 0)]
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
nchw = lambda t: t.permute(0, 3, 1, 2).double().cpu()
rel = lambda a, b: float((a - b).norm() / b.norm())
for n, cin, h, w, cout, k, s, p in CASES:
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (1.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g)
    yr = F.conv2d(x.double(), wt.double(), b.double(), s, p)
    dy = torch.randn(yr.shape, generator=g)
    dxr = F.conv_transpose2d(dy.double(), wt.double(), None, s, p, output_padding=(h + 2 * p - k) % s if s > 1 else 0)
    if dxr.shape[2:] != (h, w):
        dxr = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), s, p)
    ho, wo = yr.shape[2:]
    res = []
    for sp in (0, 1):
        prev = ops.launch_share(nt_split=sp)
        y = ops.conv_gemm([nhwc(x)], ops.pack_weight(wt.cuda(), ops.PACK_FWD), cout, k, k, s, p, 0, [(ho, wo)], bias=b.cuda())[0]
        if not (k == 1 and s > 1):
            dx = ops.conv_gemm([nhwc(dy)], ops.pack_weight(wt.cuda(), ops.PACK_DGRAD), cin, k, k, s, p, 1, [(h, w)])[0]
        else:
            dx = torch.zeros((n, h, w, cin), device='cuda')
            ops.conv_gemm([nhwc(dy)], ops.pack_weight(wt.cuda(), ops.PACK_DGRAD), cin, 1, 1, 1, 0, 0, [(ho, wo)],
                          flags=ops.CF_ACCUM, outs=[dx], remap=(s, h, w))
        ops.launch_share(*prev)
        res.append((rel(nchw(y), yr), rel(nchw(dx), dxr)))
    print(f'{(n, cin, h, w, cout, k, s, p)}: fwd err fp32 {res[0][0]:.2e} split {res[1][0]:.2e} | dgrad fp32 {res[0][1]:.2e} '
          f'split {res[1][1]:.2e}', flush=True)
