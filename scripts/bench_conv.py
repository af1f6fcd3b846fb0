"""Micro-benchmark of the implicit-GEMM conv kernels on the hot-path shapes (tuning harness)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import ops

torch.manual_seed(0)
dev = 'cuda'


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# (name, [(N,H,W)], Cin, Cout, k, stride, pad)
LEVELS = [(2, 100, 167), (2, 50, 84), (2, 25, 42), (2, 13, 21), (2, 7, 11)]
CASES = [
    ('head3x3_P3only', [LEVELS[0]], 256, 256, 3, 1, 1),
    ('head3x3_5lvl', LEVELS, 256, 256, 3, 1, 1),
    ('cls_out_5lvl', LEVELS, 256, 720, 3, 1, 1),
    ('l1_3x3_64', [(2, 200, 334)], 64, 64, 3, 1, 1),
    ('l1_1x1_64_256', [(2, 200, 334)], 64, 256, 1, 1, 0),
    ('l1_1x1_256_64', [(2, 200, 334)], 256, 64, 1, 1, 0),
    ('l2_3x3_128', [(2, 100, 167)], 128, 128, 3, 1, 1),
    ('l2_1x1_512_128', [(2, 100, 167)], 512, 128, 1, 1, 0),
    ('l2_1x1_128_512', [(2, 100, 167)], 128, 512, 1, 1, 0),
    ('l3_3x3_256', [(2, 50, 84)], 256, 256, 3, 1, 1),
    ('l3_1x1_1024_256', [(2, 50, 84)], 1024, 256, 1, 1, 0),
    ('l3_1x1_256_1024', [(2, 50, 84)], 256, 1024, 1, 1, 0),
    ('l4_3x3_512', [(2, 25, 42)], 512, 512, 3, 1, 1),
    ('l4_1x1_2048_512', [(2, 25, 42)], 2048, 512, 1, 1, 0),
    ('l4_1x1_512_2048', [(2, 25, 42)], 512, 2048, 1, 1, 0),
    ('fpn_p5_3x3', [(2, 25, 42)], 256, 256, 3, 1, 1),
    ('fpn_p4_3x3', [(2, 50, 84)], 256, 256, 3, 1, 1),
    ('balanced_3x3_256', [(4, 128, 128)], 256, 256, 3, 1, 1),
]
which = sys.argv[1:] or None
for name, lv, cin, cout, k, s, p in CASES:
    if which and name not in which:
        continue
    xs = [torch.randn(n, h, w, cin, device=dev) for n, h, w in lv]
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    hw = [((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1) for n, h, w in lv]
    wp = ops.pack_weight(wt, ops.PACK_FWD)
    wd = ops.pack_weight(wt, ops.PACK_DGRAD)
    outs = ops.conv_gemm(xs, wp, cout, k, k, s, p, 0, hw)
    dys = [torch.randn_like(o) for o in outs]
    dw = torch.empty_like(wt)
    macs = sum(n * h2 * w2 for (n, _, _), (h2, w2) in zip(lv, hw)) * cout * cin * k * k
    t_f = timeit(lambda: ops.conv_gemm(xs, wp, cout, k, k, s, p, 0, hw, outs=outs))
    dxs = [torch.empty_like(x) for x in xs]
    t_d = timeit(lambda: ops.conv_gemm(dys, wd, cin, k, k, s, p, 1, [(x.shape[1], x.shape[2]) for x in xs], outs=dxs))
    t_w = timeit(lambda: ops.conv_wgrad(dys, xs, dw, cin, k, k, s, p))
    tf = lambda t: 2 * macs / (t * 1e-3) / 1e12
    print(f'{name:18s} GMAC {macs/1e9:7.2f} | fwd {t_f*1e3:8.1f} us {tf(t_f):6.1f} TF | dgrad {t_d*1e3:8.1f} us {tf(t_d):6.1f} TF '
          f'| wgrad {t_w*1e3:8.1f} us {tf(t_w):6.1f} TF', flush=True)
