"""Bit patterns of the convolution kernels' outputs on the headline step's shapes: run once per library build
(SACV_HIP_LIB=...) and diff the two listings.  Round 4: the in-kernel split-tile finish against the fix-up launch of
round 3 (gpurun_variants/libsacv_r3.so) -- outputs AND CF_STATS column sums must be the same bits.
    python scripts/conv_bits.py > a.txt; SACV_HIP_LIB=$PWD/gpurun_variants/libsacv_r3.so python scripts/conv_bits.py > b.txt"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import ops

def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]

def rnd(seed, *shape, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).cuda()

# (N, H, W, Cin, Cout, k, stride, pad): the trunk / FPN shapes at 2 x 800 x 1333 and a few ragged ones
SHAPES = [(2, 200, 334, 64, 64, 1, 1, 0), (2, 200, 334, 64, 64, 3, 1, 1), (2, 200, 334, 64, 256, 1, 1, 0),
          (2, 200, 334, 256, 64, 1, 1, 0), (2, 200, 334, 256, 128, 1, 1, 0), (2, 200, 334, 128, 128, 3, 2, 1),
          (2, 100, 167, 128, 512, 1, 1, 0), (2, 100, 167, 512, 128, 1, 1, 0), (2, 100, 167, 128, 128, 3, 1, 1),
          (2, 100, 167, 512, 256, 1, 1, 0), (2, 100, 167, 256, 256, 3, 2, 1), (2, 50, 84, 256, 1024, 1, 1, 0),
          (2, 50, 84, 1024, 256, 1, 1, 0), (2, 50, 84, 256, 256, 3, 1, 1), (2, 50, 84, 1024, 512, 1, 1, 0),
          (2, 50, 84, 512, 512, 3, 2, 1), (2, 25, 42, 512, 2048, 1, 1, 0), (2, 25, 42, 2048, 512, 1, 1, 0),
          (2, 25, 42, 512, 512, 3, 1, 1), (2, 25, 42, 2048, 256, 1, 1, 0), (2, 25, 42, 2048, 256, 3, 2, 1),
          (2, 37, 41, 96, 160, 3, 1, 1), (3, 19, 23, 64, 72, 1, 1, 0), (1, 13, 21, 256, 720, 3, 1, 1)]
for i, (n, h, w, ci, co, k, s, p) in enumerate(SHAPES):
    x = rnd(100 + i, n, h, w, ci)
    wt = rnd(200 + i, co, k, k, ci, std=(1.0 / (ci * k * k)) ** 0.5)  # forward panel [Cout][taps*Cin]
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    (y,), st = ops.conv_gemm([x], wt.view(co, -1), co, k, k, s, p, 0, [(ho, wo)], stats=True)
    b = rnd(300 + i, co)
    y2 = ops.conv_gemm([x], wt.view(co, -1), co, k, k, s, p, 0, [(ho, wo)], bias=b, flags=ops.CF_RELU)[0]
    line = f'fwd {n}x{h}x{w} {ci}->{co} k{k}s{s}: y {sha(y)} stats {sha(st)} relu {sha(y2)}'
    if s == 1:
        dy = rnd(400 + i, n, ho, wo, co)
        wd = rnd(500 + i, ci, k * k * co, std=(1.0 / (co * k * k)) ** 0.5)
        dx = ops.conv_gemm([dy], wd, ci, k, k, 1, p, 1, [(h, w)])[0]
        acc = rnd(600 + i, n, h, w, ci)
        msk = rnd(700 + i, n, h, w, ci)
        dx2, st2 = ops.conv_gemm([dy], wd, ci, k, k, 1, p, 1, [(h, w)], flags=ops.CF_ACCUM | ops.CF_MASK, masks=[msk],
                                 outs=[acc], stats=True)
        line += f' dgrad {sha(dx)} accum+mask {sha(dx2[0])} stats {sha(st2)}'
    print(line, flush=True)
# grouped launch: five pyramid levels sharing a weight (direct 3x3 head conv)
lv = [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]
xs = [rnd(800 + j, 2, h, w, 256) for j, (h, w) in enumerate(lv)]
wt = rnd(900, 256, 9 * 256, std=(1.0 / 2304) ** 0.5)
ys = ops.conv_gemm(xs, wt, 256, 3, 3, 1, 1, 0, lv, bias=rnd(901, 256), flags=ops.CF_RELU)
print('grouped 5 levels 256->256 k3:', ' '.join(sha(y) for y in ys))
ys, st = ops.conv_gemm(xs, wt, 256, 3, 3, 1, 1, 1, lv, flags=ops.CF_MASK, masks=xs, stats=True)
print('grouped dgrad mask+stats:', ' '.join(sha(y) for y in ys), sha(st))
torch.cuda.synchronize()
