"""In-kernel phase timing of wino_gemm_kernel (diagnostic build with -DWINO_STAMP; never the shipped library): per
wave, s_memtime ticks spent in (0) issuing the LDS-DMA of tile q+2 (+ issuing the fragment reads in the split form),
(5) waiting for the fragment reads (split form), (1) the MFMA block, (2) the counted vmcnt wait, (3) the barrier,
(4) the per-g fold.  SACV_HIP_LIB must point at the stamp build.  Usage: stamp_winograd.py [fp32|split_small|split_big]"""
import sys

import torch

sys.path.insert(0, '.')
from sacv_amd import ops
from sacv_amd._lib import lib, iarr, parr, check

form = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
split = form != 'fp32'
ops.launch_share(wino_big=2 if form == 'split_big' else 0)
nb, c = 2, 256
hw = [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]
g = torch.Generator(device='cuda').manual_seed(1)
srcs = [torch.randn((nb, h, w, c), device='cuda', generator=g) for h, w in hw]
wt = torch.randn((c, 3, 3, c), device='cuda', generator=g) * 0.02
u = ops.wino_weights(wt, c, c, split=split)
t, tpad = ops.wino_tiles(nb, hw)
for npb in (1, 2):
    v = [ops.wino_input(srcs, tag=f'sv{p}', split=split) for p in range(npb)]
    outs_l = [[torch.empty((nb, h, w, c), device='cuda') for h, w in hw] for _ in range(npb)]
    big = [torch.zeros(768 * 8 * 8 * 2 + (1 << 20), device='cuda', dtype=torch.float32) for _ in range(npb)]
    dst = sum([[x.data_ptr() for x in outs_l[p]] for p in range(npb)], [])
    for _ in range(3):
        check(lib.sacv_wino_gemm(npb, 5, nb, iarr([h for h, _ in hw]), iarr([w for _, w in hw]),
                                 parr([x.data_ptr() for x in v]), parr([u.data_ptr()] * npb), iarr([tpad] * npb),
                                 iarr([c] * npb), iarr([c] * npb), None, parr(dst), None,
                                 parr([b.data_ptr() for b in big]), iarr([0] * npb), int(split),
                                 torch.cuda.current_stream().cuda_stream), 'wino_gemm')
    torch.cuda.synchronize()
    names = ['issue(+reads)', 'mfma block', 'vmcnt wait', 'barrier', 'fold', 'read wait']
    print(f'{form}, {npb} problem(s):')
    for p in range(npb):
        raw = big[p].view(torch.int64)[:768 * 8 * 8].view(768 * 8, 8).cpu()
        used = raw[:, 1] > 0
        tot = raw[:, :6].sum(-1).float()
        print(f'  problem {p}: waves stamped {int(used.sum())}, ticks per wave {float(tot[used].mean()):.0f}')
        for k, n in enumerate(names):
            vals = raw[:, k][used].float()
            print(f'    {n:14s} mean {vals.mean():10.0f} min {vals.min():10.0f} max {vals.max():10.0f} '
                  f'({100 * vals.sum() / tot[used].sum():5.1f} %)')
