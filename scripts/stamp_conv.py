"""Diagnostic (SACV_STAMP build only): per-phase cycle shares of the persistent conv kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sacv_amd import ops
from sacv_amd._lib import lib, check, iarr, parr

def run(name, n, h, w, cin, cout, k):
    x = torch.randn(n, h, w, cin, device='cuda'); wt = torch.randn(cout, cin, k, k, device='cuda') * 0.05
    wp = ops.pack_weight(wt, ops.PACK_FWD)
    out = torch.empty(n, h, w, cout, device='cuda')
    stamps = torch.zeros(512 * 64, dtype=torch.int64, device='cuda')
    ws = ops.workspace(lib.sacv_conv2d_gemm_workspace_bytes(), 'cuda', 'conv_gemm')
    for it in range(3):
        stamps.zero_()
        check(lib.sacv_conv2d_gemm(1, parr([x.data_ptr()]), parr([out.data_ptr()]), None, iarr([n]), iarr([h]), iarr([w]),
                                   iarr([h]), iarr([w]), wp.data_ptr(), None, cin, cout, k, k, 1, k // 2, 0, 0, 0, 0, 0, None,
                                   ws.data_ptr(), stamps.data_ptr(), torch.cuda.current_stream().cuda_stream), 'gemm')
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(512, 64)
    pro, loops, epis, totals = [], [], [], []
    for b in range(512):
        u = int((st[b] != 0).sum())
        if u < 4:
            continue
        d = np.diff(st[b, :u].astype(np.float64))
        pro.append(d[0]); loops += list(d[1::2]); epis += list(d[2::2]); totals.append(st[b, u - 1] - st[b, 0])
    KT = k * k * ((cin + 31) // 32)
    loops, epis = np.array(loops), np.array(epis)
    print(f'{name}: blocks {len(totals)}, tiles {len(loops)}; prologue {np.mean(pro):.0f} cyc; K-loop {loops.mean():.0f} cyc/tile '
          f'({loops.mean()/KT:.0f} per K-iteration of {KT}; ideal 8192 at 2 waves/SIMD); epilogue {epis.mean():.0f} cyc/tile; '
          f'shares of block time: loop {loops.sum()/np.sum(totals):.2f} epilogue {epis.sum()/np.sum(totals):.2f} prologue {np.sum(pro)/np.sum(totals):.2f}; '
          f'block total {np.mean(totals):.0f} cyc')


run('balanced 3x3 256', 4, 128, 128, 256, 256, 3)
run('l3 1x1 256->1024', 2, 50, 84, 256, 1024, 1)
run('l1 1x1 64->256', 2, 200, 334, 64, 256, 1)
run('l2 1x1 512->128', 2, 100, 167, 512, 128, 1)
