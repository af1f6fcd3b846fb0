import sys, os
sys.path.insert(0, '/root/repo')
import torch
from sacv_amd import ops
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.randn(4, 128, 128, 256, device='cuda'); wt = torch.randn(256, 256, 3, 3, device='cuda') * 0.05
wp = ops.pack_weight(wt, ops.PACK_FWD)
outs = ops.conv_gemm([x], wp, 256, 3, 3, 1, 1, 0, [(128, 128)])
macs = 4 * 128 * 128 * 256 * 256 * 9
for flags, name in ((0, 'normal'), (1024, 'no global loads'), (1024+2048, 'no gl + no ds_write'), (1024+2048+4096, 'no gl, no ds_write, no barrier'), (1024+2048+4096+8192, '+ broadcast LDS reads'), (4096, 'no barrier only'), (0, 'normal again')):
    t = timeit(lambda: ops.conv_gemm([x], wp, 256, 3, 3, 1, 1, 0, [(128, 128)], outs=outs, flags=flags))
    print(name, f'{t*1e3:.1f} us', f'{2*macs/(t*1e-3)/1e12:.1f} TF')
