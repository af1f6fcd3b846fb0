import sys; sys.path.insert(0,'/root/repo')
import torch
from sacv_amd import ops
n=256*1024*1024
a=torch.empty(n,device='cuda'); b=torch.empty(n,device='cuda')
def t(fn,reps=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
ms=t(lambda: ops.fill(a,1.0)); print('fill (write only) %.2f TB/s'%(n*4/ms/1e9))
ms=t(lambda: a.zero_()); print('torch zero_ %.2f TB/s'%(n*4/ms/1e9))
ms=t(lambda: b.copy_(a)); print('copy (r+w) %.2f TB/s total'%(2*n*4/ms/1e9))
ms=t(lambda: a.sum()); print('sum (read only) %.2f TB/s'%(n*4/ms/1e9))
ms=t(lambda: ops.eltwise(ops.EW_RELU,a,out=b)); print('eltwise relu (r+w) %.2f TB/s total'%(2*n*4/ms/1e9))
