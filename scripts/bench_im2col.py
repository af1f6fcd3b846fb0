import sys; sys.path.insert(0,'/root/repo')
import torch
from sacv_amd import ops
for shape in [(2,3,800,1333),(256,3,224,224)]:
    x=torch.randn(*shape,device='cuda')
    n,_,h,w=shape
    ho,wo=(h+6-7)//2+1,(w+6-7)//2+1
    for xx,name in ((x,'nchw'),(x.contiguous(memory_format=torch.channels_last),'cl')):
        c=ops.im2col_stem(xx,ho,wo); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): c=ops.im2col_stem(xx,ho,wo)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print(shape,name,f'{ms*1e3:.1f} us', f'{c.numel()*4/ms/1e9:.2f} TB/s write')
