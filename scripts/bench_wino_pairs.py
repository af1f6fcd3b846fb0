import sys, torch
sys.path.insert(0, '.')
from sacv_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
nb=2; hw=[(100,167),(50,84),(25,42),(13,21),(7,11)]
g=torch.Generator(device='cuda').manual_seed(1)
x1=[torch.randn((nb,h,w,256),device='cuda',generator=g) for h,w in hw]
x2=[torch.randn((nb,h,w,256),device='cuda',generator=g) for h,w in hw]
w720=torch.randn((720,3,3,256),device='cuda',generator=g)*0.02
w36=torch.randn((36,3,3,256),device='cuda',generator=g)*0.02
u720=ops.wino_weights(w720,720,256); u36=ops.wino_weights(w36,36,256)
b720=torch.zeros(720,device='cuda'); b36=torch.zeros(36,device='cuda')
print('fwd pair wino', timeit(lambda: ops.conv3x3_wino([x1,x2],[u720,u36],[720,36],biases=[b720,b36],flags=[ops.CF_SIGMOID,0])))
print('fwd cls wino alone', timeit(lambda: ops.conv3x3_wino([x1],[u720],[720],biases=[b720],flags=[ops.CF_SIGMOID])))
print('fwd reg direct', timeit(lambda: ops.conv_gemm(x2, w36, 36, 3,3,1,1,0,hw,bias=b36)))
print('fwd reg wino alone', timeit(lambda: ops.conv3x3_wino([x2],[u36],[36],biases=[b36])))
# dgrad
d720=[torch.randn((nb,h,w,720),device='cuda',generator=g) for h,w in hw]
d36=[torch.randn((nb,h,w,36),device='cuda',generator=g) for h,w in hw]
pd720=w720.permute(3,1,2,0).contiguous()  # [ci][9][co]
pd36=w36.permute(3,1,2,0).contiguous()
ud720=ops.wino_weights(pd720,256,720,flip=True); ud36=ops.wino_weights(pd36,256,36,flip=True)
print('dgrad pair wino', timeit(lambda: ops.conv3x3_wino([d720,d36],[ud720,ud36],[256,256],mode=1)))
print('dgrad cls wino alone', timeit(lambda: ops.conv3x3_wino([d720],[ud720],[256],mode=1)))
print('dgrad reg wino alone', timeit(lambda: ops.conv3x3_wino([d36],[ud36],[256],mode=1)))
print('dgrad reg direct', timeit(lambda: ops.conv_gemm(d36, pd36.view(256,-1), 256, 3,3,1,1,1,hw)))
