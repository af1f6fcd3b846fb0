"""bf16 NT GEMM shapes of the MAE step and of the classifier's 1x1 convolutions through sacv_conv2d_gemm_bf16
(A/B two library builds with SACV_HIP_LIB)."""
import sys

import torch

sys.path.insert(0, '.')
from sacv_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(6400, 768, 2304), (6400, 768, 3072), (6400, 3072, 768), (6400, 768, 768), (25216, 512, 2048),
          (25216, 2048, 512), (25216, 512, 1536), (25216, 512, 512), (200704, 256, 64), (200704, 64, 256),
          (200704, 128, 512), (50176, 1024, 256), (50176, 256, 1024), (12544, 2048, 512)]
g = torch.Generator(device='cuda').manual_seed(1)
for m, k, n in shapes:
    x = torch.randn((1, 1, m, k), device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn((n, k), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
    t = timeit(lambda: ops.conv_gemm([x], w, n, 1, 1, 1, 0, 0, [(1, m)]))
    print(f'M {m:7d} K {k:5d} N {n:5d}: {t*1e3:7.1f} us  {2.0*m*k*n/t/1e9:7.1f} TF', flush=True)
