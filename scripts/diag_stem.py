"""Fused stem vs im2col + GEMM: gradient arenas of one classifier / detector step (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import synthetic as synth
from sacv_amd.classification import backbones, losses as closses

def cls_run(fused):
    torch.manual_seed(0)
    net = backbones.__dict__['resnet50'](num_classes=1000).cuda().train()
    net._engine.fused_stem = fused
    x = torch.tensor(synth.images(1, 4, 64, 64)).cuda()
    lab = torch.tensor([1, 5, 7, 900]).cuda()
    loss = closses.__dict__['CELoss']()(net(x), lab)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: net.arena.grad_views[k].clone() for k, _ in net.named_parameters()}

for name, run in (('cls', cls_run),):
    l0, g0 = run(False)
    l1, g1 = run(True)
    print(name, 'loss', l0, l1)
    worst = sorted(((float((g0[k] - g1[k]).abs().max()) / (float(g0[k].abs().max()) + 1e-30), k) for k in g0), reverse=True)[:6]
    for w, k in worst:
        print(f'  {k}: max rel diff {w:.3e}')
