"""Classifier at 4 x 64 x 64: gradients and the loss after one small-lr SGD step, image-reading stem vs im2col stem,
against the CPU oracle (diagnostic for tests/test_gpu_cls.py::test_classification_trainer_small_lr_vs_oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model as omodel
from tests.test_cls_cpu import oracle_cls_run
from tests.test_gpu_cls import build
from sacv_amd.classification import losses
from sacv_amd.train import ClassificationTrainer
lr, B, R, nc = 1e-4, 4, 64, 1000
_, loss_r, params, state, x, lab = oracle_cls_run('cls_64', dict(B=B, R=R, num_classes=nc))
G = {k: p.grad.clone() for k, p in params.items()}
P = {k: p.detach().clone() for k, p in params.items()}
omodel.sgd_momentum_step(P, G, {}, lr, 0.9, 1e-4, first=True)
state.update(P)
with torch.no_grad():
    l1_r = float(omodel.ce_loss(omodel.resnet50_cls_forward(state, x, training=True), lab))
for fused in (False, True, 'eval'):
    net = build(nc)
    net._engine.fused_stem = fused
    tr = ClassificationTrainer(net, losses.__dict__['CELoss'](), lr=lr, momentum=0.9, weight_decay=1e-4)
    l0 = float(tr.step(x.cuda(), lab.cuda())['loss'])
    errs = {}
    for k, p in net.named_parameters():
        g, gg = G[k].double(), net.arena.grad_views[k].cpu().double()
        errs[k] = float((gg - g).norm() / (g.norm() + 1e-30))
    l1 = float(tr.step(x.cuda(), lab.cuda())['loss'])
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f'fused_stem={fused}: loss {l0:.5f} -> {l1:.5f} (oracle {float(loss_r):.5f} -> {l1_r:.5f}); stem dW rel-L2 '
          f'{errs["conv1.layer.0.weight"]:.3e}; median {sorted(errs.values())[len(errs)//2]:.3e}; worst {worst}')
