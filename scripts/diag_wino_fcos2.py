"""FCOS, Winograd vs direct engine, image-reading vs im2col stem, against the fp64 oracle for several input seeds:
is the Winograd + image-reading-stem combination CONSISTENTLY farther from the exact gradient (a bug) or not (noise)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth, model as omodel, losses as olosses
from sacv_amd.detection import models, losses

def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
H, W = 544, 736
torch.manual_seed(0)
state0 = {k: v.clone() for k, v in models.__dict__['resnet50_fcos'](num_classes=80).state_dict().items()}
keys = ['fpn.P3_1.weight', 'fpn.P3_2.weight', 'fpn.P4_2.weight', 'fpn.P5_2.weight', 'clsregcnt_head.cls_head.0.weight']
for seed in (7, 21, 33):
    images = torch.tensor(synth.images(seed, 2, H, W)).cuda()
    annots = torch.tensor(synth.annotations(seed + 1, 2, H, W, boxes_per_image=6)).cuda()
    def run(wino, **settings):
        net = models.__dict__['resnet50_fcos'](num_classes=80)
        net.load_state_dict(state0)
        net = net.cuda().train()
        net.engine.winograd = wino
        net.engine.WINO_MIN_TILES = 1000
        for k, v in settings.items():
            setattr(net.engine, k, v)
        ld = losses.__dict__['FCOSLoss']()(net(images), annots)
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        return {k: net.arena.grad_views[k].clone() for k in keys}
    sd64 = {k: (v.double().cpu() if v.dtype.is_floating_point else v.clone().cpu()) for k, v in state0.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.dtype.is_floating_point and 'running_' not in k}
    st = dict(sd64); st.update(params)
    ld = olosses.FCOSLossOracle()(omodel.fcos_forward(st, images.cpu().double(), training=True), annots.cpu())
    sum(ld.values()).backward()
    for label, wino, stg in (('wino  fused ', True, {}), ('direct fused ', False, {}), ('wino  im2col', True, dict(fused_stem='eval')),
                             ('direct im2col', False, dict(fused_stem='eval'))):
        g = run(wino, **stg)
        print(seed, label, ' '.join(f'{k.split(".")[-2]}:{rel(g[k].cpu(), params[k].grad):.2e}' for k in keys), flush=True)
