"""Winograd-path engine vs direct-kernel engine (tests/test_gpu_winograd.py::test_detector_step_winograd_matches_direct)
under several executor settings: which round-4 change moved fpn.P5_2.weight apart?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model as omodel, synth
from sacv_amd.detection import models, losses

def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

H, W = 544, 736
images = torch.tensor(synth.images(7, 2, H, W)).cuda()
annots = torch.tensor(synth.annotations(8, 2, H, W, boxes_per_image=6)).cuda()
state0 = omodel.init_retinanet_state(0)

def run(wino, **settings):
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    net.load_state_dict(state0)
    net = net.cuda().train()
    net.engine.winograd = wino
    net.engine.WINO_MIN_TILES = 1000
    for k, v in settings.items():
        setattr(net.engine, k, v)
    ld = losses.__dict__['RetinaLoss']()(net(images), annots)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    return {k: g.clone() for k, g in net.arena.grad_views.items()}

for label, st in (('default', {}), ('side_stream=False', dict(side_stream=False)), ('fuse_bn_bwd=False', dict(fuse_bn_bwd=False)),
                  ('fused_stem=eval', dict(fused_stem='eval'))):
    gw, gd = run(True, **st), run(False, **st)
    worst = sorted(((rel(gw[k], gd[k]), k) for k in gd if float(gd[k].double().norm()) > 0), reverse=True)[:4]
    print(label, [(f'{e:.2e}', k) for e, k in worst], flush=True)
# the direct engine against itself, run to run and side stream on / off
a, b, c = run(False), run(False), run(False, side_stream=False)
print('direct vs direct (rerun):', max(rel(a[k], b[k]) for k in a), ' side stream on vs off:', max(rel(a[k], c[k]) for k in a))
a, b, c = run(True), run(True), run(True, side_stream=False)
print('wino vs wino (rerun):', max(rel(a[k], b[k]) for k in a), ' side stream on vs off:', max(rel(a[k], c[k]) for k in a))

# which engine moved?  both against the CPU oracle (fp64) for the tensors that differ most
from oracle import losses as olosses
sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in state0.items()}
params = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.dtype.is_floating_point and 'running_' not in k}
st = dict(sd64); st.update(params)
outs = omodel.retinanet_forward(st, images.cpu().double(), training=True)
ld = olosses.RetinaLossOracle()(outs, annots.cpu())
sum(ld.values()).backward()
keys = ['fpn.P5_2.weight', 'fpn.P5_2.bias', 'backbone.layer2.0.conv1.layer.1.weight', 'backbone.layer1.0.conv2.layer.1.bias',
        'backbone.conv1.layer.0.weight', 'cls_head.cls_head.0.weight', 'fpn.P3_2.weight']
for label, wino, stg in (('wino  fused', True, {}), ('direct fused', False, {}), ('wino  im2col', True, dict(fused_stem='eval')),
                         ('direct im2col', False, dict(fused_stem='eval'))):
    g = run(wino, **stg)
    print(label, ' '.join(f'{k.split(".")[-3] if k.count(".")>3 else k}:{rel(g[k].cpu(), params[k].grad):.2e}' for k in keys), flush=True)
