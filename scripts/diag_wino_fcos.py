"""FCOS: Winograd-path engine vs direct-kernel engine, top differing gradient tensors (run per library build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from sacv_amd.detection import models, losses

def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
H, W = 544, 736
images = torch.tensor(synth.images(7, 2, H, W)).cuda()
annots = torch.tensor(synth.annotations(8, 2, H, W, boxes_per_image=6)).cuda()
torch.manual_seed(0)
state0 = {k: v.clone() for k, v in models.__dict__['resnet50_fcos'](num_classes=80).state_dict().items()}
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
    return {k: g.clone() for k, g in net.arena.grad_views.items()}
for label, st in (('default', {}), ('fused_stem=eval', dict(fused_stem='eval')), ('fuse_bn_bwd=False', dict(fuse_bn_bwd=False))):
    try:
        gw, gd = run(True, **st), run(False, **st)
    except AttributeError as e:
        print(label, 'n/a', e); continue
    worst = sorted(((rel(gw[k], gd[k]), k) for k in gd if float(gd[k].double().norm()) > 0), reverse=True)[:4]
    print(label, [(f'{e:.2e}', k) for e, k in worst], flush=True)

# both engines against the CPU oracle in fp64 (which one moved, and how far is fp32 from the exact gradient here?)
from oracle import model as omodel, losses as olosses
sd64 = {k: (v.double().cpu() if v.dtype.is_floating_point else v.clone().cpu()) for k, v in state0.items()}
params = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.dtype.is_floating_point and 'running_' not in k}
st = dict(sd64); st.update(params)
outs = omodel.fcos_forward(st, images.cpu().double(), training=True)
ld = olosses.FCOSLossOracle()(outs, annots.cpu())
sum(ld.values()).backward()
keys = ['fpn.P3_1.weight', 'fpn.P3_2.weight', 'fpn.P3_1.bias', 'fpn.P4_2.weight', 'clsregcnt_head.cls_head.0.weight',
        'clsregcnt_head.reg_head.9.weight', 'backbone.layer3.0.conv1.layer.0.weight']
for label, wino, stg in (('wino  fused ', True, {}), ('direct fused ', False, {}), ('wino  im2col', True, dict(fused_stem='eval')),
                         ('direct im2col', False, dict(fused_stem='eval'))):
    g = run(wino, **stg)
    print(label, ' '.join(f'{k}:{rel(g[k].cpu(), params[k].grad):.2e}' for k in keys), flush=True)
