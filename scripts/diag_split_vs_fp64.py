"""RetinaNet-R50 training step at 544 x 736: gradients of the fp32-pipe engine and of the split-operand engine against
the fp64 oracle.  Is the split engine CONSISTENTLY farther from the exact gradient (a defect) or as far as the fp32
engine (the two differ by what fp32 rounding does to ReLU masks and sparse regression gradients)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth, model as omodel, losses as olosses
from sacv_amd.detection import models, losses


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


H, W = 544, 736
state0 = omodel.init_retinanet_state(0)
keys = ['cls_head.cls_head.0.weight', 'reg_head.reg_head.0.weight', 'fpn.P7.1.bias', 'fpn.P6.bias', 'fpn.P6.weight',
        'fpn.P4_1.weight', 'fpn.P3_1.weight', 'backbone.layer4.2.conv3.layer.0.weight', 'backbone.layer1.0.conv3.layer.0.weight']
for seed in (7, 21):
    images = torch.tensor(synth.images(seed, 2, H, W)).cuda()
    annots = torch.tensor(synth.annotations(seed + 1, 2, H, W, boxes_per_image=6)).cuda()
    sd64 = {k: (v.double().cpu() if v.dtype.is_floating_point else v.clone().cpu()) for k, v in state0.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.dtype.is_floating_point and 'running_' not in k}
    st = dict(sd64)
    st.update(params)
    ld = olosses.RetinaLossOracle()(omodel.retinanet_forward(st, images.cpu().double(), training=True), annots.cpu())
    sum(ld.values()).backward()
    print(f'seed {seed}: oracle loss {float(sum(ld.values())):.7f}', flush=True)
    for label, cs, ws, wino in (('fp32 pipe            ', False, False, True), ('split convs         ', True, False, True),
                                ('split convs + heads ', True, True, True), ('fp32 pipe, no Winograd', False, False, False)):
        net = models.resnet50_retinanet(num_classes=80)
        net.load_state_dict(state0)
        net = net.cuda().train()
        net.engine.conv_split, net.engine.wino_split, net.engine.winograd = cs, ws, wino
        l = losses.RetinaLoss()(net(images), annots)
        sum(l.values()).backward()
        torch.cuda.synchronize()
        g = net.arena.grad_views
        allr = sorted(rel(g[k].cpu(), params[k].grad) for k in params if float(params[k].grad.norm()) > 0)
        print(seed, label, f'loss {float(sum(l.values())):.7f} median {allr[len(allr) // 2]:.2e} max {allr[-1]:.2e} |',
              ' '.join(f'{k.split(".")[-3][-6:]}.{k.split(".")[-2]}.{k.split(".")[-1][0]}:{rel(g[k].cpu(), params[k].grad):.1e}' for k in keys), flush=True)
