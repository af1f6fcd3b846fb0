"""Diagnostic: per-parameter gradient error of the GPU step vs the CPU oracle in fp32 and fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import losses as olosses, model as omodel, synth
from tests import golden_inputs as gi
from sacv_amd.detection import models, losses

tag = sys.argv[1] if len(sys.argv) > 1 else 'TINY'
H, W = gi.SHAPES[tag]
sd = omodel.init_retinanet_state(0)
images = torch.tensor(synth.images(5, 2, H, W))
annots = torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8))


def oracle(dtype):
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    state.update(params)
    outs = omodel.retinanet_forward(state, images.to(dtype), training=True)
    outs32 = [[o.float() for o in g] for g in outs] if dtype == torch.float64 else outs
    ld = olosses.RetinaLossOracle()(outs, annots) if dtype == torch.float32 else None
    if ld is None:
        # fp64 run: loss oracle works in the input dtype for preds; anchors/targets stay fp32
        ld = olosses.RetinaLossOracle()(outs, annots.to(dtype))
    sum(ld.values()).backward()
    return {k: p.grad.double() for k, p in params.items()}, {k: float(v) for k, v in ld.items()}

g32, l32 = oracle(torch.float32)
try:
    g64, l64 = oracle(torch.float64)
except Exception as e:
    print('fp64 oracle failed:', e)
    g64, l64 = None, None
net = models.__dict__['resnet50_retinanet'](num_classes=80)
net.load_state_dict(sd)
net = net.cuda().train()
crit = losses.__dict__['RetinaLoss']()
outs = net(images.cuda())
ld = crit(outs, annots.cuda())
sum(ld.values()).backward()
print('loss gpu', {k: float(v) for k, v in ld.items()}, 'cpu32', l32, 'cpu64', l64)
for name, p in net.named_parameters():
    g = p.grad.double().cpu()
    ref = g64[name] if g64 else g32[name]
    den = float(ref.abs().max()) + 1e-30
    e_gpu = float((g - ref).abs().max()) / den
    e_cpu = float((g32[name] - ref).abs().max()) / den
    print(f'{name:55s} gpu_err {e_gpu:.2e}  cpu32_err {e_cpu:.2e}  max {den:.2e}')
