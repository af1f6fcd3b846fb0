"""Which gradient tensors differ between conv_split on / off (same model, same input)?  Printed in module order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sacv_amd  # noqa
from oracle import model as omodel, synth
from sacv_amd.detection import models, losses

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (544, 736)
images = torch.tensor(synth.images(7, 2, H, W)).cuda()
annots = torch.tensor(synth.annotations(8, 2, H, W, boxes_per_image=6)).cuda()
state0 = omodel.init_retinanet_state(0)
out = {}
PAIR = [x == '1' for x in (sys.argv[3] if len(sys.argv) > 3 else '01')]
for idx, cs in enumerate(PAIR):
    net = models.resnet50_retinanet(num_classes=80)
    net.load_state_dict(state0)
    net = net.cuda().train()
    net.engine.wino_split = False
    net.engine.conv_split = cs
    crit = losses.RetinaLoss()
    preds = net(images)
    ld = crit(preds, annots)
    loss = sum(ld.values())
    loss.backward()
    torch.cuda.synchronize()
    out[idx] = (float(loss.detach()), {k: g.clone() for k, g in net.arena.grad_views.items()},
               [p.detach().clone() for lv in preds for p in (lv if isinstance(lv, (list, tuple)) else [lv])])
print('loss', out[0][0], out[1][0])
for a, b in zip(out[0][2], out[1][2]):
    print('pred rel diff', float((a - b).norm() / a.norm()))
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
for k in out[0][1]:
    r = rel(out[1][1][k], out[0][1][k])
    if k.startswith('fpn') or 'out' in k or k.endswith('conv3.layer.0.weight'):
        print(f'{k:60s} {r:.2e}')
