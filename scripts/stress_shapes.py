"""Robustness sweep (not a parity test): one trainer step of RetinaNet / FCOS at several batch sizes and image
shapes; every loss must be finite and the step must not fault.  Shapes beyond the fixtures exercise the conv
planner (tile / split choices) and the 32-bit index guards."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import synthetic as synth
from sacv_amd.detection import models, losses
from sacv_amd.train import DetectionTrainer

cases = [(1, 800, 1333), (4, 800, 1333), (8, 800, 1333), (3, 640, 960), (2, 1333, 800), (5, 512, 512), (16, 416, 416),
         (2, 96, 128), (1, 64, 64), (2, 640, 640), (2, 1024, 1024), (3, 333, 517)]


def build(name, crit_name, side):
    os.environ['SACV_WGRAD_STREAM'] = '1' if side else '0'
    net = models.__dict__[name](num_classes=80).cuda().train()
    os.environ.pop('SACV_WGRAD_STREAM')
    return net, DetectionTrainer(net, losses.__dict__[crit_name](), lr=1e-5, weight_decay=1e-3)


for name, crit_name in (('resnet50_retinanet', 'RetinaLoss'), ('resnet50_fcos', 'FCOSLoss')):
    # the engine with the side stream and a stream-ordered twin from the same weights: besides finite losses, the two
    # must agree bit for bit at every shape (round 3: a missing cross-stream dependency would show here)
    net, tr = build(name, crit_name, True)
    net0, tr0 = build(name, crit_name, False)
    net0.load_state_dict(net.state_dict())
    for b, h, w in cases:
        x = torch.tensor(synth.images(b, b, h, w)).cuda()
        ann = torch.tensor(synth.annotations(b + 1, b, h, w, boxes_per_image=6)).cuda()
        fin = tr.finish_step(tr.step(x, ann))
        fin0 = tr0.finish_step(tr0.step(x, ann))
        torch.cuda.synchronize()
        ok = all(v == v and abs(v) < 1e6 for k, v in fin.items() if k.endswith('loss'))
        same = fin == fin0 and torch.equal(net.arena.grads, net0.arena.grads) and torch.equal(net.arena.params, net0.arena.params)
        print(name, (b, h, w), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in fin.items()},
              'OK' if ok else 'BAD', 'bit-identical to the stream-ordered engine' if same else 'DIFFERS', flush=True)
        assert ok and same
print('stress sweep passed')
