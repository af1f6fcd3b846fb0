"""Memory / stability soak of the training step: N steps at the headline shape, then steps at changing canvases (the
multi-scale recipes), printing the allocator's reserved bytes -- it must plateau."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import synthetic as synth
from sacv_amd.detection import models, losses
from sacv_amd.train import DetectionTrainer

net = models.__dict__['resnet50_retinanet']().cuda().train()
tr = DetectionTrainer(net, losses.__dict__['RetinaLoss'](), lr=1e-4, weight_decay=1e-3)
shapes = [(800, 1333)] * 120 + [(640, 640), (704, 1024), (800, 1216), (512, 768), (800, 1333), (736, 1120)] * 10
batches = {}
t0 = time.time()
for i, (h, w) in enumerate(shapes):
    if (h, w) not in batches:
        batches[(h, w)] = (torch.tensor(synth.images(i, 2, h, w)).cuda(),
                           torch.tensor(synth.annotations(i, 2, h, w, boxes_per_image=8)).cuda())
    out = tr.step(*batches[(h, w)])
    if i in (5, 60, 119, 130, 150, 179):
        fin = tr.finish_step(out)
        torch.cuda.synchronize()
        print(f'step {i:4d} {h}x{w}: loss {fin["loss"]:.4f} reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB '
              f'allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  t={time.time() - t0:.1f}s', flush=True)
