"""Where do the small device copies of a forward / training step come from?  Counts Tensor.copy_ / .to / .cuda /
.contiguous / .clone calls that touch CUDA tensors, by call site (diagnostic)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import synthetic as synth
from sacv_amd.detection import models, decode, losses
from sacv_amd.train import DetectionTrainer

mode = sys.argv[1] if len(sys.argv) > 1 else 'eval'
net = models.__dict__['resnet50_retinanet']().cuda()
x = torch.tensor(synth.images(0, 2, 800, 1333)).cuda()
ann = torch.tensor(synth.annotations(0, 2, 800, 1333, boxes_per_image=8)).cuda()
dec = decode.__dict__['RetinaDecoder']()
if mode == 'eval':
    net.eval()
    run = lambda: dec.device_call(net(x))
else:
    net.train()
    tr = DetectionTrainer(net, losses.__dict__['RetinaLoss'](), lr=1e-4, weight_decay=1e-3)
    run = lambda: tr.step(x, ann)
with torch.no_grad():
    run(); run()
counts = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda or any(isinstance(t, torch.Tensor) and t.is_cuda for t in a) or name in ('cuda',):
            fr = traceback.extract_stack(limit=4)[:-1]
            site = ' <- '.join(f'{os.path.basename(s.filename)}:{s.lineno}' for s in reversed(fr))
            counts[(name, site)] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ('copy_', 'to', 'cuda', 'contiguous', 'clone', 'zero_', 'fill_'):
    wrap(n)
for fn in ('zeros', 'tensor', 'full', 'ones', 'zeros_like'):
    orig = getattr(torch, fn)
    def g(*a, _o=orig, _n=fn, **k):
        r = _o(*a, **k)
        if isinstance(r, torch.Tensor) and r.is_cuda:
            fr = traceback.extract_stack(limit=4)[:-1]
            counts[('torch.' + _n, ' <- '.join(f'{os.path.basename(s.filename)}:{s.lineno}' for s in reversed(fr)))] += 1
        return r
    setattr(torch, fn, g)
with torch.no_grad():
    run()
torch.cuda.synchronize()
for (name, site), c in counts.most_common(25):
    print(f'{c:4d} {name:14s} {site}')
