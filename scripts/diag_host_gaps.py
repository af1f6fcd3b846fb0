"""Where does the HOST spend its time while it enqueues one training step?  Wraps every sacv_amd.ops function and every
C-ABI call with a timestamp (no synchronisation) and prints the largest gaps between consecutive enqueues."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sacv_amd  # noqa
from oracle import synth
from sacv_amd import ops, _lib
from sacv_amd.detection import models, losses
from sacv_amd.train import DetectionTrainer

log = []
lib = _lib.lib
for name in _lib.SIGNATURES:
    fn = getattr(lib, name)

    def wrap(fn=fn, name=name):
        def w(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            log.append((name, t0, time.perf_counter()))
            return r
        return w
    try:
        setattr(lib, name, wrap())
    except Exception:
        pass
H, W = 800, 1333
net = models.resnet50_retinanet(backbone_pretrained_path='', num_classes=80).cuda().train()
tr = DetectionTrainer(net, losses.RetinaLoss(), lr=1e-4, weight_decay=1e-3)
images = torch.tensor(synth.images(0, 2, H, W)).cuda()
annots = torch.tensor(synth.annotations(0, 2, H, W, boxes_per_image=8)).cuda()
for _ in range(5):
    tr.step(images, annots)
torch.cuda.synchronize()
del log[:]
t_begin = time.perf_counter()
out = tr.step(images, annots)
t_enq = time.perf_counter()
torch.cuda.synchronize()
t_end = time.perf_counter()
print(f'host enqueue of one step {1e3 * (t_enq - t_begin):.2f} ms, step complete after {1e3 * (t_end - t_begin):.2f} ms, {len(log)} C-ABI calls')
inside = sum(e - s for _, s, e in log)
print(f'time inside the C-ABI calls {1e3 * inside:.2f} ms; python between them {1e3 * (t_enq - t_begin - inside):.2f} ms')
gaps = sorted(((log[i + 1][1] - log[i][2], log[i][0], log[i + 1][0], log[i][2] - t_begin) for i in range(len(log) - 1)), reverse=True)
for g, a, b, t in gaps[:15]:
    print(f'  at {1e3 * t:7.2f} ms: {1e6 * g:7.0f} us between {a} and {b}')
slow = sorted(((e - s, n, s - t_begin) for n, s, e in log), reverse=True)
for d, n, t in slow[:8]:
    print(f'  call {n} took {1e6 * d:6.0f} us at {1e3 * t:7.2f} ms')
