"""Is the launch stream ever IDLE inside an un-profiled training step (host late, or a cross-stream wait)?  Every C-ABI
call of one step is bracketed by two events on the stream it is enqueued on; an event's timestamp is taken when the queue
reaches it, so elapsed(end of call k, begin of call k+1 on the same stream) is time that stream spent with nothing to run
(or blocked in a wait).  The events cost a few us each: only gaps of tens of us mean anything."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sacv_amd  # noqa
from oracle import synth
from sacv_amd import _lib
from sacv_amd.detection import models, losses
from sacv_amd.train import DetectionTrainer

log = []
on = [False]
lib = _lib.lib
for name in _lib.SIGNATURES:
    fn = getattr(lib, name)

    def wrap(fn=fn, name=name):
        def w(*a):
            if not on[0]:
                return fn(*a)
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = fn(*a)
            e1.record(s)
            log.append((name, s.cuda_stream, e0, e1))
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
for _ in range(6):
    tr.step(images, annots)
on[0] = True
tr.step(images, annots)   # a step whose predecessor is still running when the host starts it
on[0] = False
torch.cuda.synchronize()
by = collections.defaultdict(list)
for rec in log:
    by[rec[1]].append(rec)
t_first = log[0][2]
for sid, recs in by.items():
    busy = sum(r[2].elapsed_time(r[3]) for r in recs)
    gaps = [(recs[i][3].elapsed_time(recs[i + 1][2]), recs[i][0], recs[i + 1][0], t_first.elapsed_time(recs[i][3]))
            for i in range(len(recs) - 1)]
    print(f'stream {sid:#x}: {len(recs)} calls, inside the calls {busy:.2f} ms, between them {sum(g[0] for g in gaps):.2f} ms, '
          f'span {recs[0][2].elapsed_time(recs[-1][3]):.2f} ms')
    for g, a, b, t in sorted(gaps, reverse=True)[:12]:
        print(f'   {1e3 * g:8.0f} us idle after {a} (ends at {t:7.2f} ms) before {b}')
