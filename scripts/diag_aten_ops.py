"""Which aten operators (torch kernels / hipMemcpy calls that are NOT the library's) run inside one training step, and
from which line of this package?  TorchDispatchMode over one DetectionTrainer.step after warm-up."""
import sys, os, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import sacv_amd  # noqa
from oracle import synth
from sacv_amd.detection import models, losses
from sacv_amd.train import DetectionTrainer

PKG = os.path.join(ROOT, 'simpleaicv-pytorch-imagenet-coco-training_amd')
VIEWS = ('view', 'as_strided', 'detach', 'alias', 'slice', 'select', 'unsqueeze', 'squeeze', 'permute', 'transpose',
         'expand', 'reshape', '_unsafe_view', 't.', 'unbind', 'split', 'narrow', 'is_', 'sym_', 'stride', 'size', 'numel',
         'dim', 'storage_offset', 'empty', 'lift_fresh', '_local_scalar_dense', 'record_stream')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            site = '?'
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(PKG) or fr.filename.endswith('bench.py'):
                    site = f'{os.path.relpath(fr.filename, ROOT).split("/", 1)[-1]}:{fr.lineno}'
                    break
            self.count[(name, site)] += 1
        return func(*args, **(kwargs or {}))


fcos = len(sys.argv) > 1 and sys.argv[1] == 'fcos'
H, W = 800, 1333
net = (models.resnet50_fcos if fcos else models.resnet50_retinanet)(backbone_pretrained_path='', num_classes=80).cuda().train()
tr = DetectionTrainer(net, (losses.FCOSLoss if fcos else losses.RetinaLoss)(), lr=1e-4, weight_decay=1e-3)
images = torch.tensor(synth.images(0, 2, H, W)).cuda()
annots = torch.tensor(synth.annotations(0, 2, H, W, boxes_per_image=8)).cuda()
for _ in range(4):
    tr.step(images, annots)
torch.cuda.synchronize()
with Log() as lg:
    tr.step(images, annots)
torch.cuda.synchronize()
tot = sum(lg.count.values())
print(f'{tot} non-view aten calls in one step')
for (name, site), n in lg.count.most_common(60):
    print(f'{n:4d}  {name:40s} {site}')
