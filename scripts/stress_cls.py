import sys; sys.path.insert(0,'/root/repo')
import torch
from sacv_amd import synthetic as synth
from sacv_amd.classification import backbones, losses
from sacv_amd.train import ClassificationTrainer
net = backbones.__dict__['resnet50'](num_classes=1000).cuda().train()
tr = ClassificationTrainer(net, losses.__dict__['CELoss'](), lr=0.01)
for b, r in ((1, 224), (7, 224), (32, 192), (64, 256), (128, 224), (320, 224), (16, 320)):
    x = torch.tensor(synth.images(b, min(b, 8), r, r)).cuda()
    x = x.repeat((b + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:b].contiguous()
    lab = (torch.arange(b) % 1000).cuda()
    fin = tr.finish_step(tr.step(x, lab)); torch.cuda.synchronize()
    print((b, r), fin, flush=True)
    assert fin['loss'] == fin['loss'] and fin['loss'] < 100
try:
    x = torch.zeros(512, 3, 224, 224).cuda(); tr.step(x, torch.zeros(512, dtype=torch.long).cuda())
    print('bs 512 ran')
except Exception as e:
    print('bs 512:', type(e).__name__, str(e)[:120])
