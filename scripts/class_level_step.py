"""Step time of integration level 1: the reference's own loop statements -- model(images), criterion,
sum(loss).backward(), torch.optim.AdamW in the reference's two groups -- on the HIP model, next to the fused trainer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import synthetic as synth
from sacv_amd.detection import models, losses
from sacv_amd.train import DetectionTrainer

H, W = 800, 1333
images = torch.tensor(synth.images(0, 2, H, W)).cuda()
annots = torch.tensor(synth.annotations(0, 2, H, W, boxes_per_image=8)).cuda()

def timed(step, n=20, warm=5):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

net = models.__dict__['resnet50_retinanet']().cuda().train()
crit = losses.__dict__['RetinaLoss']().cuda()
p1 = [p for p in net.parameters() if p.ndim == 1]
pn = [p for p in net.parameters() if p.ndim != 1]
opt = torch.optim.AdamW([{'params': p1, 'weight_decay': 0.0}, {'params': pn, 'weight_decay': 1e-3}], lr=1e-4)

def class_level():
    loss = sum(crit(net(images), annots).values())
    loss.backward()
    opt.step()
    opt.zero_grad()

print('class level (reference loop statements, torch.optim.AdamW): %.2f ms / step' % timed(class_level))
net2 = models.__dict__['resnet50_retinanet']().cuda().train()
tr = DetectionTrainer(net2, losses.__dict__['RetinaLoss'](), lr=1e-4, weight_decay=1e-3)
print('fused trainer: %.2f ms / step' % timed(lambda: tr.step(images, annots)))
