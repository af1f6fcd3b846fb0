"""Host-side enqueue time of one training step (no synchronisation) vs its GPU time, with a cProfile of the host path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sacv_amd import synthetic as synth
from sacv_amd.detection import models, losses
from sacv_amd.train import DetectionTrainer
net = models.__dict__['resnet50_retinanet']().cuda().train()
tr = DetectionTrainer(net, losses.__dict__['RetinaLoss']())
x = torch.tensor(synth.images(0, 2, 800, 1333)).cuda(); a = torch.tensor(synth.annotations(0, 2, 800, 1333)).cuda()
for _ in range(3): tr.step(x, a)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.step(x, a); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print('host enqueue / step wall (ms):', ['%.1f / %.1f' % t for t in ts])
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.step(x, a)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(30)
