"""Two-rank data-parallel training step on ONE GPU (both ranks on cuda:0, gloo collectives through the
host): rehearses the RCCL path of train.py/parallel.py end to end -- parameter/buffer broadcast at
construction, staged asynchronous gradient all-reduce during backward, 1/world folded into AdamW -- and
checks the DDP invariants: ranks start and stay bit-identical although they see different batches, and
the update equals the single-process update computed from the mean of the two ranks' gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import model as omodel, synth
        from sacv_amd.detection import models, losses
        from sacv_amd.train import DetectionTrainer
        H, W = 96, 128
        net = models.__dict__['resnet50_retinanet'](num_classes=80)
        net.load_state_dict(omodel.init_retinanet_state(rank))  # DIFFERENT initial weights per rank
        net = net.cuda().train()
        trainer = DetectionTrainer(net, losses.__dict__['RetinaLoss'](), lr=1e-4, weight_decay=1e-3,
                                   process_group=dist.group.WORLD)
        arena = net.arena
        p0 = arena.params.clone()
        # after construction every rank holds rank 0's parameters and BN buffers
        ref0 = p0.clone()
        dist.broadcast(ref0, src=0)
        assert torch.equal(p0, ref0)
        images = torch.tensor(synth.images(100 + rank, 2, H, W)).cuda()
        annots = torch.tensor(synth.annotations(200 + rank, 2, H, W, boxes_per_image=8)).cuda()
        out = trainer.step(images, annots)
        torch.cuda.synchronize()
        # the arena now holds the SUM of both ranks' gradients on every rank
        g = arena.grads.clone()
        gref = g.clone()
        dist.broadcast(gref, src=0)
        assert torch.equal(g, gref), 'summed gradients differ between ranks'
        p1 = arena.params.clone()
        pref = p1.clone()
        dist.broadcast(pref, src=0)
        assert torch.equal(p1, pref), 'parameters diverged between ranks'
        # first Adam step from zero moments with the mean gradient: p1 = p0*(1-lr*wd) - lr*sign-like update
        gm = g / world
        wd = torch.zeros_like(p0)
        wd[arena.split:] = 1e-3
        m = 0.1 * gm
        v = 0.001 * gm * gm
        upd = (m / (1 - 0.9)) / ((v / (1 - 0.999)).sqrt() + 1e-8)
        expect = p0 * (1 - 1e-4 * wd) - 1e-4 * upd
        assert float((p1 - expect).abs().max()) <= 2e-7 + 1e-6 * float(expect.abs().max())
        fin = trainer.finish_step(out)
        ret[rank] = (fin['loss'], float(p1.double().sum()))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_shared_gpu():
    assert torch.cuda.is_available()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    assert ret[0] == ret[1]  # averaged loss and parameter checksum agree bit for bit
