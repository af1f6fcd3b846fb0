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


def _ddp_worker(rank, world, port, ret):
    """Class-level path: the model wrapped in torch's DistributedDataParallel exactly as the reference's
    build_training_mode does (tools/utils.py:178-180), loss.backward() through autograd."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import model as omodel, synth
        from sacv_amd.detection import models, losses
        H, W = 96, 128
        net = models.__dict__['resnet50_retinanet'](num_classes=80)
        net.load_state_dict(omodel.init_retinanet_state(rank))  # DDP must overwrite rank 1 with rank 0's weights
        net = net.cuda().train()
        ddp = torch.nn.parallel.DistributedDataParallel(net)
        crit = losses.__dict__['RetinaLoss']()
        images = torch.tensor(synth.images(100 + rank, 2, H, W)).cuda()
        annots = torch.tensor(synth.annotations(200 + rank, 2, H, W, boxes_per_image=8)).cuda()
        # each rank's own gradient first (no_sync), then the synchronised one
        with ddp.no_sync():
            ld = crit(ddp(images), annots)
            sum(ld.values()).backward()
        own = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        for p in net.parameters():
            p.grad = None
        ld = crit(ddp(images), annots)
        sum(ld.values()).backward()
        worst, report = 0.0, []
        for n, p in net.named_parameters():
            tot = own[n].clone()
            dist.all_reduce(tot)
            want = tot / world  # DDP averages
            scale = float(want.abs().max()) + 1e-12
            e = float((p.grad - want).abs().max()) / scale
            if e > 1e-5 and len(report) < 6:
                report.append((n, e, float((p.grad - own[n]).abs().max()) / scale, tuple(p.grad.stride()),
                               tuple(own[n].stride())))
            worst = max(worst, e)
        if report:
            print('DDP mismatch', rank, report, flush=True)
        # BatchNorm statistics differ per rank's batch; the gradients of two consecutive forwards differ only through
        # nothing (same weights, same inputs, train-mode batch statistics): rounding level
        assert worst <= 1e-5, worst
        w0 = net.backbone.layer1[0].conv2.layer[0].weight.detach().clone()
        dist.broadcast(w0, src=0)
        assert torch.equal(w0, net.backbone.layer1[0].conv2.layer[0].weight.detach())
        ret[rank] = float(ld['cls_loss'])
    finally:
        dist.destroy_process_group()


def test_model_wrapped_in_torch_ddp_shared_gpu():
    """INTEGRATION.md's claim that nn.parallel.DistributedDataParallel(model) works on the class-level path."""
    assert torch.cuda.is_available()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert set(ret.keys()) == {0, 1}


def test_rccl_communicator_of_the_c_abi_world1():
    """sacv_comm_* / sacv_allreduce_bucket with the real RCCL on the one GPU of this box: rendezvous id, communicator,
    a bucket all-reduce on the side stream chained by events to the launch stream, the join, a broadcast."""
    from sacv_amd.parallel import RcclComm, GradExchanger
    comm = RcclComm()
    assert comm.world == 1 and comm.rank == 0
    g = torch.arange(1 << 20, dtype=torch.float32, device='cuda')
    want = g.clone()
    ex = GradExchanger(g, {'heads': (1 << 19, 1 << 20), 'rest': (0, 1 << 19)}, comm=comm)
    ex.world = 2  # force the exchange path (a 1-rank sum is the identity)
    torch.mul(g, 2.0, out=g)  # producer work on the launch stream that the side stream must wait for
    ex.stage_done('heads')
    ex.stage_done('rest')
    ex.wait()
    torch.cuda.synchronize()
    assert torch.equal(g, want * 2.0)
    comm.broadcast(g[:1024])
    torch.cuda.synchronize()
    comm.close()


def _one_comm_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import model as omodel, synth
        from sacv_amd.detection import models, losses
        from sacv_amd.train import DetectionTrainer
        real_all_reduce, real_broadcast = dist.all_reduce, dist.broadcast

        class HostComm:
            """RcclComm's surface over gloo (in order, synchronous): stands in for the C ABI's communicator."""
            world, rank = dist.get_world_size(), dist.get_rank()
            calls = 0

            def all_reduce(self, t):
                HostComm.calls += 1
                real_all_reduce(t)

            def wait(self):
                pass

            def broadcast(self, t, root=0):
                real_broadcast(t, src=root)

        def forbidden(*a, **k):
            raise AssertionError('a collective on torch.distributed while the trainer owns a communicator')

        H, W = 96, 128
        net = models.__dict__['resnet50_retinanet'](num_classes=80)
        net.load_state_dict(omodel.init_retinanet_state(rank))
        net = net.cuda().train()
        dist.all_reduce = dist.broadcast = forbidden
        try:
            tr = DetectionTrainer(net, losses.__dict__['RetinaLoss'](), process_group=dist.group.WORLD,
                                  transport=HostComm())
            outs = []
            for it in range(2):
                out = tr.step(torch.tensor(synth.images(100 + rank + 10 * it, 2, H, W)).cuda(),
                              torch.tensor(synth.annotations(200 + rank + 10 * it, 2, H, W, boxes_per_image=8)).cuda())
                outs.append(tr.finish_step(out))
        finally:
            dist.all_reduce, dist.broadcast = real_all_reduce, real_broadcast
        assert HostComm.calls == 2 * (1 + 4 + 1)  # per step: skip vote, four gradient stages, loss scalars
        p = net.arena.params.clone()
        ref = p.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(p, ref), 'parameters diverged between ranks'
        ret[rank] = (outs[1]['loss'], outs[1]['optimizer_steps'], float(p.double().sum()))
    finally:
        dist.destroy_process_group()


def test_every_collective_of_a_step_goes_through_the_one_communicator():
    """with transport 'rccl' the skip vote and the loss scalars used to travel on torch's communicator
    while the gradient stages were in flight on the C ABI's.  Now every collective of a step goes through the
    trainer's communicator; rehearsed with two ranks on this GPU and a gloo stand-in with RcclComm's surface."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_one_comm_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert set(ret.keys()) == {0, 1} and ret[0] == ret[1] and ret[0][1] == 2


def _nccl_worker(rank, world, port, ret, transport):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    try:
        from oracle import model as omodel, synth
        from sacv_amd.detection import models, losses
        from sacv_amd.train import DetectionTrainer
        H, W = 96, 128
        net = models.__dict__['resnet50_retinanet'](num_classes=80)
        net.load_state_dict(omodel.init_retinanet_state(rank))
        net = net.cuda().train()
        tr = DetectionTrainer(net, losses.__dict__['RetinaLoss'](), process_group=dist.group.WORLD,
                              transport=transport)
        out = tr.step(torch.tensor(synth.images(100 + rank, 2, H, W)).cuda(),
                      torch.tensor(synth.annotations(200 + rank, 2, H, W, boxes_per_image=8)).cuda())
        fin = tr.finish_step(out)
        p = net.arena.params.clone()
        ref = p.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(p, ref), 'parameters diverged between ranks'
        ret[rank] = (fin['loss'], float(p.double().sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('transport', ['torch', 'rccl'])
def test_two_rank_step_over_rccl(transport):
    """Real RCCL (backend "nccl"), one GPU per rank: needs two GPUs, skipped on a one-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs (the driver scaling run exercises this path on the 8-GPU node)')
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_nccl_worker, args=(2, _free_port(), ret, transport), nprocs=2, join=True)
    assert ret[0] == ret[1]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain shell must start the two ranks itself and print ONE JSON line with
    n_gpus 2 (rehearsed on this one GPU: both ranks share cuda:0, collectives through gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SACV_SHARE_GPU='1', SACV_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['value'] > 0 and d['scaling'] == 'weak'
