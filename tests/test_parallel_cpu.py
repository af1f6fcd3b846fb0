"""world_size-2 gloo tests (CPU) of the data-parallel exchange used by the trainer: staged asynchronous
all-reduce over zero-copy slices of the flat gradient arena, rank-0 buffer broadcast, fused scalar vote."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sacv_amd.detection import models
        from sacv_amd.parallel import GradExchanger, stage_ranges, broadcast_buffers, vote_and_reduce_scalars
        m = models.__dict__['resnet50_retinanet']()
        # arena layout without touching the GPU: reuse the arena's ordering logic on CPU parameters
        from sacv_amd.detection.models._arena import ParamArena
        arena = ParamArena(m)
        ranges = stage_ranges(arena.order, arena.offsets, arena.numel, arena.split)
        # stages tile the arena exactly, in backward order heads -> neck -> rest
        assert ranges['rest'][0] == 0 and ranges['rest'][1] == ranges['mid'][0]
        assert ranges['mid'] == (arena.offsets['backbone.layer3.0.conv1.layer.0.weight'], ranges['neck'][0])
        assert ranges['neck'][1] == ranges['heads'][0] and ranges['heads'][1] == arena.numel
        for n, _ in arena.order:
            o = arena.offsets[n]
            st = 'heads' if ('head' in n and arena.module_params[n].ndim != 1) else None
            if st:
                assert ranges['heads'][0] <= o < ranges['heads'][1], n
        g = arena.grads
        g.copy_(torch.arange(g.numel(), dtype=torch.float32) % 97 + rank)
        ex = GradExchanger(g, ranges)
        for stage in ('heads', 'neck', 'mid', 'rest'):
            ex.stage_done(stage)
        ex.wait()
        expect = (torch.arange(g.numel(), dtype=torch.float32) % 97) * world + sum(range(world))
        assert torch.equal(g, expect)
        # p.grad views alias the arena (zero copy)
        arena.attach_grads()
        name, p = arena.order[-1]
        assert p.grad.data_ptr() == arena.grad_views[name].data_ptr()
        # buffer broadcast: rank 0 wins
        arena.buffers.fill_(float(rank + 1))
        broadcast_buffers(arena.buffers)
        assert float(arena.buffers.min()) == 1.0 and float(arena.buffers.max()) == 1.0
        assert float(m.backbone.conv1.layer[1].running_mean[0]) == 1.0  # module buffers are arena views
        v = vote_and_reduce_scalars(torch.tensor([1.0 + rank, 0.5]))
        assert v.tolist() == [sum(1.0 + r for r in range(world)), 0.5 * world]

        # transport 'rccl' (ADVICE r2): with a communicator of the C ABI EVERY collective of a step -- stages, skip vote,
        # loss scalars, buffer broadcast -- must go through it, none through torch's group.  A recording stand-in with
        # RcclComm's surface (all_reduce / wait / broadcast, in-order) over gloo proves the routing on CPU.
        class Recorder:
            def __init__(self):
                self.world, self.rank, self.log, self.outstanding = world, rank, [], 0

            def all_reduce(self, t):
                self.log.append(('all_reduce', t.numel()))
                self.outstanding += 1
                dist.all_reduce(t)

            def wait(self):
                self.log.append(('wait', self.outstanding))
                self.outstanding = 0

            def broadcast(self, t, root=0):
                self.log.append(('broadcast', t.numel()))
                dist.broadcast(t, src=root)

        comm = Recorder()
        real_all_reduce, real_broadcast = dist.all_reduce, dist.broadcast
        g.copy_(torch.arange(g.numel(), dtype=torch.float32) % 89 + rank)
        ex = GradExchanger(g, ranges, comm=comm)
        flag = torch.tensor([float(rank)])
        comm.all_reduce(flag)  # the trainer's _vote_begin, ahead of the stages on the same communicator
        for stage in ('heads', 'neck', 'mid', 'rest'):
            ex.stage_done(stage)
        comm.wait()            # _after_backward: vote + every stage
        ex.wait()
        assert torch.equal(g, (torch.arange(g.numel(), dtype=torch.float32) % 89) * world + sum(range(world)))
        assert flag.item() == sum(range(world))
        v = vote_and_reduce_scalars(torch.tensor([2.0 + rank, 0.25]), comm=comm)
        assert v.tolist() == [sum(2.0 + r for r in range(world)), 0.25 * world]
        arena.buffers.fill_(float(rank + 7))
        broadcast_buffers(arena.buffers, comm=comm)
        assert float(arena.buffers.min()) == 7.0 and float(arena.buffers.max()) == 7.0
        kinds = [k for k, _ in comm.log]
        assert kinds == ['all_reduce'] * 5 + ['wait', 'wait', 'all_reduce', 'wait', 'broadcast'], kinds
        assert comm.log[5] == ('wait', 5) and comm.log[6] == ('wait', 0)
        ret[rank] = 'ok'
    finally:
        dist.destroy_process_group()


def test_grad_exchange_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 'ok', 1: 'ok'}


def test_stage_ranges_of_the_classifier_arena():
    """ResNet50 classifier: fc is the 'heads' stage, layer4 the 'neck', everything else 'rest'."""
    from sacv_amd.classification import backbones
    from sacv_amd.detection.models._arena import ParamArena
    from sacv_amd.parallel import stage_ranges
    arena = ParamArena(backbones.__dict__['resnet50'](num_classes=1000))
    r = stage_ranges(arena.order, arena.offsets, arena.numel, arena.split)
    assert r['rest'][0] == 0 and r['rest'][1] == r['mid'][0] and r['mid'][1] == r['neck'][0]
    assert r['neck'][1] == r['heads'][0] and r['mid'][0] == arena.offsets['layer3.0.conv1.layer.0.weight']
    assert r['heads'] == (arena.offsets['fc.weight'], arena.numel)
    assert r['neck'][0] == arena.offsets['layer4.0.conv1.layer.0.weight']
    assert arena.split % 4 == 0 and arena.numel % 4 == 0
