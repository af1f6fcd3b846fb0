"""The side HIP stream of the engines (weight gradients, bias-gradient column sums, the shortcut branch of the forward
pass, the data-gradient panel refresh) must not change a single bit: the same kernels run with the same operands, only
on another queue.  Any missing dependency (a weight gradient reading a dy that the launch stream is still writing, a
scratch buffer shared by two streams) shows up here as a difference between the two modes or between repetitions."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth
from tests import golden_inputs as gi


def _build(kind, side):
    from sacv_amd.detection import models, losses
    fcos = kind == 'fcos'
    net = models.__dict__['resnet50_fcos' if fcos else 'resnet50_retinanet'](num_classes=80)
    net.load_state_dict(omodel.init_retinanet_state(0, fcos=fcos))
    net = net.cuda().train()
    net.engine.side_stream = side  # False: the stream-ordered twin
    assert (net.engine._side is not None) == side
    crit = losses.__dict__['FCOSLoss' if fcos else 'RetinaLoss']()
    return net, crit


@pytest.mark.parametrize('kind', ['retina', 'fcos'])
def test_side_stream_changes_no_bit(kind):
    from sacv_amd.train import DetectionTrainer
    H, W = 256, 320
    batches = [(torch.tensor(synth.images(30 + i, 2, H, W)).cuda(),
                torch.tensor(synth.annotations(40 + i, 2, H, W, boxes_per_image=8)).cuda()) for i in range(3)]
    results = {}
    for side in (False, True):
        net, crit = _build(kind, side)
        tr = DetectionTrainer(net, crit, lr=1e-4, weight_decay=1e-3)
        grads, losses_ = [], []
        for x, a in batches:
            out = tr.step(x, a)
            torch.cuda.synchronize()
            grads.append(net.arena.grads.clone())
            losses_.append({k: float(v) for k, v in out.items()})
        results[side] = (grads, losses_, net.arena.params.clone(), net.arena.buffers.clone())
    for i in range(3):
        assert results[False][1][i] == results[True][1][i], i
        assert torch.equal(results[False][0][i], results[True][0][i]), f'gradients of step {i} differ'
    assert torch.equal(results[False][2], results[True][2]) and torch.equal(results[False][3], results[True][3])


def test_side_stream_repeats_bit_for_bit_at_full_size():
    """Full 800 x 1333 step, side stream on, five repetitions from the same state: identical gradients each time (a
    race would need luck to repeat), and identical to the stream-ordered engine."""
    H, W = 800, 1333
    x = torch.tensor(synth.images(5, 2, H, W)).cuda()
    ann = torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8)).cuda()
    ref = None
    for side in (False, True):
        net, crit = _build('retina', side)
        for rep in range(1 if not side else 5):
            outs = net.engine.forward(x, training=True, keep=True)
            fused = crit.fused(list(outs), ann, want_grad=True)
            net.engine.backward(*fused[2:])
            torch.cuda.synchronize()
            g = net.arena.grads.clone()
            if ref is None:
                ref = g
            assert torch.equal(g, ref), (side, rep)
            # BN running statistics move with every training forward: reset them so every repetition starts equal
            net.load_state_dict(omodel.init_retinanet_state(0))


def test_classifier_and_mae_side_streams_change_no_bit():
    from sacv_amd.classification import backbones, losses as closses
    from sacv_amd.train import ClassificationTrainer
    res = {}
    for side in (False, True):
        net = backbones.__dict__['resnet50'](num_classes=40)
        net.load_state_dict(omodel.init_resnet50_cls_state(0, 40), strict=True)
        net = net.cuda().train()
        net.engine.side_stream = side
        tr = ClassificationTrainer(net, closses.__dict__['CELoss'](), lr=1e-3, use_amp=True)
        x = torch.tensor(synth.images(11, 8, 96, 96)).cuda()
        lab = (torch.arange(8) % 40).cuda()
        for _ in range(2):
            tr.step(x, lab)
        torch.cuda.synchronize()
        res[side] = (net.arena.grads.clone(), net.arena.params.clone())
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])


@pytest.mark.parametrize('kind', ['retina', 'fcos'])
def test_class_count_not_a_multiple_of_four(kind):
    """Objects365's 365 classes (3285 / 365 class channels): the engine computes with zero-padded shadow weights and
    strips the outputs.  The trainer's fast path must produce the gradients of the class-level (autograd) path bit for
    bit, the decoder must read 365-wide class rows, and the padded channels must never reach the caller."""
    from sacv_amd.detection import models, losses, decode
    from sacv_amd.train import DetectionTrainer
    from oracle import decode as odecode
    fcos = kind == 'fcos'
    nc, H, W = 365, 96, 128
    sd = omodel.init_retinanet_state(0, num_classes=nc, fcos=fcos)
    name = 'resnet50_fcos' if fcos else 'resnet50_retinanet'
    x = torch.tensor(synth.images(7, 2, H, W)).cuda()
    ann = synth.annotations(9, 2, H, W, boxes_per_image=5)
    ann[..., 4] = (ann[..., 4] * 4 + 1) % nc * (ann[..., 4] >= 0) - (ann[..., 4] < 0)  # spread labels over 365 classes
    ann = torch.tensor(ann).cuda()
    crit = losses.__dict__['FCOSLoss' if fcos else 'RetinaLoss']()
    # class-level path
    net = models.__dict__[name](num_classes=nc)
    net.load_state_dict(sd)
    net = net.cuda().train()
    outs = net(x)
    assert tuple(outs[0][0].shape[3:]) == ((nc,) if fcos else (9, nc)) and outs[0][0].is_contiguous()
    ld = crit(outs, ann)
    sum(ld.values()).backward()
    ref = net.arena.grads.clone()
    w = 'clsregcnt_head.cls_out.weight' if fcos else 'cls_head.cls_out.weight'
    assert tuple(dict(net.named_parameters())[w].shape) == ((nc, 256, 3, 3) if fcos else (9 * nc, 256, 3, 3))
    assert float(dict(net.named_parameters())[w].grad.abs().sum()) > 0
    # trainer fast path from the same state (lr 0: parameters stay, gradients comparable)
    net2 = models.__dict__[name](num_classes=nc)
    net2.load_state_dict(sd)
    net2 = net2.cuda().train()
    tr = DetectionTrainer(net2, crit, lr=0.0, weight_decay=0.0)
    out = tr.step(x, ann)
    torch.cuda.synchronize()
    for k in ld:
        assert float(out[k]) == float(ld[k]), k
    assert torch.equal(net2.arena.grads, ref)
    # decoder on 365-wide class rows against the oracle
    net.eval()
    with torch.no_grad():
        preds = net(x)
    if fcos:
        dec = decode.__dict__['FCOSDecoder'](min_score_threshold=0.005)
        want = odecode.fcos_decode(*[[t.cpu().numpy() for t in grp] for grp in preds], min_score_threshold=0.005)
    else:
        dec = decode.__dict__['RetinaDecoder'](min_score_threshold=0.005)
        want = odecode.retina_decode(*[[t.cpu().numpy() for t in grp] for grp in preds], min_score_threshold=0.005)
    got = dec(preds)
    import numpy as np
    assert (want[0] > 0).sum() > 10
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_, w_)


def test_training_step_with_the_fused_stem():
    """engine.fused_stem = True (the 7x7 stem reads the image; its weight gradient rebuilds the im2col matrix on the side
    stream) against the default training path (im2col + GEMM in the forward pass, matrix kept): same loss to 1e-5, and
    every parameter's gradient norm within 5e-2 (two fp32 summation orders of the stem; the elementwise spread of deep
    graphs at this size is documented in test_gpu_pinned.py) -- the stem weight's own gradient included."""
    from sacv_amd.detection import models, losses
    H, W = gi.MID
    images = torch.tensor(synth.images(41, 2, H, W)).cuda()
    annots = torch.tensor(synth.annotations(42, 2, H, W, boxes_per_image=8)).cuda()
    sd = omodel.init_retinanet_state(0)
    res = []
    for fused in (False, True):
        net = models.__dict__['resnet50_retinanet']().cuda().train()
        net.load_state_dict(sd)
        net._engine.fused_stem = fused
        ld = losses.__dict__['RetinaLoss']()(net(images), annots)
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        res.append(({k: float(v) for k, v in ld.items()},
                    {k: float(p.grad.double().norm()) for k, p in net.named_parameters()}))
    (l0, g0), (l1, g1) = res
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-5 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    rel = {k: abs(g0[k] - g1[k]) / (g0[k] + 1e-12) for k in g0}
    assert max(rel.values()) <= 5e-2, sorted(rel.items(), key=lambda kv: -kv[1])[:3]
    assert rel['backbone.conv1.layer.0.weight'] <= 5e-2


@pytest.mark.parametrize('fn', ['sacv_stream_wait_stream', 'sacv_stream_wait_stream_ticket'])
def test_stream_wait_stream_orders_two_streams(fn):
    """sacv_stream_wait_stream (events without a system-scope fence) / sacv_stream_wait_stream_ticket (a ticket in device
    memory, polled by a one-lane kernel): a consumer on stream B behind a long producer on stream A must see the
    producer's result -- every iteration (the LAST element of the buffer, written by the producer's last workgroups), in
    both directions, and past the wrap of the event ring (300 > 256).  No ticket wait may have timed out."""
    from sacv_amd._lib import lib as lib_, check

    class lib:  # the function under test under one name
        sacv_stream_wait_stream = staticmethod(getattr(lib_, fn))
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    n, iters = 1 << 24, 300
    x = torch.zeros(n, device='cuda')
    snap = torch.zeros(iters, device='cuda')
    torch.cuda.synchronize()
    for it in range(iters):
        prod, cons = (a, b) if it % 2 == 0 else (b, a)
        with torch.cuda.stream(prod):
            for _ in range(4):          # a producer chain long enough for an unordered consumer to overtake it
                x.add_(0.25)
        check(lib.sacv_stream_wait_stream(cons.cuda_stream, prod.cuda_stream), 'stream_wait_stream')
        with torch.cuda.stream(cons):
            snap[it:it + 1].copy_(x[-1:])
        check(lib.sacv_stream_wait_stream(prod.cuda_stream, cons.cuda_stream), 'stream_wait_stream')  # cons reads x
    torch.cuda.synchronize()
    want = torch.arange(1, iters + 1, dtype=torch.float32, device='cuda')
    assert torch.equal(snap, want), (snap - want).abs().max()
    assert lib_.sacv_stream_ticket_timeouts() == 0
