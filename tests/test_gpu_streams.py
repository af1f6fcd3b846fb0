"""The side HIP stream of the engines (weight gradients, bias-gradient column sums, the shortcut branch of the forward
pass, the data-gradient panel refresh) must not change a single bit: the same kernels run with the same operands, only
on another queue.  Any missing dependency (a weight gradient reading a dy that the launch stream is still writing, a
scratch buffer shared by two streams) shows up here as a difference between the two modes or between repetitions."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth


def _build(kind, side):
    from sacv_amd.detection import models, losses
    old = os.environ.get('SACV_WGRAD_STREAM')
    os.environ['SACV_WGRAD_STREAM'] = '1' if side else '0'
    try:
        fcos = kind == 'fcos'
        net = models.__dict__['resnet50_fcos' if fcos else 'resnet50_retinanet'](num_classes=80)
        net.load_state_dict(omodel.init_retinanet_state(0, fcos=fcos))
        net = net.cuda().train()
        assert (net.engine._side is not None) == side
    finally:
        if old is None:
            os.environ.pop('SACV_WGRAD_STREAM', None)
        else:
            os.environ['SACV_WGRAD_STREAM'] = old
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
        old = os.environ.get('SACV_WGRAD_STREAM')
        os.environ['SACV_WGRAD_STREAM'] = '1' if side else '0'
        try:
            net = backbones.__dict__['resnet50'](num_classes=40)
            net.load_state_dict(omodel.init_resnet50_cls_state(0, 40), strict=True)
            net = net.cuda().train()
        finally:
            if old is None:
                os.environ.pop('SACV_WGRAD_STREAM', None)
            else:
                os.environ['SACV_WGRAD_STREAM'] = old
        tr = ClassificationTrainer(net, closses.__dict__['CELoss'](), lr=1e-3, use_amp=True)
        x = torch.tensor(synth.images(11, 8, 96, 96)).cuda()
        lab = (torch.arange(8) % 40).cuda()
        for _ in range(2):
            tr.step(x, lab)
        torch.cuda.synchronize()
        res[side] = (net.arena.grads.clone(), net.arena.params.clone())
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
