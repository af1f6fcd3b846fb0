"""Size-independent properties at the largest configuration of BASELINE.json (1333x1333, SURVEY.md 8c 'S2'), where
the CPU oracle would take minutes: batch independence of the eval-mode forward pass, decoder invariants
(descending scores, padded tail, integer-truncated boxes inside the image, NMS idempotence) and the zero-gradient
identity of a zero loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth
from tests import golden_inputs as gi


def test_s2_forward_batch_independence_and_decoder_invariants():
    from sacv_amd.detection import models, decode
    H, W = gi.S2
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    sd = omodel.init_retinanet_state(0)
    # a classification bias that lets a few hundred anchors pass the 0.05 score threshold
    sd['cls_head.cls_out.bias'] = sd['cls_head.cls_out.bias'] + 2.0
    net.load_state_dict(sd)
    net = net.cuda().eval()
    x = torch.tensor(synth.images(9, 2, H, W)).cuda()
    with torch.no_grad():
        both = net(x)
        one = net(x[1:2].contiguous())
    assert [tuple(o.shape[1:3]) for o in both[0]] == [tuple(reversed(s)) for s in synth.feature_sizes(H, W)]
    # eval-mode BatchNorm uses running statistics: an image's outputs do not depend on its batch mates (up to the
    # summation order: the K split of a convolution is planned from the problem size)
    for grp_b, grp_o in zip(both, one):
        for b, o in zip(grp_b, grp_o):
            torch.testing.assert_close(b[1:2], o, rtol=2e-5, atol=2e-6)
    dec = decode.__dict__['RetinaDecoder'](nms_type='python_nms', nms_threshold=0.5)
    scores, classes, boxes = dec(both)
    assert scores.shape == (2, 100) and classes.shape == (2, 100) and boxes.shape == (2, 100, 4)
    for b in range(2):
        n = int((scores[b] > -1).sum())
        assert n > 0
        assert np.all(np.diff(scores[b, :n]) <= 0) and np.all(scores[b, :n] > 0.05)
        assert np.all(scores[b, n:] == -1) and np.all(classes[b, n:] == -1) and np.all(boxes[b, n:] == -1)
        bx = boxes[b, :n]
        assert np.all(bx == np.trunc(bx))  # decode.py:135 casts to int
        assert np.all(bx[:, 2] >= bx[:, 0]) and np.all(bx[:, 3] >= bx[:, 1])
        # greedy NMS output is a fixed point of NMS: no kept pair of one class overlaps more than the threshold
        for c in np.unique(classes[b, :n]):
            k = bx[classes[b, :n] == c]
            if len(k) < 2:
                continue
            area = (k[:, 2] - k[:, 0]) * (k[:, 3] - k[:, 1])
            for i in range(len(k)):
                iw = np.clip(np.minimum(k[i, 2], k[:, 2]) - np.maximum(k[i, 0], k[:, 0]), 0, None)
                ih = np.clip(np.minimum(k[i, 3], k[:, 3]) - np.maximum(k[i, 1], k[:, 1]), 0, None)
                iou = iw * ih / np.maximum(area[i] + area - iw * ih, 1e-4)
                iou[i] = 0
                assert iou.max() <= 0.5 + 1e-6


def test_s2_training_step_runs_and_is_linear_in_the_loss_scale():
    """One full 1333x1333 step: finite losses; the backward pass is linear in the upstream gradient
    (grad_scale 0.5 gives exactly half of every gradient: scaling by a power of two is exact in fp32)."""
    from sacv_amd.detection import models, losses
    H, W = gi.S2
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    net.load_state_dict(omodel.init_retinanet_state(0))
    net = net.cuda().train()
    crit = losses.__dict__['RetinaLoss']()
    x = torch.tensor(synth.images(9, 2, H, W)).cuda()
    ann = torch.tensor(synth.annotations(4, 2, H, W, boxes_per_image=8)).cuda()
    eng = net.engine
    outs = eng.forward(x, training=True, keep=True)
    fused = crit.fused(list(outs), ann, want_grad=True)
    assert torch.isfinite(fused[0]).all() and float(fused[0].sum()) > 0
    eng.backward(*fused[2:], grad_scale=1.0)
    g1 = net.arena.grads.clone()
    outs = eng.forward(x, training=True, keep=True)
    fused = crit.fused(list(outs), ann, want_grad=True)
    eng.backward(*fused[2:], grad_scale=0.5)
    g2 = net.arena.grads
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert torch.equal(g2 * 2.0, g1)
