"""GPU parity of the FCOS rows (SURVEY.md 8 a-3', a-4', a-9, a-10, FCOSDecoder) against the CPU oracle and the
reference-generated fixtures: bit-exact point targets' labels, losses 1e-4, GroupNorm vs torch, decoder
outputs bit-exact, end-to-end forward/loss/gradients."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import anchors as oanchors, assign as oassign, decode as odecode, losses as olosses, model as omodel, synth
from tests import golden_inputs as gi


def rnd(seed, *shape, std=1.0):
    return torch.from_numpy(synth.normal(seed, shape, std))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
def test_fcos_assign(tag, golden_dir):
    from sacv_amd import ops
    H, W = gi.SHAPES[tag]
    fs = synth.feature_sizes(H, W)
    pos = oanchors.fcos_positions(fs)
    meta = json.load(open(os.path.join(golden_dir, 'assign.json')))
    off = [0]
    for w, h in fs:
        off.append(off[-1] + w * h)
    pts = ops.fcos_points(fs, oanchors.DEFAULT_STRIDES, 'cuda')
    cases = gi.annots_cases(H, W)
    cases['none'] = synth.annotations(1, 2, H, W, boxes_per_image=0)
    for cname, ann in cases.items():
        tg, counts = ops.fcos_assign(pts, off, oanchors.DEFAULT_STRIDES, oassign.FCOS_MI, torch.tensor(ann).cuda())
        tg = tg.cpu().numpy()
        ref = oassign.fcos_assign(pos, fs, ann)
        assert np.array_equal(tg[..., 4], ref[..., 4]), (tag, cname)            # class labels: bit-exact
        assert np.array_equal(tg[..., 0:4], ref[..., 0:4])                        # ltrb: exact fp32 differences
        assert np.array_equal(tg[..., 6:8], ref[..., 6:8])
        np.testing.assert_allclose(tg[..., 5], ref[..., 5], rtol=1e-6, atol=1e-7)  # centerness (sqrt/div)
        assert int(counts[0]) == int((ref[..., 4] > 0).sum())
        key = f'fcos_{tag}_{cname}'
        if key in meta:
            assert hashlib.sha256(tg[..., 4].astype(np.int8).tobytes()).hexdigest() == meta[key]['label_sha256']


def _oracle_loss(crit, preds_np, ann):
    preds = [[torch.tensor(p, requires_grad=True) for p in lv] for lv in preds_np]
    ld = crit(preds, torch.tensor(ann))
    total = sum(ld.values())
    grads = None
    if total.requires_grad:
        total.backward()
        grads = [[(p.grad if p.grad is not None else torch.zeros_like(p)) for p in lv] for lv in preds]
    return {k: float(v) for k, v in ld.items()}, grads


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
@pytest.mark.parametrize('iou', ['GIoU', 'IoU', 'DIoU', 'CIoU', 'EIoU'])
def test_fcos_loss_and_grads(tag, iou):
    from sacv_amd.detection import losses
    H, W = gi.SHAPES[tag]
    cases = gi.annots_cases(H, W)
    cases['none'] = synth.annotations(1, 2, H, W, boxes_per_image=0)
    if tag == 'S1':
        cases = {k: cases[k] for k in ('g8', 'one_empty')}
    cls, reg, ctr = gi.synth_preds_fcos(13, 2, H, W)
    crit = losses.__dict__['FCOSLoss'](box_loss_iou_type=iou)
    ocrit = olosses.FCOSLossOracle(box_loss_iou_type=iou)
    for cname, ann in cases.items():
        ref, rgrads = _oracle_loss(ocrit, [cls, reg, ctr], ann)
        gp = [[torch.tensor(x).cuda().requires_grad_(True) for x in grp] for grp in (cls, reg, ctr)]
        ld = crit(gp, torch.tensor(ann).cuda())
        assert set(ld) == {'cls_loss', 'reg_loss', 'center_ness_loss'}
        for k in ld:
            assert abs(float(ld[k]) - ref[k]) <= 1e-4 * max(1.0, abs(ref[k])), (tag, cname, k)
        total = sum(ld.values())
        if rgrads is None:
            assert not total.requires_grad
            continue
        total.backward()
        for grp, rgrp in zip(gp, rgrads):
            for got, want in zip(grp, rgrp):
                scale = float(want.abs().max()) + 1e-12
                assert float((got.grad.cpu() - want).abs().max()) <= 5e-5 * scale + 1e-9, (tag, cname)


@pytest.mark.parametrize('shape', [(2, 256, 13, 21), (2, 256, 7, 11), (1, 256, 50, 84), (2, 256, 1, 2)])
def test_groupnorm_fwd_bwd(shape):
    from sacv_amd import ops
    n, c, h, w = shape
    x = rnd(1, *shape) * 1.5 + 0.3
    gamma, beta = 1.0 + 0.2 * rnd(2, c), 0.2 * rnd(3, c)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.relu(F.group_norm(xr, 32, gr, br, 1e-5))
    dz = rnd(4, *shape)
    z.backward(dz)
    zd, st = ops.groupnorm_fwd(nhwc(x), gamma.cuda(), beta.cuda(), 32, 1e-5, True)
    assert float((nchw(zd) - z.detach()).abs().max()) <= 2e-5 * max(1.0, float(z.abs().max()))
    dg, db = torch.zeros(c).cuda(), torch.zeros(c).cuda()
    dx = ops.groupnorm_bwd(nhwc(x), nhwc(dz), zd, st, gamma.cuda(), dg, db, 32)
    assert float((nchw(dx) - xr.grad).abs().max()) <= 5e-5 * max(1e-3, float(xr.grad.abs().max()))
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), rtol=2e-4, atol=2e-4)
    # accumulate over "levels"
    ops.groupnorm_bwd(nhwc(x), nhwc(dz), zd, st, gamma.cuda(), dg, db, 32, accumulate=True)
    np.testing.assert_allclose(dg.cpu().numpy(), 2 * gr.grad.numpy(), rtol=2e-4, atol=4e-4)


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
@pytest.mark.parametrize('nms_type', ['python_nms', 'diou_python_nms', 'torch_nms'])
def test_fcos_decoder(tag, nms_type, golden_dir):
    from sacv_amd.detection import decode
    H, W = gi.SHAPES[tag]
    cls, reg, ctr = gi.synth_preds_decode(23, 2, H, W, fcos=True)
    dec = decode.__dict__['FCOSDecoder'](nms_type=nms_type, nms_threshold=0.6)
    s, c, b = dec([[torch.tensor(x).cuda() for x in grp] for grp in (cls, reg, ctr)])
    rs, rc, rb = odecode.fcos_decode(cls, reg, ctr, nms_type=nms_type, nms_threshold=0.6)
    # sqrt(cls*ctr) is rounded identically on both sides only to 1 ulp: compare scores to 1e-6, and the
    # selection (classes, boxes) exactly
    np.testing.assert_allclose(s, rs, rtol=1e-6, atol=1e-7)
    assert np.array_equal(c, rc) and np.array_equal(b, rb)
    G = np.load(os.path.join(golden_dir, 'decode.npz'))
    key = f'fcos_{tag}_{nms_type}'
    if key + '_scores' in G:
        assert np.array_equal(c, G[key + '_classes']) and np.array_equal(b, G[key + '_boxes'])


def oracle_step(sd, images, annots, dtype=torch.float32):
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    state.update(params)
    outs = omodel.fcos_forward(state, images.to(dtype), training=True)
    ld = olosses.FCOSLossOracle()(outs, annots.to(dtype))
    sum(ld.values()).backward()
    return outs, ld, params


@pytest.mark.parametrize('tag', ['TINY', 'MID'])
def test_fcos_end_to_end(tag):
    from sacv_amd.detection import models, losses
    H, W = gi.SHAPES[tag]
    sd = omodel.init_retinanet_state(0, fcos=True)
    sd['scales'] = torch.tensor([1.0, 0.9, 1.1, 0.8, 1.2])
    images = torch.tensor(synth.images(5, 2, H, W))
    annots = torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8))
    outs_r, ld_r, params_r = oracle_step(sd, images, annots)
    _, _, params_64 = oracle_step(sd, images, annots, torch.float64)
    net = models.__dict__['resnet50_fcos'](num_classes=80)
    net.load_state_dict(sd)
    net = net.cuda().train()
    outs = net(images.cuda())
    assert len(outs) == 3 and all(len(g) == 5 for g in outs)
    for grp, grp_r in zip(outs, outs_r):
        for o, r in zip(grp, grp_r):
            assert tuple(o.shape) == tuple(r.shape)
            # GroupNorm over as few as 8 values (1x1 level, 8 channels per group) amplifies fp32 rounding: 2e-4
            assert float((o.detach().cpu() - r.detach()).abs().max()) <= 2e-4 * max(1.0, float(r.abs().max()))
    ld = losses.__dict__['FCOSLoss']()(outs, annots.cuda())
    for k in ('cls_loss', 'reg_loss', 'center_ness_loss'):
        assert abs(float(ld[k]) - float(ld_r[k])) <= 1e-4 * max(1.0, abs(float(ld_r[k]))), k
    sum(ld.values()).backward()
    num_g = num_c = den = 0.0
    for name, p in net.named_parameters():
        exact = params_64[name].grad
        num_g += float((p.grad.double().cpu() - exact).norm())**2
        num_c += float((params_r[name].grad.double() - exact).norm())**2
        den += float(exact.norm())**2
    e_gpu, e_cpu = (num_g / den)**0.5, (num_c / den)**0.5
    print(f'FCOS {tag}: gradient rel-L2 error vs fp64: gpu {e_gpu:.2e}, cpu fp32 {e_cpu:.2e}')
    assert e_gpu <= 2.0 * e_cpu + 1e-3
    # the learnable per-level scales (tiny, well-conditioned): direct comparison
    np.testing.assert_allclose(dict(net.named_parameters())['scales'].grad.cpu().numpy(),
                               params_r['scales'].grad.numpy(), rtol=2e-2, atol=1e-4)


def test_fcos_full_size_against_reference_fixture(golden_dir):
    from sacv_amd.detection import models, losses
    meta = json.load(open(os.path.join(golden_dir, 'e2e_s1.json')))['fcos_S1']
    H, W = gi.S1
    net = models.__dict__['resnet50_fcos'](num_classes=80)
    net.load_state_dict(omodel.init_retinanet_state(0, fcos=True))
    net = net.cuda().train()
    outs = net(torch.tensor(synth.images(5, 2, H, W)).cuda())
    ld = losses.__dict__['FCOSLoss']()(outs, torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8)).cuda())
    for k, v in meta['loss'].items():
        assert abs(float(ld[k]) - v) <= 1e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
    sum(ld.values()).backward()
    rel = []
    for name, p in net.named_parameters():
        n = float(p.grad.double().norm())
        rel.append(abs(n - meta['grad_norm'][name]) / (meta['grad_norm'][name] + 1e-12))
    print('FCOS S1 grad-norm rel err: median %.2e max %.2e' % (float(np.median(rel)), max(rel)))
    assert float(np.median(rel)) <= 1e-2 and max(rel) <= 1e-1
