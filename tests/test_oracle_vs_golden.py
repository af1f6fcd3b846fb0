"""Pins the CPU oracle (oracle/) against outputs of the REFERENCE itself.

The fixtures in tests/golden/ were produced by oracle/make_golden.py, which runs
the reference code from /root/reference on CPU.  These tests run anywhere (no GPU,
no reference) and establish that the restatement equals the reference: bit-exact
for anchors / labels / matched indices / detections, <=1e-6 for box targets,
1e-5 relative for fp32 losses and gradients.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import anchors as oanchors, assign as oassign, decode as odecode, losses as olosses, synth
from tests import golden_inputs as gi


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope='module')
def G(golden_dir):
    class _G:
        pass
    g = _G()
    for name in ('anchors', 'assign', 'losses', 'decode'):
        setattr(g, name, np.load(os.path.join(golden_dir, name + '.npz')))
        setattr(g, name + '_meta', json.load(open(os.path.join(golden_dir, name + '.json'))))
    return g


def test_feature_sizes_match_reference_model(golden_dir):
    e2e = json.load(open(os.path.join(golden_dir, 'e2e.json')))
    assert e2e['retina_TINY']['feature_sizes'] == synth.feature_sizes(*gi.TINY)
    assert e2e['retina_MID']['feature_sizes'] == synth.feature_sizes(*gi.MID)
    assert synth.feature_sizes(*gi.S1) == [[167, 100], [84, 50], [42, 25], [21, 13], [11, 7]]


def test_base_anchors_bit_exact(G):
    ref = G.anchors['base_anchors']
    for l, area in enumerate(oanchors.DEFAULT_AREAS):
        got = oanchors.base_anchors(area, oanchors.DEFAULT_SCALES, oanchors.DEFAULT_RATIOS)
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref[l].view(np.uint32))


@pytest.mark.parametrize('tag', ['S1', 'S2', 'TINY', 'MID'])
def test_anchors_and_points_sha(G, tag):
    m = G.anchors_meta[tag]
    fs = synth.feature_sizes(*gi.SHAPES[tag])
    assert fs == m['feature_sizes']
    a = oanchors.retina_anchors(fs)
    assert a.shape[0] == m['n_anchors'] and sha(a) == m['anchors_sha256']
    p = oanchors.fcos_positions(fs)
    assert p.shape[0] == m['n_points'] and sha(p) == m['points_sha256']


def test_anchor_samples(G):
    lv = oanchors.retina_anchors_per_level(synth.feature_sizes(*gi.S1))
    for l, a in enumerate(lv):
        f = a.reshape(-1, 4)
        assert np.array_equal(f[:64], G.anchors[f'S1_l{l}_first64'])
        assert np.array_equal(f[-64:], G.anchors[f'S1_l{l}_last64'])
    assert np.array_equal(oanchors.retina_anchors(synth.feature_sizes(*gi.TINY)), G.anchors['anchors_TINY'])


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
def test_retina_assign(G, tag):
    H, W = gi.SHAPES[tag]
    anc = oanchors.retina_anchors(synth.feature_sizes(H, W))
    for cname, ann in gi.annots_cases(H, W).items():
        for blt in ('SmoothL1', 'GIoU'):
            key = f'retina_{tag}_{cname}_{blt}'
            tg = oassign.retina_assign(anc, ann, blt)
            lab = tg[..., 4].astype(np.int8)
            m = G.assign_meta[key]
            assert sha(lab) == m['label_sha256'], key
            assert int((lab > 0).sum()) == m['npos']
            if tag == 'TINY':
                ref = G.assign[key]
                assert np.array_equal(tg[..., 4], ref[..., 4])
                np.testing.assert_allclose(tg[..., :4], ref[..., :4], rtol=0, atol=1e-6)
            else:
                ref = G.assign[key + '_postargets']
                np.testing.assert_allclose(tg[lab > 0][:, :4], ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
def test_fcos_assign(G, tag):
    H, W = gi.SHAPES[tag]
    fs = synth.feature_sizes(H, W)
    pos = oanchors.fcos_positions(fs)
    for cname, ann in gi.annots_cases(H, W).items():
        key = f'fcos_{tag}_{cname}'
        tg = oassign.fcos_assign(pos, fs, ann)
        lab = tg[..., 4].astype(np.int8)
        assert sha(lab) == G.assign_meta[key]['label_sha256'], key
        if tag == 'TINY':
            np.testing.assert_allclose(tg, G.assign[key], rtol=1e-6, atol=1e-6)
        else:
            np.testing.assert_allclose(tg[lab > 0], G.assign[key + '_postargets'], rtol=1e-6, atol=1e-6)


def _check_digest(got_list, meta, samples, rtol=2e-5):
    d = gi.digest(np.concatenate([np.asarray(g).reshape(-1) for g in got_list]))
    assert d['n'] == meta['n']
    np.testing.assert_allclose(d['sum'], meta['sum'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(d['abssum'], meta['abssum'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(d['samples'], samples, rtol=rtol, atol=1e-9)


def _run_oracle_loss(crit, preds_np, ann):
    preds = [[torch.tensor(p, requires_grad=True) for p in lv] for lv in preds_np]
    ld = crit(preds, torch.tensor(ann))
    total = sum(ld.values())
    grads = None
    if total.requires_grad:
        total.backward()
        grads = [[(p.grad if p.grad is not None else torch.zeros_like(p)).numpy() for p in lv] for lv in preds]
    return {k: float(v) for k, v in ld.items()}, grads


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
def test_losses_and_grads(G, tag):
    H, W = gi.SHAPES[tag]
    cases = gi.annots_cases(H, W)
    cases['none'] = synth.annotations(1, 2, H, W, boxes_per_image=0)
    todo = [k for k in G.losses_meta if k.split('_')[1] == tag]
    assert todo
    for key in todo:
        parts = key.split('_')
        kind = parts[0]
        if kind == 'retina':
            cname, blt = '_'.join(parts[2:-1]), parts[-1]
            crit = olosses.RetinaLossOracle(box_loss_type=blt)
            preds = list(gi.synth_preds_retina(11, 2, H, W))
            names = ('dcls', 'dreg')
        else:
            cname = '_'.join(parts[2:])
            crit = olosses.FCOSLossOracle()
            preds = list(gi.synth_preds_fcos(13, 2, H, W))
            names = ('dcls', 'dreg', 'dctr')
        rec, grads = _run_oracle_loss(crit, preds, cases[cname])
        m = G.losses_meta[key]
        for lk, lv in rec.items():
            np.testing.assert_allclose(lv, m[lk], rtol=1e-5, atol=1e-7, err_msg=key)
        assert (grads is not None) == m['has_grad'], key
        if grads is not None:
            for nm, gl in zip(names, grads):
                _check_digest(gl, m[nm], G.losses[f'{key}_{nm}_samples'])


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
def test_decoders(G, tag):
    H, W = gi.SHAPES[tag]
    for nms_type in ('python_nms', 'diou_python_nms'):
        key = f'retina_{tag}_{nms_type}'
        if key + '_scores' not in G.decode:
            continue
        cls, reg = gi.synth_preds_decode(21, 2, H, W)
        s, c, b = odecode.retina_decode(cls, reg, nms_type=nms_type, nms_threshold=0.5)
        assert np.array_equal(s, G.decode[key + '_scores'])
        assert np.array_equal(c, G.decode[key + '_classes'])
        assert np.array_equal(b, G.decode[key + '_boxes'])
        key = f'fcos_{tag}_{nms_type}'
        cls, reg, ctr = gi.synth_preds_decode(23, 2, H, W, fcos=True)
        s, c, b = odecode.fcos_decode(cls, reg, ctr, nms_type=nms_type, nms_threshold=0.6)
        assert np.array_equal(s, G.decode[key + '_scores'])
        assert np.array_equal(c, G.decode[key + '_classes'])
        assert np.array_equal(b, G.decode[key + '_boxes'])


def test_torchvision_nms_restatement_agrees_with_python_nms_on_regular_boxes():
    """nms_type='torch_nms' (torchvision.ops.nms; parity unpinned, torchvision absent): on boxes with positive area
    and no IoU exactly at the threshold it must keep what the reference's python_nms keeps; a zero-area duplicate
    (IoU 0/0) is kept by torchvision's rule."""
    rng = np.random.default_rng(0)
    xy = rng.uniform(0, 200, (300, 2)).astype(np.float32)
    wh = rng.uniform(5, 80, (300, 2)).astype(np.float32)
    boxes = np.trunc(np.concatenate([xy, xy + wh], axis=1)).astype(np.float32)
    assert np.array_equal(odecode.torchvision_nms(boxes, 0.5), odecode.greedy_nms(boxes, 0.5))
    deg = np.array([[10, 10, 10, 10], [10, 10, 10, 10], [0, 0, 30, 30]], dtype=np.float32)
    assert odecode.torchvision_nms(deg, 0.5).tolist() == [0, 1, 2]


def test_soft_nms_restatement_known_answers_and_limits():
    """Soft-NMS (Bodla et al. 2017; the north-star's extra nms mode, PARITY UNPINNED: the reference has none).  The
    restatement is pinned by what the paper's algorithm implies: a hand-computed case, the hard-NMS limit (prune at the
    score a linear decay by 1 - IoU leaves: picks of linear soft-NMS that never decayed == python_nms's keeps when
    overlapping boxes are pruned outright), non-increasing pick scores, and no decay below the threshold."""
    F32 = np.float32
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=F32)
    scores = np.array([0.9, 0.8, 0.7, 0.6], dtype=F32)
    # IoU(0,1) = 50/100 = 0.5; IoU(0,3) = 1; box 2 is disjoint
    picks, dec = odecode.soft_nms(scores, boxes, nms_threshold=0.3, method='linear', prune_score=0.001)
    # pick 0 (0.9): s1 = 0.8*(1-0.5) = 0.4, s3 = 0.6*0 = 0 -> pruned; then 2 (0.7, no overlap), then 1 (0.4)
    assert picks.tolist() == [0, 2, 1]
    assert dec.tolist() == [F32(0.9), F32(0.7), F32(0.8) * (F32(1) - F32(0.5))]
    picks, dec = odecode.soft_nms(scores, boxes, nms_threshold=0.3, method='gaussian', sigma=0.5, prune_score=0.001)
    assert picks.tolist() == [0, 2, 1, 3]
    np.testing.assert_allclose(dec, [0.9, 0.7, 0.8 * np.exp(-0.25 / 0.5), 0.6 * np.exp(-1.0 / 0.5) * np.exp(-0.25 / 0.5)],
                               rtol=1e-6)
    # below the threshold linear soft-NMS changes nothing
    picks, dec = odecode.soft_nms(scores[:3], boxes[:3], nms_threshold=0.6, method='linear')
    assert picks.tolist() == [0, 1, 2] and dec.tolist() == scores[:3].tolist()
    # random boxes: pick scores never increase, every pick is unique, and with a prune level above every decayed
    # score (decay kills) the picks are exactly hard NMS's keeps at the same threshold with >= semantics
    rng = np.random.RandomState(3)
    xy = rng.rand(300, 2).astype(F32) * 200
    wh = (rng.rand(300, 2).astype(F32) * 60 + 5)
    b = np.concatenate([xy, xy + wh], axis=1).astype(np.int32).astype(F32)
    s = np.sort(rng.rand(300).astype(F32) * F32(0.9) + F32(0.05))[::-1].copy()
    for method in ('linear', 'gaussian'):
        picks, dec = odecode.soft_nms(s, b, nms_threshold=0.5, method=method)
        assert len(set(picks.tolist())) == len(picks) and np.all(np.diff(dec) <= 0)
        assert np.all(dec <= s[picks]) and np.all(dec > F32(0.001))
    picks, dec = odecode.soft_nms(s, b, nms_threshold=0.5, method='linear', prune_score=10.0)
    assert picks.size == 0  # nothing is live above a prune level of 10
    keep = odecode.greedy_nms(b, 0.5)
    picks, dec = odecode.soft_nms(s, b, nms_threshold=0.5, method='linear')
    assert picks[0] == keep[0] == 0 and dec[0] == s[0]
    # a box that overlaps NO other box by >= 0.5 is never decayed by the linear rule, and hard NMS keeps it too
    wh_ = b[:, 2:4] - b[:, 0:2]
    area = wh_[:, 0] * wh_[:, 1]
    tl = np.maximum(b[:, None, 0:2], b[None, :, 0:2])
    br = np.minimum(b[:, None, 2:4], b[None, :, 2:4])
    d_ = np.maximum(br - tl, 0)
    inter = d_[..., 0] * d_[..., 1]
    iou = inter / np.maximum(area[:, None] + area[None, :] - inter, F32(1e-4))
    np.fill_diagonal(iou, 0)
    lonely = np.where(iou.max(axis=1) < 0.5)[0]
    assert lonely.size > 10
    score_of = dict(zip(picks.tolist(), dec.tolist()))
    assert all(score_of[int(i)] == s[i] for i in lonely) and set(lonely.tolist()) <= set(keep.tolist())
    # select_detections wraps it like the other modes (filter / sort / top-n / pad)
    sc = np.stack([s, s[::-1]])
    out = odecode.select_detections(sc, np.zeros((2, 300), dtype=np.int64), np.stack([b, b]).astype(np.int32),
                                    nms_type='soft_nms', max_object_num=20)
    assert out[0].shape == (2, 20) and np.all(np.diff(out[0], axis=1) <= 0) and out[2].shape == (2, 20, 4)
