"""Oracle (test infrastructure): RetinaDecoder / FCOSDecoder postprocess in NumPy.

Restates simpleAICV/detection/decode.py: greedy class-agnostic NMS :45-102
(python_nms and diou_python_nms), score filter / sort / top-n / pad :121-172,
RetinaDecoder :201-271 (argmax class, box decode, int32 truncation),
FCOSDecoder :293-364 (sqrt(cls*ctr) score, exp(ltrb) boxes).  `torch_nms`
(torchvision) is absent in this image: parity unpinned, not restated.
"""
import numpy as np

from . import anchors as _anchors

F32 = np.float32


def torchvision_nms(boxes, nms_threshold=0.5):
    """torchvision.ops.nms as documented (the reference's nms_type='torch_nms', decode.py:39-43): greedy over boxes
    sorted by descending score; a box is dropped when its IoU with a kept box is > threshold; IoU = inter /
    (area_a + area_b - inter) with unclamped areas (0/0 = nan never suppresses).  PARITY UNPINNED: torchvision is
    not installed here, so this restates the published algorithm instead of being checked against the library."""
    boxes = boxes.astype(F32)
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    n = boxes.shape[0]
    dead = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        o = boxes[i + 1:]
        w = np.maximum(np.minimum(boxes[i, 2], o[:, 2]) - np.maximum(boxes[i, 0], o[:, 0]), F32(0))
        h = np.maximum(np.minimum(boxes[i, 3], o[:, 3]) - np.maximum(boxes[i, 1], o[:, 1]), F32(0))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = inter / (areas[i] + areas[i + 1:] - inter)
        dead[i + 1:] |= iou > F32(nms_threshold)
    return np.asarray(keep, dtype=np.int64)


def greedy_nms(boxes, nms_threshold=0.5, diou=False):
    """Indices kept, boxes [N,4] fp32 already sorted by descending score."""
    boxes = boxes.astype(F32)
    wh = boxes[:, 2:4] - boxes[:, 0:2]
    areas = np.maximum(wh[:, 0] * wh[:, 1], F32(0))
    alive = np.arange(boxes.shape[0])
    keep = []
    while alive.size:
        k = alive[0]
        keep.append(k)
        alive = alive[1:]
        if alive.size == 0:
            break
        o = boxes[alive]
        tl = np.maximum(boxes[k, 0:2], o[:, 0:2])
        br = np.minimum(boxes[k, 2:4], o[:, 2:4])
        s = np.maximum(br - tl, F32(0))
        inter = s[:, 0] * s[:, 1]
        union = np.maximum(areas[k] + areas[alive] - inter, F32(1e-4))
        iou = inter / union
        if diou:
            etl = np.minimum(boxes[k, 0:2], o[:, 0:2])
            ebr = np.maximum(boxes[k, 2:4], o[:, 2:4])
            es = np.maximum(ebr - etl, F32(0))
            c2 = np.maximum((es**2).sum(axis=1), F32(1e-4))
            kc = (boxes[k, 2:4] + boxes[k, 0:2]) / F32(2)
            oc = (o[:, 2:4] + o[:, 0:2]) / F32(2)
            p2 = ((kc - oc)**2).sum(axis=1)
            iou = iou - p2 / c2
        alive = alive[iou < nms_threshold]
    return np.asarray(keep, dtype=np.int64)


def soft_nms(scores, boxes, nms_threshold=0.5, method='linear', sigma=0.5, prune_score=0.001, max_picks=None):
    """Soft-NMS of Bodla, Singh, Chellappa, Davis, "Soft-NMS -- Improving Object Detection With One Line of Code"
    (ICCV 2017), algorithm of its figure 2 with the rescoring functions of eq. 2 (linear) / eq. 3 (gaussian) -- the
    extra `nms_type` BASELINE.json's north_star names.  PARITY UNPINNED: /root/reference has no soft-NMS
    (DetNMSMethod, decode.py:26-104, stops at torch_nms / python_nms / diou_python_nms) and no third-party
    implementation is importable here, so this restates the published algorithm; the IoU is the reference family's
    (python_nms, decode.py:45-72: area = max(w*h, 0), union clamped at 1e-4), fp32 throughout.

    scores [N] / boxes [N,4] fp32 sorted by descending score.  Returns (picked indices, their decayed scores) in pick
    order = descending decayed score; ties go to the earlier index; a box leaves once its score is <= prune_score."""
    s = scores.astype(F32).copy()
    boxes = boxes.astype(F32)
    wh = boxes[:, 2:4] - boxes[:, 0:2]
    areas = np.maximum(wh[:, 0] * wh[:, 1], F32(0))
    live = s > F32(prune_score)
    picks, picked_scores = [], []
    limit = boxes.shape[0] if max_picks is None else max_picks
    while live.any() and len(picks) < limit:
        cand = np.where(live)[0]
        m = cand[np.argmax(s[cand])]  # first maximum = lowest index among equal scores
        picks.append(int(m))
        picked_scores.append(s[m])
        live[m] = False
        o = np.where(live)[0]
        if o.size == 0:
            break
        tl = np.maximum(boxes[m, 0:2], boxes[o, 0:2])
        br = np.minimum(boxes[m, 2:4], boxes[o, 2:4])
        d = np.maximum(br - tl, F32(0))
        inter = d[:, 0] * d[:, 1]
        union = np.maximum((areas[m] + areas[o]) - inter, F32(1e-4))
        iou = inter / union
        if method == 'linear':
            w = np.where(iou >= F32(nms_threshold), F32(1) - iou, F32(1))
        else:
            w = np.exp(-(iou * iou) / F32(sigma)).astype(F32)
        s[o] = s[o] * w
        live[o] = s[o] > F32(prune_score)
    return np.asarray(picks, dtype=np.int64), np.asarray(picked_scores, dtype=F32)


_SOFT_METHODS = {'soft_nms': 'linear', 'soft_nms_linear': 'linear', 'soft_nms_gaussian': 'gaussian'}


def select_detections(scores, classes, boxes, max_object_num=100, min_score_threshold=0.05,
                      topn=1000, nms_type='python_nms', nms_threshold=0.5, soft_nms_sigma=0.5,
                      soft_nms_prune_score=0.001):
    """decode.py:121-172.  scores [B,N] f32, classes [B,N] int, boxes [B,N,4] int32.  The soft_nms* types are the
    extension described at soft_nms() (same filter / sort / top-n / pad around it; output scores = decayed scores)."""
    assert nms_type in ('python_nms', 'diou_python_nms', 'torch_nms') or nms_type in _SOFT_METHODS
    B = scores.shape[0]
    out_s = np.full((B, max_object_num), -1, dtype=F32)
    out_c = np.full((B, max_object_num), -1, dtype=F32)
    out_b = np.zeros((B, max_object_num, 4), dtype=F32)
    for i in range(B):
        m = scores[i] > min_score_threshold
        s = scores[i][m].astype(F32)
        c = classes[i][m].astype(F32)
        b = boxes[i][m].astype(F32)
        if s.shape[0] == 0:
            continue
        # the reference calls np.argsort(-scores) (decode.py:141, quicksort: the order of EQUAL scores is unspecified);
        # 'stable' picks one of the permitted orders -- equal scores by ascending anchor index, the device's rule
        order = np.argsort(-s, kind='stable')[:topn]
        s, c, b = s[order], c[order], b[order]
        if nms_type in _SOFT_METHODS:
            keep, decayed = soft_nms(s, b, nms_threshold, _SOFT_METHODS[nms_type], soft_nms_sigma,
                                     soft_nms_prune_score, max_picks=max_object_num)
            n = keep.shape[0]
            out_s[i, :n], out_c[i, :n], out_b[i, :n] = decayed, c[keep], b[keep]
            continue
        keep = (torchvision_nms(b, nms_threshold) if nms_type == 'torch_nms' else
                greedy_nms(b, nms_threshold, diou=(nms_type == 'diou_python_nms')))
        keep = keep[:max_object_num]
        n = keep.shape[0]
        out_s[i, :n], out_c[i, :n], out_b[i, :n] = s[keep], c[keep], b[keep]
    return [out_s, out_c, out_b]


def _flat(preds):
    return np.concatenate([np.asarray(p, dtype=F32).reshape(p.shape[0], -1, p.shape[-1])
                           for p in preds], axis=1)


def retina_decode_boxes(reg, anchors):
    """decode.py:251-271: fp32 decode then truncation toward zero to int32."""
    awh = anchors[..., 2:4] - anchors[..., 0:2]
    actr = anchors[..., 0:2] + F32(0.5) * awh
    wh = np.exp(reg[..., 2:4]) * awh
    ctr = reg[..., 0:2] * awh + actr
    boxes = np.concatenate([ctr - F32(0.5) * wh, ctr + F32(0.5) * wh], axis=-1)
    return boxes.astype(np.int32)


def retina_decode(cls_preds, reg_preds, anchor_cfg=None, **kw):
    """cls_preds: 5 arrays [B,H,W,A,C]; reg_preds: 5 arrays [B,H,W,A,4]."""
    fs = [[p.shape[2], p.shape[1]] for p in cls_preds]
    anc = _anchors.retina_anchors(fs, **(anchor_cfg or {}))
    cls = _flat(cls_preds)
    reg = _flat(reg_preds)
    classes = np.argmax(cls, axis=2)
    scores = np.take_along_axis(cls, classes[..., None], axis=2)[..., 0]
    boxes = retina_decode_boxes(reg, anc[None])
    return select_detections(scores, classes, boxes, **kw)


def fcos_decode(cls_preds, reg_preds, ctr_preds, strides=_anchors.DEFAULT_STRIDES, **kw):
    kw.setdefault('nms_threshold', 0.6)
    fs = [[p.shape[2], p.shape[1]] for p in cls_preds]
    pos = _anchors.fcos_positions(fs, strides)[None]
    cls, reg, ctr = _flat(cls_preds), _flat(reg_preds), _flat(ctr_preds)
    classes = np.argmax(cls, axis=2)
    scores = np.take_along_axis(cls, classes[..., None], axis=2)[..., 0]
    scores = np.sqrt(scores * ctr[..., 0])
    e = np.exp(reg)
    boxes = np.concatenate([pos - e[..., 0:2], pos + e[..., 2:4]], axis=2).astype(np.int32)
    return select_detections(scores, classes, boxes, **kw)


def collate(samples, resize=800, resize_type='retina_style', max_annots_num=100):
    """Normalize + DetectionCollater (simpleAICV/detection/common.py:223-286) in NumPy: samples hold uint8 HWC
    images; returns (images [B,R,R,3] fp32 = value/255 zero padded, annots [B,max,5] padded with -1)."""
    r = int(round(resize * 1333. / 800)) if resize_type == 'retina_style' else resize
    out = np.zeros((len(samples), r, r, 3), dtype=F32)
    ann = np.full((len(samples), max_annots_num, 5), -1, dtype=F32)
    for i, s in enumerate(samples):
        im = s['image'].astype(F32) / 255.
        out[i, :im.shape[0], :im.shape[1], :] = im
        if s['annots'].shape[0] > 0:
            ann[i, :s['annots'].shape[0]] = s['annots']
    return out, ann


def eval_postprocess(boxes, scales, sizes):
    """tools/scripts.py:742-758 on a copy of boxes [B,M,4] fp32."""
    b = boxes.astype(F32).copy()
    b /= np.asarray(scales, dtype=F32)[:, None, None]
    for i in range(b.shape[0]):
        b[i, :, 0] = np.maximum(b[i, :, 0], 0)
        b[i, :, 1] = np.maximum(b[i, :, 1], 0)
        b[i, :, 2] = np.minimum(b[i, :, 2], sizes[i][1])
        b[i, :, 3] = np.minimum(b[i, :, 3], sizes[i][0])
        b[i, :, 2:] -= b[i, :, :2]
    return b
