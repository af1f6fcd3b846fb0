"""Oracle (test infrastructure): anchor<->GT and point<->GT target assignment, NumPy fp32.

Restates RetinaLoss.get_batch_anchors_annotations / snap_annotations_to_txtytwth
(simpleAICV/detection/losses.py:322-409), IoUMethod 'IoU' (losses.py:54-70) and
FCOSLoss.get_batch_position_annotations (losses.py:612-833).  Every intermediate
is float32 with the reference's operation order, so labels / matched indices are
bit-exact and regression targets agree to the last ulp of log/div.
"""
import numpy as np

F32 = np.float32


def iou_matrix(anchors, gts):
    """[A, G] fp32 IoU, operation order of losses.py:54-70."""
    a = anchors[:, None, :].astype(F32)
    g = gts[None, :, :].astype(F32)
    tl = np.maximum(a[..., 0:2], g[..., 0:2])
    br = np.minimum(a[..., 2:4], g[..., 2:4])
    wh = np.maximum(br - tl, F32(0))
    inter = wh[..., 0] * wh[..., 1]
    awh = np.maximum(a[..., 2:4] - a[..., 0:2], F32(0))
    gwh = np.maximum(g[..., 2:4] - g[..., 0:2], F32(0))
    a_area = awh[..., 0] * awh[..., 1]
    g_area = gwh[..., 0] * gwh[..., 1]
    union = np.maximum(a_area + g_area - inter, F32(1e-4))
    return (inter / union).astype(F32)


def retina_assign_image(anchors, annots, box_loss_type='SmoothL1'):
    """One image.  anchors [A,4] fp32, annots [N,5] fp32 (cls<0 rows are padding).

    Returns (targets [A,5] fp32, matched_gt [A] int32 (-1 when the image has no GT),
    overlap [A] fp32).  losses.py:336-381.
    """
    A = anchors.shape[0]
    valid = annots[annots[:, 4] >= 0]
    if valid.shape[0] == 0:
        return (np.full((A, 5), -1, dtype=F32), np.full((A,), -1, dtype=np.int32),
                np.zeros((A,), dtype=F32))
    gts = valid[:, 0:4].astype(F32)
    gcls = valid[:, 4].astype(F32)
    ious = iou_matrix(anchors, gts)
    idx = np.argmax(ious, axis=1)  # first maximal index, like torch.max on CPU
    overlap = ious[np.arange(A), idx]
    label = np.full((A,), -1, dtype=F32)
    label[overlap < F32(0.4)] = 0
    pos = overlap >= F32(0.5)
    label[pos] = gcls[idx][pos] + F32(1)
    matched = gts[idx]
    if box_loss_type == 'SmoothL1':
        # losses.py:390-409
        awh = anchors[:, 2:4] - anchors[:, 0:2]
        actr = anchors[:, 0:2] + F32(0.5) * awh
        gwh = np.maximum(matched[:, 2:4] - matched[:, 0:2], F32(1e-4))
        gctr = matched[:, 0:2] + F32(0.5) * gwh
        box = np.concatenate([(gctr - actr) / awh, np.log(gwh / awh)], axis=1)
    else:
        box = matched
    targets = np.concatenate([box.astype(F32), label[:, None]], axis=1)
    return targets.astype(F32), idx.astype(np.int32), overlap.astype(F32)


def retina_assign(anchors, annotations, box_loss_type='SmoothL1'):
    """Batch.  anchors [A,4], annotations [B,N,5] -> targets [B,A,5] fp32."""
    return np.stack([
        retina_assign_image(anchors, annotations[b], box_loss_type)[0]
        for b in range(annotations.shape[0])
    ], axis=0)


FCOS_MI = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, 100000000]]


def fcos_assign_image(positions, point_mi, point_stride, annots,
                      center_sample_radius=1.5, use_center_sample=True):
    """One image.  positions [P,2], point_mi [P,2], point_stride [P] fp32.

    Returns [P,8] fp32: l,t,r,b, cls+1 (0 = background), centerness, px, py.
    losses.py:664-826 (the one-candidate branch :758-777 is the G==1 special
    case of the smallest-area pick :778-823).
    """
    P = positions.shape[0]
    out = np.zeros((P, 8), dtype=F32)
    out[:, 6:8] = positions
    valid = annots[annots[:, 4] >= 0]
    if valid.shape[0] == 0:
        return out
    gts = valid[:, 0:4].astype(F32)
    gcls = valid[:, 4].astype(F32)
    px = positions[:, 0:1].astype(F32)
    py = positions[:, 1:2].astype(F32)
    ltrb = np.stack([px - gts[None, :, 0], py - gts[None, :, 1],
                     gts[None, :, 2] - px, gts[None, :, 3] - py], axis=-1).astype(F32)
    inside = (ltrb.min(axis=-1) > 0).astype(F32)
    cand = ltrb * inside[..., None]
    if use_center_sample:
        cx = (gts[:, 2] + gts[:, 0]) / F32(2)
        cy = (gts[:, 3] + gts[:, 1]) / F32(2)
        dist = np.sqrt((px - cx[None, :])**2 + (py - cy[None, :])**2).astype(F32)
        judge = (point_stride.astype(F32) * F32(center_sample_radius))[:, None]
        cand = cand * (dist < judge).astype(F32)[..., None]
    cmax = cand.max(axis=-1)
    cand = cand * (cmax > point_mi[:, 0:1]).astype(F32)[..., None]
    cand = cand * (cmax < point_mi[:, 1:2]).astype(F32)[..., None]
    value = cand.sum(axis=-1)
    positive = value.sum(axis=-1) > 0
    if not positive.any():
        return out
    gwh = gts[:, 2:4] - gts[:, 0:2]
    area = (gwh[:, 0] * gwh[:, 1]).astype(F32)
    area = np.where(value == 0, F32(100000000), area[None, :]).astype(F32)
    pick = np.argmin(area, axis=1)
    rows = np.nonzero(positive)[0]
    sel = cand[rows, pick[rows]]
    out[rows, 0:4] = sel
    out[rows, 4] = gcls[pick[rows]] + F32(1)
    l, t, r, b = sel[:, 0], sel[:, 1], sel[:, 2], sel[:, 3]
    with np.errstate(divide='ignore', invalid='ignore'):
        out[rows, 5] = np.sqrt((np.minimum(l, r) / np.maximum(l, r)) *
                               (np.minimum(t, b) / np.maximum(t, b)))
    return out


def fcos_point_tables(feature_sizes, strides, mi=FCOS_MI):
    """Per-point (mi [P,2], stride [P]) fp32 tables, levels concatenated."""
    mis, sts = [], []
    for (fw, fh), m, s in zip(feature_sizes, mi, strides):
        n = fw * fh
        mis.append(np.tile(np.asarray(m, dtype=F32)[None, :], (n, 1)))
        sts.append(np.full((n,), s, dtype=F32))
    return np.concatenate(mis, 0), np.concatenate(sts, 0)


def fcos_assign(positions, feature_sizes, annotations, strides=(8, 16, 32, 64, 128),
                mi=FCOS_MI, center_sample_radius=1.5, use_center_sample=True):
    """Batch.  positions [P,2]; annotations [B,N,5] -> [B,P,8] fp32."""
    pmi, pst = fcos_point_tables(feature_sizes, strides, mi)
    return np.stack([
        fcos_assign_image(positions, pmi, pst, annotations[b],
                          center_sample_radius, use_center_sample)
        for b in range(annotations.shape[0])
    ], axis=0)
