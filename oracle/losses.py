"""Oracle (test infrastructure): RetinaLoss / FCOSLoss restated on torch-CPU fp32.

Differentiable w.r.t. the predictions (used for loss AND gradient parity and for
the CPU-baseline training step).  Follows simpleAICV/detection/losses.py:
IoU family :54-123, RetinaLoss.forward :161-218, focal :220-261, box loss
:263-320, decode-for-IoU-loss :411-429, FCOSLoss.forward :462-511, focal
:513-548, IoU loss :550-586, centerness :588-610.  Target assignment is the
NumPy restatement in oracle/assign.py.
"""
import math

import numpy as np
import torch

from . import anchors as _anchors
from . import assign as _assign


def iou_family(b1, b2, iou_type='IoU'):
    """Element-wise IoU/GIoU/DIoU/CIoU/EIoU of [N,4] xyxy boxes (losses.py:54-123)."""
    tl = torch.max(b1[..., 0:2], b2[..., 0:2])
    br = torch.min(b1[..., 2:4], b2[..., 2:4])
    wh = torch.clamp(br - tl, min=0)
    inter = wh[..., 0] * wh[..., 1]
    wh1 = torch.clamp(b1[..., 2:4] - b1[..., 0:2], min=0)
    wh2 = torch.clamp(b2[..., 2:4] - b2[..., 0:2], min=0)
    union = torch.clamp(wh1[..., 0] * wh1[..., 1] + wh2[..., 0] * wh2[..., 1] - inter,
                        min=1e-4)
    iou = inter / union
    if iou_type == 'IoU':
        return iou
    etl = torch.min(b1[..., 0:2], b2[..., 0:2])
    ebr = torch.max(b1[..., 2:4], b2[..., 2:4])
    ewh = torch.clamp(ebr - etl, min=0)
    if iou_type == 'GIoU':
        earea = torch.clamp(ewh[..., 0] * ewh[..., 1], min=1e-4)
        return iou - (earea - union) / earea
    c2 = torch.clamp(ewh[..., 0]**2 + ewh[..., 1]**2, min=1e-4)
    c1 = (b1[..., 2:4] + b1[..., 0:2]) / 2
    c2c = (b2[..., 2:4] + b2[..., 0:2]) / 2
    p2 = (c1[..., 0] - c2c[..., 0])**2 + (c1[..., 1] - c2c[..., 1])**2
    if iou_type == 'DIoU':
        return iou - p2 / c2
    if iou_type == 'CIoU':
        v = (4 / math.pi**2) * torch.pow(
            torch.atan(wh2[:, 0] / wh2[:, 1]) - torch.atan(wh1[:, 0] / wh1[:, 1]), 2)
        with torch.no_grad():
            alpha = v / torch.clamp(1 - iou + v, min=1e-4)
        return iou - (p2 / c2 + v * alpha)
    if iou_type == 'EIoU':
        pw2 = (wh2[..., 0] - wh1[..., 0])**2
        ph2 = (wh2[..., 1] - wh1[..., 1])**2
        cw2 = torch.clamp(ewh[..., 0]**2, min=1e-4)
        ch2 = torch.clamp(ewh[..., 1]**2, min=1e-4)
        return iou - (p2 / c2 + pw2 / cw2 + ph2 / ch2)
    raise ValueError(iou_type)


def focal_sum(probs, labels, alpha, gamma):
    """Sum of the focal terms.  probs [M,C] already clamped, labels [M] (0 = bg, k = class k-1)."""
    C = probs.shape[1]
    onehot = torch.nn.functional.one_hot(labels.long(), C + 1)[:, 1:].float()
    is_pos = onehot == 1.
    a = torch.where(is_pos, torch.full_like(probs, alpha), torch.full_like(probs, 1. - alpha))
    pt = torch.where(is_pos, probs, 1. - probs)
    w = a * torch.pow(1. - pt, gamma)
    bce = -(onehot * torch.log(probs) + (1. - onehot) * torch.log(1. - probs))
    return (w * bce).sum()


def decode_txtytwth(reg, anchors):
    """losses.py:411-429."""
    awh = anchors[:, 2:4] - anchors[:, 0:2]
    actr = anchors[:, 0:2] + 0.5 * awh
    wh = torch.exp(reg[:, 2:4]) * awh
    ctr = reg[:, :2] * awh + actr
    return torch.cat([ctr - 0.5 * wh, ctr + 0.5 * wh], dim=1)


class RetinaLossOracle:

    def __init__(self, areas=_anchors.DEFAULT_AREAS, ratios=_anchors.DEFAULT_RATIOS,
                 scales=_anchors.DEFAULT_SCALES, strides=_anchors.DEFAULT_STRIDES,
                 alpha=0.25, gamma=2, beta=1.0 / 9.0, cls_loss_weight=1.,
                 box_loss_weight=1., box_loss_type='SmoothL1'):
        self.cfg = dict(areas=areas, ratios=ratios, scales=scales, strides=strides)
        self.alpha, self.gamma, self.beta = alpha, gamma, beta
        self.cls_w, self.box_w = cls_loss_weight, box_loss_weight
        self.box_loss_type = box_loss_type

    def targets(self, cls_preds, annotations):
        fs = [[p.shape[2], p.shape[1]] for p in cls_preds]
        anc = _anchors.retina_anchors(fs, **self.cfg)
        tg = _assign.retina_assign(anc, annotations.detach().cpu().numpy().astype(np.float32),
                                   self.box_loss_type)
        return anc, tg

    def __call__(self, preds, annotations):
        cls_preds, reg_preds = preds
        B = annotations.shape[0]
        anc, tg = self.targets(cls_preds, annotations)
        anc_t = torch.from_numpy(anc).unsqueeze(0).repeat(B, 1, 1).view(-1, 4)
        tg_t = torch.from_numpy(tg).view(-1, 5)
        cls = torch.cat([p.reshape(B, -1, p.shape[-1]) for p in cls_preds], dim=1)
        reg = torch.cat([p.reshape(B, -1, p.shape[-1]) for p in reg_preds], dim=1)
        cls = torch.clamp(cls, min=1e-4, max=1. - 1e-4).view(-1, cls.shape[-1])
        reg = reg.view(-1, 4)
        lab = tg_t[:, 4]
        keep = lab >= 0
        pos = lab > 0
        npos = int(pos.sum())
        if npos == 0:
            z = torch.tensor(0.)
            return {'cls_loss': self.cls_w * z, 'reg_loss': self.box_w * z.clone()}
        cls_loss = focal_sum(cls[keep], lab[keep], self.alpha, self.gamma) / npos
        rp, tp, ap = reg[pos], tg_t[pos], anc_t[pos]
        if self.box_loss_type == 'SmoothL1':
            x = torch.abs(rp - tp[:, 0:4])
            reg_loss = torch.where(x >= self.beta, x - 0.5 * self.beta,
                                   0.5 * (x**2) / self.beta).sum() / npos
        else:
            boxes = decode_txtytwth(rp, ap)
            reg_loss = (1 - iou_family(boxes, tp[:, 0:4], self.box_loss_type)).sum() / npos
        return {'cls_loss': self.cls_w * cls_loss, 'reg_loss': self.box_w * reg_loss}


class FCOSLossOracle:

    def __init__(self, strides=_anchors.DEFAULT_STRIDES, mi=_assign.FCOS_MI, alpha=0.25,
                 gamma=2., cls_loss_weight=1., box_loss_weight=1.,
                 center_ness_loss_weight=1., box_loss_iou_type='GIoU',
                 center_sample_radius=1.5, use_center_sample=True):
        self.strides, self.mi = strides, mi
        self.alpha, self.gamma = alpha, gamma
        self.cls_w, self.box_w, self.ctr_w = cls_loss_weight, box_loss_weight, center_ness_loss_weight
        self.iou_type = box_loss_iou_type
        self.radius, self.use_center_sample = center_sample_radius, use_center_sample

    def targets(self, cls_preds, annotations):
        fs = [[p.shape[2], p.shape[1]] for p in cls_preds]
        pos = _anchors.fcos_positions(fs, self.strides)
        return _assign.fcos_assign(pos, fs, annotations.detach().cpu().numpy().astype(np.float32),
                                   self.strides, self.mi, self.radius, self.use_center_sample)

    def __call__(self, preds, annotations):
        cls_preds, reg_preds, ctr_preds = preds
        B = annotations.shape[0]
        tg = torch.from_numpy(self.targets(cls_preds, annotations)).view(-1, 8)
        cls = torch.cat([p.reshape(B, -1, p.shape[-1]) for p in cls_preds], dim=1)
        reg = torch.cat([p.reshape(B, -1, p.shape[-1]) for p in reg_preds], dim=1).view(-1, 4)
        ctr = torch.cat([p.reshape(B, -1, p.shape[-1]) for p in ctr_preds], dim=1).view(-1, 1)
        cls = torch.clamp(cls.view(-1, cls.shape[-1]), min=1e-4, max=1. - 1e-4)
        ctr = torch.clamp(ctr, min=1e-4, max=1. - 1e-4)
        pos = tg[:, 4] > 0
        npos = int(pos.sum())
        if npos == 0:
            z = torch.tensor(0.)
            return {'cls_loss': self.cls_w * z, 'reg_loss': self.box_w * z.clone(),
                    'center_ness_loss': self.ctr_w * z.clone()}
        cls_loss = focal_sum(cls, tg[:, 4], self.alpha, self.gamma) / npos
        tp = tg[pos]
        rp = torch.exp(reg)[pos]
        pred = torch.cat([tp[:, 6:8] - rp[:, 0:2], tp[:, 6:8] + rp[:, 2:4]], dim=1)
        gt = torch.cat([tp[:, 6:8] - tp[:, 0:2], tp[:, 6:8] + tp[:, 2:4]], dim=1)
        reg_loss = ((1 - iou_family(pred, gt, self.iou_type)) * tp[:, 5]).sum() / npos
        cp = ctr[pos]
        ct = tp[:, 5:6]
        ctr_loss = (-(ct * torch.log(cp) + (1. - ct) * torch.log(1. - cp))).sum() / npos
        return {'cls_loss': self.cls_w * cls_loss, 'reg_loss': self.box_w * reg_loss,
                'center_ness_loss': self.ctr_w * ctr_loss}
