"""RetinaLoss / FCOSLoss with the reference's constructor / call contract (simpleAICV/detection/losses.py:
126-429 and 432-833), computed by fused HIP kernels: device-side anchors (anchor.py:18-86), IoU target
assignment (:322-409), focal loss with the [1e-4, 1-1e-4] probability clamp (:196, :220-261)
and SmoothL1 / IoU / GIoU box loss (:263-320, :411-429).  Forward and the gradient w.r.t.
the predictions are produced by the same kernel pass; autograd only scales them.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops

__all__ = ['RetinaLoss', 'FCOSLoss']

_BOX_TYPES = {'SmoothL1': ops.BOX_SMOOTH_L1, 'IoU': ops.BOX_IOU, 'GIoU': ops.BOX_GIOU, 'DIoU': ops.BOX_DIOU,
              'CIoU': ops.BOX_CIOU, 'EIoU': ops.BOX_EIOU}
_IOU_TYPES = {'IoU': 0, 'GIoU': 1, 'DIoU': 2, 'CIoU': 3, 'EIoU': 4}


class _ScaledGrad(torch.autograd.Function):
    """loss value computed by the kernels; backward hands out the pre-computed gradients."""

    @staticmethod
    def forward(ctx, value, grads, *preds):
        ctx.grads = grads
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        return (None, None) + tuple(g * grad_out for g in ctx.grads)


class AnchorCache:
    """Anchors depend only on the feature-map sizes: generated once per shape on the device
    (the reference regenerates them on the host and uploads them every step, losses.py:169-179)."""

    def __init__(self, areas, ratios, scales, strides):
        self.cfg = (areas, ratios, scales, strides)
        self._cache = {}

    def get(self, feature_sizes, device):
        key = (tuple(map(tuple, feature_sizes)), str(device))
        if key not in self._cache:
            areas, ratios, scales, strides = self.cfg
            self._cache[key] = ops.retina_anchors(feature_sizes, areas, ratios, scales, strides, device)
        return self._cache[key]


class RetinaLoss(nn.Module):

    def __init__(self, areas=[[32, 32], [64, 64], [128, 128], [256, 256], [512, 512]], ratios=[0.5, 1, 2],
                 scales=[2**0, 2**(1.0 / 3.0), 2**(2.0 / 3.0)], strides=[8, 16, 32, 64, 128], alpha=0.25, gamma=2,
                 beta=1.0 / 9.0, cls_loss_weight=1., box_loss_weight=1., box_loss_type='SmoothL1'):
        super().__init__()
        assert box_loss_type in ['SmoothL1', 'IoU', 'GIoU', 'DIoU', 'CIoU', 'EIoU'], 'wrong IoU type!'
        self.anchors = AnchorCache(areas, ratios, scales, strides)
        self.alpha, self.gamma, self.beta = alpha, gamma, beta
        self.cls_loss_weight, self.box_loss_weight = cls_loss_weight, box_loss_weight
        self.box_loss_type = box_loss_type

    def fused(self, preds, annotations, want_grad=True):
        """Kernel-level entry: -> (loss_values f32[2] on device, counts int32[2], dcls, dreg).

        No host synchronisation.  loss_values = (cls_loss, reg_loss) already weighted and divided
        by the positive count (0 when there are no positives); dcls/dreg are gradients of the
        SUM of both losses w.r.t. the predictions."""
        cls_preds, reg_preds = preds
        device = annotations.device
        if device.type != 'cuda':
            raise RuntimeError('sacv_amd RetinaLoss runs on MI355X only (no CPU fallback)')
        feature_sizes = [[p.shape[2], p.shape[1]] for p in cls_preds]
        anchors = self.anchors.get(feature_sizes, device)
        annotations = annotations.float().contiguous()
        targets, counts = ops.retina_assign(anchors, annotations, smooth_l1=self.box_loss_type == 'SmoothL1')
        values = torch.empty(2, device=device, dtype=torch.float32)
        cls_c = [p if p.is_contiguous() else p.contiguous() for p in cls_preds]
        reg_c = [p if p.is_contiguous() else p.contiguous() for p in reg_preds]
        dcls = ops.focal_loss(cls_c, targets, counts, self.alpha, float(self.gamma), self.cls_loss_weight, values, 0,
                              want_grad=want_grad)
        dreg = ops.retina_box_loss(reg_c, targets, anchors, counts, _BOX_TYPES[self.box_loss_type], self.beta,
                                   self.box_loss_weight, values, 1, want_grad=want_grad)
        return values, counts, dcls, dreg

    def forward(self, preds, annotations):
        cls_preds, reg_preds = preds
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in list(cls_preds) + list(reg_preds))
        values, counts, dcls, dreg = self.fused(preds, annotations, want_grad=need_grad)
        # the reference returns plain zeros WITHOUT grad_fn when the batch has no positive anchor
        # (losses.py:234-235, 279-280); mirroring that needs the count on the host (one 8-byte read)
        if int(counts[0].item()) == 0:
            zero = torch.tensor(0.).to(annotations.device)
            return {'cls_loss': self.cls_loss_weight * zero, 'reg_loss': self.box_loss_weight * zero.clone()}
        if need_grad:
            cls_loss = _ScaledGrad.apply(values[0], dcls, *cls_preds)
            reg_loss = _ScaledGrad.apply(values[1], dreg, *reg_preds)
        else:
            cls_loss, reg_loss = values[0].clone(), values[1].clone()
        return {'cls_loss': cls_loss, 'reg_loss': reg_loss}


class PointCache:
    """FCOS point centres per feature-map shape, generated once on the device (anchor.py:94-130)."""

    def __init__(self, strides):
        self.strides = strides
        self._cache = {}

    def get(self, feature_sizes, device):
        key = (tuple(map(tuple, feature_sizes)), str(device))
        if key not in self._cache:
            self._cache[key] = ops.fcos_points(feature_sizes, self.strides, device)
        return self._cache[key]


class FCOSLoss(nn.Module):

    def __init__(self, strides=[8, 16, 32, 64, 128],
                 mi=[[-1, 64], [64, 128], [128, 256], [256, 512], [512, 100000000]], alpha=0.25, gamma=2.,
                 cls_loss_weight=1., box_loss_weight=1., center_ness_loss_weight=1., box_loss_iou_type='GIoU',
                 center_sample_radius=1.5, use_center_sample=True):
        super().__init__()
        assert box_loss_iou_type in ['IoU', 'GIoU', 'DIoU', 'CIoU', 'EIoU'], 'wrong IoU type!'
        self.positions = PointCache(strides)
        self.alpha, self.gamma = alpha, gamma
        self.strides, self.mi = strides, mi
        self.cls_loss_weight, self.box_loss_weight = cls_loss_weight, box_loss_weight
        self.center_ness_loss_weight = center_ness_loss_weight
        self.box_loss_iou_type = box_loss_iou_type
        self.center_sample_radius, self.use_center_sample = center_sample_radius, use_center_sample

    def fused(self, preds, annotations, want_grad=True):
        """-> (loss_values f32[3] = cls, reg, center_ness; counts; dcls, dreg, dctr).  No host sync."""
        cls_preds, reg_preds, ctr_preds = preds
        device = annotations.device
        if device.type != 'cuda':
            raise RuntimeError('sacv_amd FCOSLoss runs on MI355X only (no CPU fallback)')
        feature_sizes = [[p.shape[2], p.shape[1]] for p in cls_preds]
        points = self.positions.get(feature_sizes, device)
        level_off = [0]
        for w, h in feature_sizes:
            level_off.append(level_off[-1] + w * h)
        annotations = annotations.float().contiguous()
        targets, counts = ops.fcos_assign(points, level_off, self.strides, self.mi, annotations,
                                          float(self.center_sample_radius), self.use_center_sample)
        values = torch.empty(3, device=device, dtype=torch.float32)
        cls_c = [p if p.is_contiguous() else p.contiguous() for p in cls_preds]
        reg_c = [p if p.is_contiguous() else p.contiguous() for p in reg_preds]
        ctr_c = [p if p.is_contiguous() else p.contiguous() for p in ctr_preds]
        dcls = ops.focal_loss(cls_c, targets, counts, self.alpha, float(self.gamma), self.cls_loss_weight, values, 0,
                              want_grad=want_grad, tstride=8)
        dreg, dctr = ops.fcos_reg_ctr_loss(reg_c, ctr_c, targets, counts, _IOU_TYPES[self.box_loss_iou_type],
                                           self.box_loss_weight, self.center_ness_loss_weight, values, 1, 2,
                                           want_grad=want_grad)
        return values, counts, dcls, dreg, dctr

    def forward(self, preds, annotations):
        cls_preds, reg_preds, ctr_preds = preds
        allp = list(cls_preds) + list(reg_preds) + list(ctr_preds)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in allp)
        values, counts, dcls, dreg, dctr = self.fused(preds, annotations, want_grad=need_grad)
        if int(counts[0].item()) == 0:  # reference: plain zeros without grad_fn (losses.py:523-524 etc.)
            zero = torch.tensor(0.).to(annotations.device)
            return {'cls_loss': self.cls_loss_weight * zero, 'reg_loss': self.box_loss_weight * zero.clone(),
                    'center_ness_loss': self.center_ness_loss_weight * zero.clone()}
        if need_grad:
            return {'cls_loss': _ScaledGrad.apply(values[0], dcls, *cls_preds),
                    'reg_loss': _ScaledGrad.apply(values[1], dreg, *reg_preds),
                    'center_ness_loss': _ScaledGrad.apply(values[2], dctr, *ctr_preds)}
        return {'cls_loss': values[0].clone(), 'reg_loss': values[1].clone(), 'center_ness_loss': values[2].clone()}
