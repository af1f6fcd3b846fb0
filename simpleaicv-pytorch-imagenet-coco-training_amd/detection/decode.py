"""RetinaDecoder with the reference's constructor / call contract (simpleAICV/detection/
decode.py:175-271 + DecodeMethod :107-172 + DetNMSMethod :26-104), run on the device:
argmax/score, box decode with int32 truncation, score > threshold, exact top-n, hard
class-agnostic NMS (python_nms or diou_python_nms), pad to max_object_num.  Only the three
small result arrays ([B,100], [B,100], [B,100,4]) cross PCIe; they are returned as host
NumPy float32 arrays because the caller edits them in place (tools/scripts.py:742-758).
"""
import numpy as np
import torch

from .. import ops
from .losses import AnchorCache, PointCache

__all__ = ['RetinaDecoder', 'FCOSDecoder']

_NMS = {'python_nms': 0, 'diou_python_nms': 1, 'torch_nms': 2}  # torch_nms: torchvision.ops.nms semantics
# Extension (BASELINE.json north_star: "batched soft-NMS postprocess"; the reference's DetNMSMethod has none, parity
# unpinned): Soft-NMS of Bodla et al. (ICCV 2017), linear or gaussian decay, batched over the images on the device.
_SOFT = {'soft_nms': 0, 'soft_nms_linear': 0, 'soft_nms_gaussian': 1}
_NMS_TYPES = list(_NMS) + list(_SOFT)


def _soft_args(nms_type, sigma, prune_score):
    if nms_type not in _SOFT:
        return None
    return (_SOFT[nms_type], float(np.float32(sigma)), float(np.float32(prune_score)))


class RetinaDecoder:

    def __init__(self, areas=[[32, 32], [64, 64], [128, 128], [256, 256], [512, 512]], ratios=[0.5, 1, 2],
                 scales=[2**0, 2**(1.0 / 3.0), 2**(2.0 / 3.0)], strides=[8, 16, 32, 64, 128], max_object_num=100,
                 min_score_threshold=0.05, topn=1000, nms_type='python_nms', nms_threshold=0.5, soft_nms_sigma=0.5,
                 soft_nms_prune_score=0.001):
        assert nms_type in _NMS_TYPES, 'wrong nms type!'
        self.anchors = AnchorCache(areas, ratios, scales, strides)
        self.max_object_num, self.min_score_threshold, self.topn = max_object_num, min_score_threshold, topn
        self.nms_type, self.nms_threshold = nms_type, nms_threshold
        self.soft = _soft_args(nms_type, soft_nms_sigma, soft_nms_prune_score)

    def device_call(self, preds):
        """-> (scores [B,M], classes [B,M], boxes [B,M,4]) fp32 tensors on the device."""
        cls_preds, reg_preds = preds
        device = cls_preds[0].device
        if device.type != 'cuda':
            raise RuntimeError('sacv_amd RetinaDecoder runs on MI355X only (no CPU fallback)')
        feature_sizes = [[p.shape[2], p.shape[1]] for p in cls_preds]
        anchors = self.anchors.get(feature_sizes, device)
        cls_c = [p.detach().float().contiguous() for p in cls_preds]
        reg_c = [p.detach().float().contiguous() for p in reg_preds]
        return ops.decode_topk_nms(0, cls_c, reg_c, None, anchors, float(np.float32(self.min_score_threshold)),
                                   self.topn, _NMS.get(self.nms_type, 0), float(np.float32(self.nms_threshold)),
                                   self.max_object_num, soft=self.soft)

    def __call__(self, preds):
        s, c, b = self.device_call(preds)
        return [s.cpu().numpy(), c.cpu().numpy(), b.cpu().numpy()]


class FCOSDecoder:
    """decode.py:274-364: score = sqrt(cls * centerness), boxes = point -/+ exp(ltrb) truncated to int32."""

    def __init__(self, strides=[8, 16, 32, 64, 128], max_object_num=100, min_score_threshold=0.05, topn=1000,
                 nms_type='python_nms', nms_threshold=0.6, soft_nms_sigma=0.5, soft_nms_prune_score=0.001):
        assert nms_type in _NMS_TYPES, 'wrong nms type!'
        self.positions = PointCache(strides)
        self.max_object_num, self.min_score_threshold, self.topn = max_object_num, min_score_threshold, topn
        self.nms_type, self.nms_threshold = nms_type, nms_threshold
        self.soft = _soft_args(nms_type, soft_nms_sigma, soft_nms_prune_score)

    def device_call(self, preds):
        cls_preds, reg_preds, ctr_preds = preds
        device = cls_preds[0].device
        if device.type != 'cuda':
            raise RuntimeError('sacv_amd FCOSDecoder runs on MI355X only (no CPU fallback)')
        feature_sizes = [[p.shape[2], p.shape[1]] for p in cls_preds]
        points = self.positions.get(feature_sizes, device)
        f = lambda lst: [p.detach().float().contiguous() for p in lst]
        return ops.decode_topk_nms(1, f(cls_preds), f(reg_preds), f(ctr_preds), points,
                                   float(np.float32(self.min_score_threshold)), self.topn,
                                   _NMS.get(self.nms_type, 0), float(np.float32(self.nms_threshold)),
                                   self.max_object_num, soft=self.soft)

    def __call__(self, preds):
        s, c, b = self.device_call(preds)
        return [s.cpu().numpy(), c.cpu().numpy(), b.cpu().numpy()]
