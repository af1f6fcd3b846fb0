"""FCOS-ResNet50 with the reference's constructor, module tree and state_dict keys
(simpleAICV/detection/models/fcos.py:27-115, head.py:88-181), executed by the HIP engine."""
import math

import torch
import torch.nn as nn

from ._base import HipDetector, _detection_backbone, _fpn

__all__ = ['resnet18_fcos', 'resnet34_fcos', 'resnet50_fcos', 'resnet101_fcos', 'resnet152_fcos', 'FCOS']


def _fcos_head(planes, num_classes):
    h = nn.Module()
    for name in ('cls_head', 'reg_head'):
        layers = []
        for _ in range(4):
            layers += [nn.Conv2d(planes, planes, 3, padding=1, bias=False), nn.GroupNorm(32, planes),
                       nn.ReLU(inplace=True)]
        setattr(h, name, nn.Sequential(*layers))
    h.cls_out = nn.Conv2d(planes, num_classes, 3, padding=1)
    h.reg_out = nn.Conv2d(planes, 4, 3, padding=1)
    h.center_out = nn.Conv2d(planes, 1, 3, padding=1)
    for m in h.modules():  # head.py:152-160
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, std=0.01)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
    h.cls_out.bias.data.fill_(-math.log((1 - 0.01) / 0.01))
    return h


class FCOS(HipDetector):
    _fcos = True

    def __init__(self, backbone_type='resnet50backbone', backbone_pretrained_path='', planes=256, num_classes=80,
                 use_gradient_checkpoint=False):
        super().__init__()
        # use_gradient_checkpoint (torch.utils.checkpoint in the reference, retinanet.py:64-67, fcos.py likewise) trades
        # memory for a second forward pass and leaves every result unchanged; here the activations of the 2 x 800x1333
        # step (a few GB of 288) stay in HBM, so the flag is accepted and has nothing to switch
        self.planes, self.num_classes = planes, num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.backbone = _detection_backbone(backbone_type)  # raises for anything but the five ResNets
        self.fpn = _fpn(self.backbone.out_channels[1:4], planes, use_p5=True)
        self.clsregcnt_head = _fcos_head(planes, num_classes)
        self.scales = nn.Parameter(torch.tensor([1., 1., 1., 1., 1.], dtype=torch.float32))
        if backbone_pretrained_path:
            self._load_backbone(backbone_pretrained_path)
        else:
            print('no backbone pretrained model!')
        self._init_runtime()


def _factory(backbone_type):
    def make(backbone_pretrained_path='', **kwargs):  # fcos.py:89-125
        return FCOS(backbone_type, backbone_pretrained_path=backbone_pretrained_path, **kwargs)
    make.__name__ = backbone_type.replace('backbone', '_fcos')
    return make


resnet18_fcos = _factory('resnet18backbone')
resnet34_fcos = _factory('resnet34backbone')
resnet50_fcos = _factory('resnet50backbone')
resnet101_fcos = _factory('resnet101backbone')
resnet152_fcos = _factory('resnet152backbone')
