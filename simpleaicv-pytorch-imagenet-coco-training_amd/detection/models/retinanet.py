"""RetinaNet on the reference's five ResNet backbones (ResNet50 = the graded recipe) with the reference's constructor, module tree and state_dict keys
(simpleAICV/detection/models/retinanet.py:30-120, backbones/resnet.py:27-113, fpn.py:14-58,
head.py:15-85), executed by the HIP engine instead of torch.nn layers.

The torch.nn modules are parameter CONTAINERS only (they give the reference key names, `.cuda()`, DDP and
checkpoint compatibility); their forward is never called.  The model fails loudly on CPU tensors: there is
no non-HIP execution path.
"""
from ._base import HipDetector, _detection_backbone, _fpn, _head

__all__ = ['resnet18_retinanet', 'resnet34_retinanet', 'resnet50_retinanet', 'resnet101_retinanet',
           'resnet152_retinanet', 'RetinaNet']


class RetinaNet(HipDetector):

    def __init__(self, backbone_type='resnet50backbone', backbone_pretrained_path='', planes=256, num_anchors=9,
                 num_classes=80, use_gradient_checkpoint=False):
        super().__init__()
        # use_gradient_checkpoint (torch.utils.checkpoint in the reference, retinanet.py:64-67, fcos.py likewise) trades
        # memory for a second forward pass and leaves every result unchanged; here the activations of the 2 x 800x1333
        # step (a few GB of 288) stay in HBM, so the flag is accepted and has nothing to switch
        self.planes, self.num_anchors, self.num_classes = planes, num_anchors, num_classes
        self.use_gradient_checkpoint = use_gradient_checkpoint
        self.backbone = _detection_backbone(backbone_type)  # raises for anything but the five ResNets
        self.fpn = _fpn(self.backbone.out_channels[1:4], planes, use_p5=False)
        self.cls_head = _head(planes, num_anchors * num_classes, 'cls_head', 'cls_out', cls_prior=0.01)
        self.reg_head = _head(planes, num_anchors * 4, 'reg_head', 'reg_out')
        if backbone_pretrained_path:
            self._load_backbone(backbone_pretrained_path)
        else:
            print('no backbone pretrained model!')
        self._init_runtime()


def _factory(backbone_type):
    def make(backbone_pretrained_path='', **kwargs):  # retinanet.py:94-130
        return RetinaNet(backbone_type, backbone_pretrained_path=backbone_pretrained_path, **kwargs)
    make.__name__ = backbone_type.replace('backbone', '_retinanet')
    return make


resnet18_retinanet = _factory('resnet18backbone')
resnet34_retinanet = _factory('resnet34backbone')
resnet50_retinanet = _factory('resnet50backbone')
resnet101_retinanet = _factory('resnet101backbone')
resnet152_retinanet = _factory('resnet152backbone')
