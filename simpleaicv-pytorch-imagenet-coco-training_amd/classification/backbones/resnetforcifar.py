"""CIFAR ResNets with the reference's constructor, module tree and state_dict keys
(simpleAICV/classification/backbones/resnetforcifar.py:27-125: 3x3/s1 stem, no max pool, 4 residual stages,
global average pool, FC), executed by the HIP engine -- BASELINE.json config #1 is resnet18cifar, CIFAR-100, bs 32."""
from .resnet import ResNet

__all__ = ['resnet18cifar', 'resnet34cifar', 'resnet50cifar', 'resnet101cifar', 'resnet152cifar', 'ResNetCifar']


class ResNetCifar(ResNet):
    STEM = 'cifar'

    def __init__(self, block='BasicBlock', layer_nums=(2, 2, 2, 2), inplanes=64, num_classes=100):
        super().__init__(block, layer_nums, inplanes, num_classes=num_classes)


def resnet18cifar(**kwargs):
    return ResNetCifar('BasicBlock', [2, 2, 2, 2], 64, **kwargs)


def resnet34cifar(**kwargs):
    return ResNetCifar('BasicBlock', [3, 4, 6, 3], 64, **kwargs)


def resnet50cifar(**kwargs):
    return ResNetCifar('Bottleneck', [3, 4, 6, 3], 64, **kwargs)


def resnet101cifar(**kwargs):
    return ResNetCifar('Bottleneck', [3, 4, 23, 3], 64, **kwargs)


def resnet152cifar(**kwargs):
    return ResNetCifar('Bottleneck', [3, 8, 36, 3], 64, **kwargs)
