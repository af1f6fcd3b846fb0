"""ResNet classifiers (ResNet18/34 BasicBlock, ResNet50/101/152 Bottleneck) with the reference's constructor, module
tree and state_dict keys (simpleAICV/classification/backbones/resnet.py:158-300), executed by the HIP engine.

As for the detectors, the torch.nn modules are parameter containers only; forward runs the hand-written
kernels (implicit-GEMM convolutions with fused BatchNorm statistics, global average pool, FC as a 1x1 GEMM
problem) and there is no CPU / eager execution path.
"""
import torch
import torch.nn as nn

from ...engine import ClassifierEngine
from ...detection.models._arena import ParamArena, deepcopy_without_runtime, getstate_without_runtime, refuse_sync_batchnorm
from ...detection.models._base import _resnet_trunk

__all__ = ['resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152', 'ResNet']


class _ClassifierFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, model, images, *params):
        ctx.model = model
        return model._engine.forward(images, training=model.training, keep=True)

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        model._engine.backward(dlogits.contiguous())
        return (None, None) + tuple(model._arena.grad_views[name] for name, _ in model._param_list)


class ResNet(nn.Module):
    __deepcopy__ = deepcopy_without_runtime
    __getstate__ = getstate_without_runtime

    STEM = 'imagenet'

    def __init__(self, block='Bottleneck', layer_nums=(3, 4, 6, 3), inplanes=64, num_classes=1000,
                 use_gradient_checkpoint=False):
        super().__init__()
        block = getattr(block, '__name__', block)  # the reference passes the block class
        if block not in ('Bottleneck', 'BasicBlock') or inplanes != 64 or len(layer_nums) != 4:
            raise NotImplementedError('ResNet with Bottleneck/BasicBlock stages and 64 stem channels is what the '
                                      'MI355X engine executes')
        if use_gradient_checkpoint:
            raise NotImplementedError('use_gradient_checkpoint: activations are kept in HBM (288 GB) instead')
        self.block = block
        self.expansion = 4 if block == 'Bottleneck' else 1
        self.layer_nums, self.num_classes = list(layer_nums), num_classes
        self.planes = [64, 128, 256, 512]
        self.inplanes = 512 * self.expansion
        self.use_gradient_checkpoint = use_gradient_checkpoint
        _resnet_trunk(self, block, layer_nums, 64, self.STEM)  # kaiming fan_out / BN 1,0 init (resnet.py:205-212)
        if self.STEM == 'imagenet':
            self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * self.expansion, num_classes)
        self._arena = self._engine = self._param_list = None
        self._amp = False

    def set_amp(self, enabled=True):
        """The reference's `use_amp` (autocast around forward + loss, tools/scripts.py:140-158): run the trunk on the
        bf16 matrix cores -- bf16 activations and weight panels, fp32 accumulation, BatchNorm statistics, loss,
        master weights, gradients and optimizer state."""
        self._amp = bool(enabled)
        if self._engine is not None:
            self._engine.set_amp(self._amp, self._arena)
        return self

    def prepare(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('sacv_amd ResNet runs on MI355X only: move the model to the GPU (.cuda()); there is '
                               'no CPU/PyTorch fallback path')
        refuse_sync_batchnorm(self)
        self._arena = ParamArena(self)
        self._param_list = list(self._arena.order)
        tensors = {n: p.data for n, p in self.named_parameters()}
        tensors.update({n: b for n, b in self.named_buffers()})
        self._engine = ClassifierEngine(tensors, self._arena.grad_views, self.num_classes, block=self.block,
                                        layers=self.layer_nums, stem=self.STEM)
        self._engine.set_amp(self._amp, self._arena)
        return self

    def _apply(self, fn, recurse=True):
        """.cuda() / .to(device) builds the arenas at once (see HipDetector._apply)."""
        out = super()._apply(fn, recurse)
        p = next(self.parameters(), None)
        if p is not None and p.device.type == 'cuda':
            self.prepare()
        return out

    def _ready(self):
        if self._arena is None or not self._arena.is_current():
            self.prepare()

    @property
    def arena(self):
        self._ready()
        return self._arena

    @property
    def engine(self):
        self._ready()
        return self._engine

    @torch.compiler.disable  # config.use_compile (torch.compile(model), tools/train_detection_model.py:176-178): nothing to trace --
    # the forward pass is one call into the HIP executor; the compiled wrapper runs it as is
    def forward(self, x):
        if x.device.type != 'cuda':
            raise RuntimeError('sacv_amd ResNet runs on MI355X only: inputs must be on the GPU (no CPU fallback)')
        self._ready()
        x = x.float()
        if torch.is_grad_enabled() and self.training:
            return _ClassifierFunction.apply(self, x, *[p for _, p in self._param_list])
        return self._engine.forward(x, training=self.training, keep=False)


def resnet18(**kwargs):
    return ResNet('BasicBlock', [2, 2, 2, 2], 64, **kwargs)


def resnet34(**kwargs):
    return ResNet('BasicBlock', [3, 4, 6, 3], 64, **kwargs)


def resnet50(**kwargs):
    return ResNet('Bottleneck', [3, 4, 6, 3], 64, **kwargs)


def resnet101(**kwargs):
    return ResNet('Bottleneck', [3, 4, 23, 3], 64, **kwargs)


def resnet152(**kwargs):
    return ResNet('Bottleneck', [3, 8, 36, 3], 64, **kwargs)
