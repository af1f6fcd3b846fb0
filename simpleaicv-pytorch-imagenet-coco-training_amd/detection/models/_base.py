"""Shared machinery of the HIP-executed detectors: parameter containers with the reference's key layout,
the flat arenas, the executor and the single autograd node around the whole network."""
import math

import torch
import torch.nn as nn

from ...engine import DetectorEngine, RESNET50_LAYERS
from ._arena import ParamArena, deepcopy_without_runtime, getstate_without_runtime, refuse_sync_batchnorm

def _conv_bn_act(cin, cout, k, stride, pad, act=True):
    """Container with the key layout of ConvBnActBlock: layer.0 = conv (no bias), layer.1 = BN."""
    blk = nn.Module()
    blk.layer = nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False),
                              nn.BatchNorm2d(cout), nn.ReLU(inplace=True) if act else nn.Sequential())
    return blk


def _bottleneck(cin, planes, stride):
    blk = nn.Module()
    blk.conv1 = _conv_bn_act(cin, planes, 1, 1, 0)
    blk.conv2 = _conv_bn_act(planes, planes, 3, stride, 1)
    blk.conv3 = _conv_bn_act(planes, planes * 4, 1, 1, 0, act=False)
    if stride != 1 or cin != planes * 4:
        blk.downsample_conv = _conv_bn_act(cin, planes * 4, 1, stride, 0, act=False)
    return blk


def _basic_block(cin, planes, stride):
    blk = nn.Module()
    blk.conv1 = _conv_bn_act(cin, planes, 3, stride, 1)
    blk.conv2 = _conv_bn_act(planes, planes, 3, 1, 1, act=False)
    if stride != 1 or cin != planes:
        blk.downsample_conv = _conv_bn_act(cin, planes, 1, stride, 0, act=False)
    return blk


def _resnet_trunk(bb=None, block='Bottleneck', layer_nums=RESNET50_LAYERS, inplanes=64, stem='imagenet'):
    """ResNet trunk containers (conv1, layer1..layer4 with the reference's key layout and initialisers,
    backbones/resnet.py:158-212, resnetforcifar.py:27-75); populates `bb` (the bare classifiers register them on
    themselves)."""
    bb = nn.Module() if bb is None else bb
    bb.conv1 = _conv_bn_act(3, inplanes, 7, 2, 3) if stem == 'imagenet' else _conv_bn_act(3, inplanes, 3, 1, 1)
    expansion = 4 if block == 'Bottleneck' else 1
    make = _bottleneck if block == 'Bottleneck' else _basic_block
    cin = inplanes
    for li, (n, planes) in enumerate(zip(layer_nums, (inplanes, inplanes * 2, inplanes * 4, inplanes * 8)), start=1):
        blocks = []
        for bi in range(n):
            blocks.append(make(cin, planes, 2 if (bi == 0 and li > 1) else 1))
            cin = planes * expansion
        setattr(bb, f'layer{li}', nn.Sequential(*blocks))
    for m in bb.modules():  # backbones/resnet.py:76-83
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    bb.out_channels = [inplanes * expansion, inplanes * 2 * expansion, inplanes * 4 * expansion,
                       inplanes * 8 * expansion]
    return bb


def _resnet50_backbone(bb=None):
    return _resnet_trunk(bb)


# detection/models/backbones/resnet.py:127-169: the five detection backbones (block type, stage depths)
DETECTION_BACKBONES = {'resnet18backbone': ('BasicBlock', (2, 2, 2, 2)), 'resnet34backbone': ('BasicBlock', (3, 4, 6, 3)),
                       'resnet50backbone': ('Bottleneck', (3, 4, 6, 3)), 'resnet101backbone': ('Bottleneck', (3, 4, 23, 3)),
                       'resnet152backbone': ('Bottleneck', (3, 8, 36, 3))}


def _detection_backbone(backbone_type):
    if backbone_type not in DETECTION_BACKBONES:
        raise NotImplementedError(f'{backbone_type}: the MI355X hot path runs the ResNet backbones '
                                  f'{sorted(DETECTION_BACKBONES)} (SURVEY.md 8a-1)')
    block, layers = DETECTION_BACKBONES[backbone_type]
    bb = _resnet_trunk(None, block=block, layer_nums=layers)
    bb.block, bb.layer_nums = block, layers
    return bb


def _fpn(inplanes, planes, use_p5):
    f = nn.Module()
    f.P3_1 = nn.Conv2d(inplanes[0], planes, 1)
    f.P3_2 = nn.Conv2d(planes, planes, 3, padding=1)
    f.P4_1 = nn.Conv2d(inplanes[1], planes, 1)
    f.P4_2 = nn.Conv2d(planes, planes, 3, padding=1)
    f.P5_1 = nn.Conv2d(inplanes[2], planes, 1)
    f.P5_2 = nn.Conv2d(planes, planes, 3, padding=1)
    f.P6 = nn.Conv2d(planes if use_p5 else inplanes[2], planes, 3, stride=2, padding=1)
    f.P7 = nn.Sequential(nn.ReLU(), nn.Conv2d(planes, planes, 3, stride=2, padding=1))
    return f


def _head(planes, n_out, tower_name, out_name, cls_prior=None):
    h = nn.Module()
    layers = []
    for _ in range(4):
        layers += [nn.Conv2d(planes, planes, 3, padding=1), nn.ReLU(inplace=True)]
    setattr(h, tower_name, nn.Sequential(*layers))
    setattr(h, out_name, nn.Conv2d(planes, n_out, 3, padding=1))
    for m in h.modules():  # head.py:36-44
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)
    if cls_prior is not None:
        getattr(h, out_name).bias.data.fill_(-math.log((1 - cls_prior) / cls_prior))
    return h


class _DetectorFunction(torch.autograd.Function):
    """autograd boundary of the whole detector: one node, gradients come from engine.backward."""

    @staticmethod
    def forward(ctx, model, images, *params):
        groups = model._engine.forward(images, training=model.training, keep=True)
        ctx.model = model
        ctx.n_levels = len(groups[0])
        ctx.n_groups = len(groups)
        return tuple(t for grp in groups for t in grp)

    @staticmethod
    def backward(ctx, *grads):
        model = ctx.model
        n = ctx.n_levels
        outs = model._last_outputs
        gs = [g if g is not None else torch.zeros_like(o) for g, o in zip(grads, outs)]
        gs = [g.contiguous() for g in gs]
        model._engine.backward(*[gs[k * n:(k + 1) * n] for k in range(ctx.n_groups)])
        arena = model._arena
        return (None, None) + tuple(arena.grad_views[name] for name, _ in model._param_list)




class HipDetector(nn.Module):
    """nn.Module whose forward is the HIP executor.  Subclasses build the parameter containers and set
    `_fcos`; there is no non-HIP execution path."""
    __deepcopy__ = deepcopy_without_runtime
    __getstate__ = getstate_without_runtime
    _fcos = False
    num_anchors = 1

    def _init_runtime(self):
        self._arena = None
        self._engine = None
        self._param_list = None
        self._last_outputs = None
        self._co_handle = None
        self._co_params = None

    def _load_backbone(self, path):
        """Name-and-shape matched warm start (simpleAICV/classification/common.py:758-784)."""
        saved = torch.load(path, map_location='cpu', weights_only=True)
        own = self.backbone.state_dict()
        keep = {k: v for k, v in saved.items() if k in own and v.shape == own[k].shape}
        own.update(keep)
        self.backbone.load_state_dict(own)

    def prepare(self):
        """(Re)build the flat arenas and the executor; called lazily by forward."""
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError(f'sacv_amd {type(self).__name__} runs on MI355X only: move the model to the GPU '
                               '(.cuda()); there is no CPU/PyTorch fallback path')
        refuse_sync_batchnorm(self)
        self._arena = ParamArena(self)
        self._param_list = list(self._arena.order)
        tensors = {n: p.data for n, p in self.named_parameters()}
        tensors.update({n: b for n, b in self.named_buffers()})
        self._engine = DetectorEngine(tensors, self._arena.grad_views, self.num_classes, self.num_anchors,
                                      self.planes, fcos=self._fcos, block=getattr(self.backbone, 'block', 'Bottleneck'),
                                      layers=getattr(self.backbone, 'layer_nums', RESNET50_LAYERS))
        # operands of the registered operator (custom_ops.py): what torch.compile's graph takes as inputs
        from ... import custom_ops
        custom_ops.register_model(self)
        self._co_params = [p for _, p in self._param_list]
        return self

    def _apply(self, fn, recurse=True):
        """.cuda() / .to(device): build the flat arenas at once, so that whatever wraps the model next
        (nn.parallel.DistributedDataParallel records parameter strides at construction, tools/utils.py:178-180)
        sees the parameters' final storage: k x k conv weights are channels_last views into the arena."""
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

    def forward(self, inputs):
        """Eager: one autograd node around the executor (_forward_eager).  Under torch.compile (config.use_compile,
        tools/train_detection_model.py:176-178) the same pass is ONE registered operator, sacv::detector_forward
        (custom_ops.py), so the tracer captures the model without a graph break."""
        if torch.compiler.is_compiling():
            outs = torch.ops.sacv.detector_forward(inputs.float(), self._co_params, self._co_handle, self.training,
                                                   torch.is_grad_enabled() and self.training)
            return [list(outs[k:k + 5]) for k in range(0, len(outs), 5)]
        return self._forward_eager(inputs)

    @torch.compiler.disable
    def _forward_eager(self, inputs):
        if inputs.device.type != 'cuda':
            raise RuntimeError(f'sacv_amd {type(self).__name__} runs on MI355X only: inputs must be on the GPU '
                               '(no CPU fallback)')
        self._ready()
        inputs = inputs.float()
        if torch.is_grad_enabled() and self.training:
            params = [p for _, p in self._param_list]
            outs = _DetectorFunction.apply(self, inputs, *params)
            self._last_outputs = outs
            n = 5
            return [list(outs[k:k + n]) for k in range(0, len(outs), n)]
        return [list(g) for g in self._engine.forward(inputs, training=self.training, keep=False)]
