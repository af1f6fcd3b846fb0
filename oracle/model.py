"""Oracle (test infrastructure): RetinaNet-/FCOS-ResNet50 forward on torch-CPU fp32.

A functional restatement over a flat {state_dict key: tensor} dictionary (same key
names as the reference modules) of
  ResNetBackbone.forward   simpleAICV/detection/models/backbones/resnet.py:96-113
  ConvBnActBlock/Bottleneck simpleAICV/classification/backbones/resnet.py:19-155
  RetinaFPN.forward        simpleAICV/detection/models/fpn.py:60-87
  RetinaClsHead/RegHead    simpleAICV/detection/models/head.py:46-52, 81-85
  FCOSClsRegCntHead        simpleAICV/detection/models/head.py:162-181
  RetinaNet.forward        simpleAICV/detection/models/retinanet.py:61-94
  FCOS.forward             simpleAICV/detection/models/fcos.py:56-89
Layers are the stock torch CPU ops (conv2d, batch_norm, max_pool2d, interpolate,
group_norm); autograd through them gives the reference gradients.
"""
import math
import zlib

import torch
import torch.nn.functional as F

from . import synth

RESNET50_LAYERS = [3, 4, 6, 3]
# detection/models/backbones/resnet.py:127-169: block type and stage depths of the five detection backbones
BACKBONES = {'resnet18backbone': ('BasicBlock', [2, 2, 2, 2]), 'resnet34backbone': ('BasicBlock', [3, 4, 6, 3]),
             'resnet50backbone': ('Bottleneck', [3, 4, 6, 3]), 'resnet101backbone': ('Bottleneck', [3, 4, 23, 3]),
             'resnet152backbone': ('Bottleneck', [3, 8, 36, 3])}


def init_retinanet_state(seed=0, num_classes=80, num_anchors=9, planes=256, fcos=False,
                         backbone='resnet50backbone'):
    """Random-init state dict with the reference key names, shapes and init laws.

    Backbone convs kaiming-normal fan_out (backbones/resnet.py:76-83); FPN convs the
    nn.Conv2d default law uniform(-b, b), b = 1/sqrt(fan_in); heads N(0, 0.01) with
    cls bias -log(99) (head.py:36-44).  Values come from the portable counter-based
    generator in oracle/synth.py (seeded per tensor name), NOT from the reference's
    torch RNG stream: parity tests load this one state dict into both sides.
    """
    sd = {}

    def randn(key, *shape):
        return torch.from_numpy(synth.normal(seed * 1000003 + zlib.crc32(key.encode()), shape))

    def rand(key, *shape):
        return torch.from_numpy(synth.uniform(seed * 1000003 + zlib.crc32(key.encode()) + 17, shape))

    def conv_bn(prefix, cin, cout, k):
        fan_out = cout * k * k
        sd[prefix + '.layer.0.weight'] = randn(prefix, cout, cin, k, k) * math.sqrt(2.0 / fan_out)
        sd[prefix + '.layer.1.weight'] = torch.ones(cout)
        sd[prefix + '.layer.1.bias'] = torch.zeros(cout)
        sd[prefix + '.layer.1.running_mean'] = torch.zeros(cout)
        sd[prefix + '.layer.1.running_var'] = torch.ones(cout)
        sd[prefix + '.layer.1.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)

    def conv_bias(prefix, cin, cout, k, std=None, bias_val=0.0, bias=True):
        if std is None:
            bound = 1.0 / math.sqrt(cin * k * k)
            sd[prefix + '.weight'] = (rand(prefix + '.w', cout, cin, k, k) * 2 - 1) * bound
            if bias:
                sd[prefix + '.bias'] = (rand(prefix + '.b', cout) * 2 - 1) * bound
        else:
            sd[prefix + '.weight'] = randn(prefix, cout, cin, k, k) * std
            if bias:
                sd[prefix + '.bias'] = torch.full((cout,), bias_val)

    conv_bn('backbone.conv1', 3, 64, 7)
    inpl = 64
    block, layers = BACKBONES[backbone]
    ex = 4 if block == 'Bottleneck' else 1
    for li, (n, pl) in enumerate(zip(layers, [64, 128, 256, 512]), start=1):
        for bi in range(n):
            p = f'backbone.layer{li}.{bi}'
            stride = 2 if (bi == 0 and li > 1) else 1
            if block == 'Bottleneck':  # classification/backbones/resnet.py:100-139
                conv_bn(p + '.conv1', inpl, pl, 1)
                conv_bn(p + '.conv2', pl, pl, 3)
                conv_bn(p + '.conv3', pl, pl * 4, 1)
            else:  # BasicBlock, :51-82
                conv_bn(p + '.conv1', inpl, pl, 3)
                conv_bn(p + '.conv2', pl, pl, 3)
            if stride != 1 or inpl != pl * ex:
                conv_bn(p + '.downsample_conv', inpl, pl * ex, 1)
            inpl = pl * ex
    conv_bias('fpn.P3_1', 128 * ex, planes, 1)
    conv_bias('fpn.P3_2', planes, planes, 3)
    conv_bias('fpn.P4_1', 256 * ex, planes, 1)
    conv_bias('fpn.P4_2', planes, planes, 3)
    conv_bias('fpn.P5_1', 512 * ex, planes, 1)
    conv_bias('fpn.P5_2', planes, planes, 3)
    conv_bias('fpn.P6', planes if fcos else 512 * ex, planes, 3)
    conv_bias('fpn.P7.1', planes, planes, 3)
    if not fcos:
        for i in (0, 2, 4, 6):
            conv_bias(f'cls_head.cls_head.{i}', planes, planes, 3, std=0.01)
            conv_bias(f'reg_head.reg_head.{i}', planes, planes, 3, std=0.01)
        conv_bias('cls_head.cls_out', planes, num_anchors * num_classes, 3, std=0.01,
                  bias_val=-math.log((1 - 0.01) / 0.01))
        conv_bias('reg_head.reg_out', planes, num_anchors * 4, 3, std=0.01)
    else:
        sd['scales'] = torch.tensor([1., 1., 1., 1., 1.], dtype=torch.float32)
        for tower in ('cls_head', 'reg_head'):
            for i in (0, 3, 6, 9):
                conv_bias(f'clsregcnt_head.{tower}.{i}', planes, planes, 3, std=0.01, bias=False)
                sd[f'clsregcnt_head.{tower}.{i + 1}.weight'] = torch.ones(planes)
                sd[f'clsregcnt_head.{tower}.{i + 1}.bias'] = torch.zeros(planes)
        conv_bias('clsregcnt_head.cls_out', planes, num_classes, 3, std=0.01,
                  bias_val=-math.log((1 - 0.01) / 0.01))
        conv_bias('clsregcnt_head.reg_out', planes, 4, 3, std=0.01)
        conv_bias('clsregcnt_head.center_out', planes, 1, 3, std=0.01)
    return sd


def _cba(sd, prefix, x, stride, pad, act, training, momentum=0.1, eps=1e-5):
    """ConvBnActBlock: conv (no bias) -> BatchNorm2d -> optional ReLU."""
    y = F.conv2d(x, sd[prefix + '.layer.0.weight'], None, stride, pad)
    rm, rv = sd[prefix + '.layer.1.running_mean'], sd[prefix + '.layer.1.running_var']
    y = F.batch_norm(y, rm, rv, sd[prefix + '.layer.1.weight'], sd[prefix + '.layer.1.bias'],
                     training, momentum, eps)
    if training:
        sd[prefix + '.layer.1.num_batches_tracked'] += 1
    return F.relu(y) if act else y


def backbone_forward(sd, x, training=True, prefix='backbone.'):
    x = _cba(sd, prefix + 'conv1', x, 2, 3, True, training)
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li in range(1, 5):  # the state dict says how deep a stage is and which block it is made of
        bi = 0
        while f'{prefix}layer{li}.{bi}.conv1.layer.0.weight' in sd:
            p = f'{prefix}layer{li}.{bi}'
            stride = 2 if (bi == 0 and li > 1) else 1
            idt = x
            if (p + '.conv3.layer.0.weight') in sd:  # Bottleneck
                y = _cba(sd, p + '.conv1', x, 1, 0, True, training)
                y = _cba(sd, p + '.conv2', y, stride, 1, True, training)
                y = _cba(sd, p + '.conv3', y, 1, 0, False, training)
            else:  # BasicBlock: 3x3 (stride) -> 3x3
                y = _cba(sd, p + '.conv1', x, stride, 1, True, training)
                y = _cba(sd, p + '.conv2', y, 1, 1, False, training)
            if (p + '.downsample_conv.layer.0.weight') in sd:
                idt = _cba(sd, p + '.downsample_conv', x, stride, 0, False, training)
            x = F.relu(y + idt)
            bi += 1
        feats.append(x)
    return feats  # [C2, C3, C4, C5]


def _conv(sd, prefix, x, stride=1, pad=1):
    return F.conv2d(x, sd[prefix + '.weight'], sd.get(prefix + '.bias'), stride, pad)


def fpn_forward(sd, c3, c4, c5, use_p5):
    p5 = _conv(sd, 'fpn.P5_1', c5, 1, 0)
    p4 = _conv(sd, 'fpn.P4_1', c4, 1, 0)
    p4 = F.interpolate(p5, size=p4.shape[2:], mode='bilinear', align_corners=True) + p4
    p3 = _conv(sd, 'fpn.P3_1', c3, 1, 0)
    p3 = F.interpolate(p4, size=p3.shape[2:], mode='bilinear', align_corners=True) + p3
    p5 = _conv(sd, 'fpn.P5_2', p5)
    p4 = _conv(sd, 'fpn.P4_2', p4)
    p3 = _conv(sd, 'fpn.P3_2', p3)
    p6 = _conv(sd, 'fpn.P6', p5 if use_p5 else c5, 2, 1)
    p7 = _conv(sd, 'fpn.P7.1', F.relu(p6), 2, 1)
    return [p3, p4, p5, p6, p7]


def retinanet_forward(sd, images, training=True, num_classes=80):
    """-> [cls_heads, reg_heads]; cls [B,H,W,A,C] probabilities, reg [B,H,W,A,4]."""
    _, c3, c4, c5 = backbone_forward(sd, images, training)
    feats = fpn_forward(sd, c3, c4, c5, use_p5=False)
    cls_heads, reg_heads = [], []
    for f in feats:
        x = f
        for i in (0, 2, 4, 6):
            x = F.relu(_conv(sd, f'cls_head.cls_head.{i}', x))
        x = torch.sigmoid(_conv(sd, 'cls_head.cls_out', x).float())
        x = x.permute(0, 2, 3, 1).contiguous()
        cls_heads.append(x.view(x.shape[0], x.shape[1], x.shape[2], -1, num_classes))
        y = f
        for i in (0, 2, 4, 6):
            y = F.relu(_conv(sd, f'reg_head.reg_head.{i}', y))
        y = _conv(sd, 'reg_head.reg_out', y).permute(0, 2, 3, 1).contiguous()
        reg_heads.append(y.view(y.shape[0], y.shape[1], y.shape[2], -1, 4))
    return [cls_heads, reg_heads]


def fcos_forward(sd, images, training=True):
    """-> [cls, reg, center]; cls [B,H,W,C] probs, reg [B,H,W,4] raw*exp(scale), center [B,H,W,1]."""
    _, c3, c4, c5 = backbone_forward(sd, images, training)
    feats = fpn_forward(sd, c3, c4, c5, use_p5=True)
    cls_heads, reg_heads, ctr_heads = [], [], []
    for lvl, f in enumerate(feats):
        towers = {}
        for tower in ('cls_head', 'reg_head'):
            x = f
            for i in (0, 3, 6, 9):
                x = _conv(sd, f'clsregcnt_head.{tower}.{i}', x)
                x = F.group_norm(x, 32, sd[f'clsregcnt_head.{tower}.{i + 1}.weight'],
                                 sd[f'clsregcnt_head.{tower}.{i + 1}.bias'], 1e-5)
                x = F.relu(x)
            towers[tower] = x
        c = torch.sigmoid(_conv(sd, 'clsregcnt_head.cls_out', towers['cls_head']).float())
        r = _conv(sd, 'clsregcnt_head.reg_out', towers['reg_head']).float()
        t = torch.sigmoid(_conv(sd, 'clsregcnt_head.center_out', towers['reg_head']).float())
        cls_heads.append(c.permute(0, 2, 3, 1).contiguous())
        reg_heads.append((r.permute(0, 2, 3, 1).contiguous()) * torch.exp(sd['scales'][lvl]))
        ctr_heads.append(t.permute(0, 2, 3, 1).contiguous())
    return [cls_heads, reg_heads, ctr_heads]


# ----------------------------------------------------------------------------- ResNet50 classifier (8f-1)
def init_resnet50_cls_state(seed=0, num_classes=1000):
    """State dict of classification/backbones/resnet.py::ResNet (Bottleneck [3,4,6,3]): the trunk of
    init_retinanet_state without the 'backbone.' prefix + fc (nn.Linear default law, bound 1/sqrt(2048))."""
    det = init_retinanet_state(seed)
    sd = {k[len('backbone.'):]: v for k, v in det.items() if k.startswith('backbone.')}
    bound = 1.0 / math.sqrt(2048)
    u = lambda key, *shape: torch.from_numpy(synth.uniform(seed * 1000003 + zlib.crc32(key.encode()) + 17, shape))
    sd['fc.weight'] = (u('fc.w', num_classes, 2048) * 2 - 1) * bound
    sd['fc.bias'] = (u('fc.b', num_classes) * 2 - 1) * bound
    return sd


def resnet50_cls_forward(sd, images, training=True):
    """ResNet.forward (classification/backbones/resnet.py:224-245): trunk -> global avg pool -> fc."""
    c5 = backbone_forward(sd, images, training, prefix='')[3]
    x = F.adaptive_avg_pool2d(c5, (1, 1)).flatten(1)
    return F.linear(x, sd['fc.weight'], sd['fc.bias'])


def ce_loss(logits, labels):
    """CELoss (classification/losses.py:14-26): nn.CrossEntropyLoss(reduction='mean')."""
    return F.cross_entropy(logits, labels, reduction='mean')


def sgd_momentum_step(params, grads, bufs, lr, momentum, weight_decay, first):
    """torch.optim.SGD as built by tools/utils.py:558-565 (weight decay on ndim>1 tensors only)."""
    for k in params:
        g = grads[k] + (weight_decay * params[k] if params[k].ndim > 1 else 0.0)
        bufs[k] = g.clone() if first else momentum * bufs[k] + g
        params[k] = params[k] - lr * bufs[k]


# ----------------------------------------------------------------------------- general ResNet classifiers
# (BASELINE.json config #1: resnet18cifar, CIFAR-100; classification/backbones/resnetforcifar.py:27-108,
#  BasicBlock classification/backbones/resnet.py:51-97)
def init_resnet_cls_state(seed=0, num_classes=100, block='BasicBlock', layers=(2, 2, 2, 2), stem='cifar'):
    """State dict of ResNet / ResNetCifar for any block type: kaiming-normal fan_out convs, BN (1, 0), nn.Linear
    default law for fc; values from the portable generator, seeded per tensor name."""
    sd = {}
    expansion = 4 if block == 'Bottleneck' else 1

    def conv_bn(prefix, cin, cout, k):
        w = torch.from_numpy(synth.normal(seed * 1000003 + zlib.crc32(prefix.encode()), (cout, cin, k, k)))
        sd[prefix + '.layer.0.weight'] = w * math.sqrt(2.0 / (cout * k * k))
        sd[prefix + '.layer.1.weight'] = torch.ones(cout)
        sd[prefix + '.layer.1.bias'] = torch.zeros(cout)
        sd[prefix + '.layer.1.running_mean'] = torch.zeros(cout)
        sd[prefix + '.layer.1.running_var'] = torch.ones(cout)
        sd[prefix + '.layer.1.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)

    conv_bn('conv1', 3, 64, 7 if stem == 'imagenet' else 3)
    inpl = 64
    for li, (n, pl) in enumerate(zip(layers, [64, 128, 256, 512]), start=1):
        for bi in range(n):
            p = f'layer{li}.{bi}'
            stride = 2 if (bi == 0 and li > 1) else 1
            if block == 'Bottleneck':
                conv_bn(p + '.conv1', inpl, pl, 1)
                conv_bn(p + '.conv2', pl, pl, 3)
                conv_bn(p + '.conv3', pl, pl * 4, 1)
            else:
                conv_bn(p + '.conv1', inpl, pl, 3)
                conv_bn(p + '.conv2', pl, pl, 3)
            if stride != 1 or inpl != pl * expansion:
                conv_bn(p + '.downsample_conv', inpl, pl * expansion, 1)
            inpl = pl * expansion
    bound = 1.0 / math.sqrt(inpl)
    u = lambda key, *shape: torch.from_numpy(synth.uniform(seed * 1000003 + zlib.crc32(key.encode()) + 17, shape))
    sd['fc.weight'] = (u('fc.w', num_classes, inpl) * 2 - 1) * bound
    sd['fc.bias'] = (u('fc.b', num_classes) * 2 - 1) * bound
    return sd


def resnet_cls_forward(sd, images, training=True, block='BasicBlock', layers=(2, 2, 2, 2), stem='cifar'):
    """ResNetCifar.forward (resnetforcifar.py:88-101) / ResNet.forward (resnet.py:224-245)."""
    if stem == 'imagenet':
        x = F.max_pool2d(_cba(sd, 'conv1', images, 2, 3, True, training), 3, 2, 1)
    else:
        x = _cba(sd, 'conv1', images, 1, 1, True, training)
    for li, n in enumerate(layers, start=1):
        for bi in range(n):
            p = f'layer{li}.{bi}'
            stride = 2 if (bi == 0 and li > 1) else 1
            idt = x
            if block == 'Bottleneck':
                y = _cba(sd, p + '.conv1', x, 1, 0, True, training)
                y = _cba(sd, p + '.conv2', y, stride, 1, True, training)
                y = _cba(sd, p + '.conv3', y, 1, 0, False, training)
            else:
                y = _cba(sd, p + '.conv1', x, stride, 1, True, training)
                y = _cba(sd, p + '.conv2', y, 1, 1, False, training)
            if (p + '.downsample_conv.layer.0.weight') in sd:
                idt = _cba(sd, p + '.downsample_conv', x, stride, 0, False, training)
            x = F.relu(y + idt)
    x = F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
    return F.linear(x, sd['fc.weight'], sd['fc.bias'])
