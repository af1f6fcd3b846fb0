"""Block-level GPU parity of the executor's composite forward/backward pieces (bottleneck with and
without projection shortcut and stride, FPN, shared head towers over 5 levels) against torch-CPU
autograd of the oracle's functional graph, on WELL-CONDITIONED random data: structural mistakes
(a missing gradient fan-in, a wrong mask, a wrong accumulation) show up as O(1) errors, fp32
rounding as ~1e-5.  Metric: relative L2 per tensor <= 1e-4 (the max-abs metric is not used because a
single ReLU whose pre-activation rounds across zero flips one element)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    from sacv_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True, params=['fp32_pipe', 'bf16x3_pipe'])
def matrix_pipe(request, ops):
    """Every block test runs twice: direct convolutions on the fp32 matrix pipe, and with the operands split into three
    bf16 terms (six products on the bf16 pipe, fp32 accumulators) -- same tolerances.  The engine methods called here
    do not go through forward() / backward(), which would set the plan from engine.conv_split."""
    prev = ops.launch_share(nt_split=int(request.param == 'bf16x3_pipe'))
    yield request.param
    ops.launch_share(*prev)


def rnd(seed, *shape, std=1.0):
    return torch.from_numpy(synth.normal(seed, shape, std))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def make_engine(sd):
    from sacv_amd.engine import DetectorEngine
    p = {k: v.clone().cuda() for k, v in sd.items()}
    g = {k: torch.full_like(v, float('nan')) for k, v in p.items() if v.dtype.is_floating_point}
    return DetectorEngine(p, g), p, g


def bn_entries(sd, prefix, c, seed):
    sd[prefix + '.layer.1.weight'] = 1.0 + 0.2 * rnd(seed, c)
    sd[prefix + '.layer.1.bias'] = 0.2 * rnd(seed + 1, c)
    sd[prefix + '.layer.1.running_mean'] = torch.zeros(c)
    sd[prefix + '.layer.1.running_var'] = torch.ones(c)
    sd[prefix + '.layer.1.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


@pytest.mark.parametrize('cin,planes,stride,h,w', [(256, 64, 1, 19, 23), (64, 64, 1, 20, 17), (256, 128, 2, 21, 26),
                                                   (512, 256, 2, 12, 15)])
def test_bottleneck_fwd_bwd(cin, planes, stride, h, w):
    pre = 'backbone.layer2.0'
    sd = {}
    for name, ci, co, k in (('conv1', cin, planes, 1), ('conv2', planes, planes, 3), ('conv3', planes, planes * 4, 1)):
        sd[f'{pre}.{name}.layer.0.weight'] = rnd(zlib.crc32(name.encode()) % 1000, co, ci, k, k, std=(2.0 / (ci * k * k))**0.5)
        bn_entries(sd, f'{pre}.{name}', co, 7 + len(name))
    if stride != 1 or cin != planes * 4:
        sd[f'{pre}.downsample_conv.layer.0.weight'] = rnd(5, planes * 4, cin, 1, 1, std=(2.0 / cin)**0.5)
        bn_entries(sd, f'{pre}.downsample_conv', planes * 4, 11)
    x = rnd(1, 2, cin, h, w)
    # oracle: the same functional code path the full oracle model uses
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    state = dict(sd)
    state = {k: v.clone() for k, v in state.items()}
    state.update(params)
    xr = x.clone().requires_grad_(True)
    y = omodel._cba(state, pre + '.conv1', xr, 1, 0, True, True)
    y = omodel._cba(state, pre + '.conv2', y, stride, 1, True, True)
    y = omodel._cba(state, pre + '.conv3', y, 1, 0, False, True)
    idt = omodel._cba(state, pre + '.downsample_conv', xr, stride, 0, False, True) if (pre + '.downsample_conv.layer.0.weight') in sd else xr
    out_r = F.relu(y + idt)
    dout = rnd(2, *out_r.shape)
    out_r.backward(dout)

    eng, p, g = make_engine(sd)
    out, (c1, c2, c3, cd) = eng._bottleneck_fwd(nhwc(x), pre, stride, True, True)
    assert rel_l2(nchw(out), out_r.detach()) <= 1e-5
    base = rnd(3, *x.shape)
    for accum in (None, nhwc(base)) if cd is not None else (None,):
        dx, _ = eng._bottleneck_bwd((c1, c2, c3, cd), nhwc(dout), accum)
        eng._side_join()  # weight gradients run on a side stream; engine.backward joins it, a bare block call does not
        ref = xr.grad if accum is None else xr.grad + base
        assert rel_l2(nchw(dx), ref) <= 1e-4, (accum is not None)
    for k, pr in params.items():
        assert rel_l2(g[k].cpu(), pr.grad) <= 1e-4, k


@pytest.mark.parametrize('fuse', [True, False])
def test_two_bottlenecks_bn_backward_partials_from_the_data_gradient(fuse):
    """A stage's first block (shortcut conv) followed by an identity block: with engine.fuse_bn_bwd the first-stage sums of
    bn2 / bn1 come from the epilogues of conv3's / conv2's data gradients (mask recomputed from y) and those of the FIRST
    block's bn3 from the identity block's conv1 data gradient (accumulating epilogue, mask from the stored z) -- against
    autograd through the oracle's functional blocks, and the fused run against the unfused one."""
    sd = {}
    cin, planes, h, w = 64, 64, 23, 37
    for bi, ci in ((0, cin), (1, planes * 4)):
        pre = f'backbone.layer1.{bi}'
        for name, a, b, k in (('conv1', ci, planes, 1), ('conv2', planes, planes, 3), ('conv3', planes, planes * 4, 1)):
            sd[f'{pre}.{name}.layer.0.weight'] = rnd(zlib.crc32((pre + name).encode()) % 1000, b, a, k, k,
                                                     std=(2.0 / (a * k * k))**0.5)
            bn_entries(sd, f'{pre}.{name}', b, 7 + len(name) + 10 * bi)
    sd['backbone.layer1.0.downsample_conv.layer.0.weight'] = rnd(5, planes * 4, cin, 1, 1, std=(2.0 / cin)**0.5)
    bn_entries(sd, 'backbone.layer1.0.downsample_conv', planes * 4, 11)
    x = rnd(1, 2, cin, h, w)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    state = {k: v.clone() for k, v in sd.items()}
    state.update(params)
    xr = x.clone().requires_grad_(True)
    cur = xr
    ref_relu = []  # the six ReLU outputs of the reference pass
    for bi in (0, 1):
        pre = f'backbone.layer1.{bi}'
        y = omodel._cba(state, pre + '.conv1', cur, 1, 0, True, True)
        ref_relu.append(y)
        y = omodel._cba(state, pre + '.conv2', y, 1, 1, True, True)
        ref_relu.append(y)
        y = omodel._cba(state, pre + '.conv3', y, 1, 0, False, True)
        idt = omodel._cba(state, pre + '.downsample_conv', cur, 1, 0, False, True) if bi == 0 else cur
        cur = F.relu(y + idt)
        ref_relu.append(cur)
    dout = rnd(2, *cur.shape)
    cur.backward(dout)

    eng, p, g = make_engine(sd)
    eng.fuse_bn_bwd = fuse
    z0, s0 = eng._bottleneck_fwd(nhwc(x), 'backbone.layer1.0', 1, True, True)
    z1, s1 = eng._bottleneck_fwd(z0, 'backbone.layer1.1', 1, True, True)
    assert rel_l2(nchw(z1), cur.detach()) <= 1e-5
    d0, st = eng._bottleneck_bwd(s1, nhwc(dout), None, fuse_prev=s0[2])
    assert (st is not None) == fuse
    dx, st0 = eng._bottleneck_bwd(s0, d0, None, partial=st)
    assert st0 is None
    eng._side_join()
    # Six ReLUs deep on 2 x 23 x 37 positions: a pre-activation within fp32 rounding of zero is passed by one
    # implementation and masked by the other -- a property of fp32 arithmetic (every kernel here has the fp32 pipe's error
    # against fp64, scripts/diag_conv_split.py), not of an implementation.  One flipped element moves its channel's mean
    # gradient by 1/1702 in the BatchNorm backward below it, i.e. EVERYTHING upstream by ~5e-4.  So: count the flips
    # against the reference pass; without flips the old gate holds, with (a few) flips the gate is what flips cost.
    gpu_relu = [s0[1].x, s0[2].x, z0, s1[1].x, s1[2].x, z1]  # conv2's / conv3's inputs and the block outputs
    flips = sum(int(((nchw(a) > 0) != (b.detach() > 0)).sum()) for a, b in zip(gpu_relu, ref_relu))
    assert flips <= 4, flips
    gate = 1e-4 if flips == 0 else 3e-3
    print(f'ReLU mask elements differing from the reference pass: {flips}; dx rel_l2 {rel_l2(nchw(dx), xr.grad):.2e}')
    assert rel_l2(nchw(dx), xr.grad) <= gate
    for k, pr in params.items():
        assert rel_l2(g[k].cpu(), pr.grad) <= gate, k


def test_fpn_fwd_bwd():
    sd = {k: v for k, v in omodel.init_retinanet_state(3).items() if k.startswith('fpn.')}
    sizes = [(13, 21), (7, 11), (4, 6)]
    cs = [rnd(10 + i, 2, c, h, w) for i, (c, (h, w)) in enumerate(zip((512, 1024, 2048), sizes))]
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    crs = [c.clone().requires_grad_(True) for c in cs]
    outs_r = omodel.fpn_forward(params, *crs, use_p5=False)
    douts = [rnd(20 + i, *o.shape) for i, o in enumerate(outs_r)]
    sum((o * d).sum() for o, d in zip(outs_r, douts)).backward()

    eng, p, g = make_engine(sd)
    ctx = {}
    outs = eng._fpn_fwd(*[nhwc(c) for c in cs], ctx)
    for o, r in zip(outs, outs_r):
        assert rel_l2(nchw(o), r.detach()) <= 1e-5
    eng.ctx = ctx
    dcs = eng._fpn_bwd([nhwc(d) for d in douts])
    eng._side_join()
    for d, cr in zip(dcs, crs):
        assert rel_l2(nchw(d), cr.grad) <= 1e-4
    for k, pr in params.items():
        assert rel_l2(g[k].cpu(), pr.grad) <= 1e-4, k


def test_heads_fwd_bwd_grouped_levels():
    full = omodel.init_retinanet_state(4)
    sd = {k: (v * 8 if k.endswith('weight') else v) for k, v in full.items() if 'head' in k}  # O(1) activations
    sizes = [(13, 21), (7, 11), (4, 6), (2, 3), (1, 2)]
    feats = [rnd(30 + i, 2, 256, h, w) for i, (h, w) in enumerate(sizes)]
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    frs = [f.clone().requires_grad_(True) for f in feats]
    cls_r, reg_r = [], []
    for f in frs:
        x = f
        for i in (0, 2, 4, 6):
            x = F.relu(omodel._conv(params, f'cls_head.cls_head.{i}', x))
        cls_r.append(torch.sigmoid(omodel._conv(params, 'cls_head.cls_out', x)))
        y = f
        for i in (0, 2, 4, 6):
            y = F.relu(omodel._conv(params, f'reg_head.reg_head.{i}', y))
        reg_r.append(omodel._conv(params, 'reg_head.reg_out', y))
    dcls = [rnd(40 + i, *o.shape) for i, o in enumerate(cls_r)]
    dreg = [rnd(50 + i, *o.shape) for i, o in enumerate(reg_r)]
    sum((o * d).sum() for o, d in zip(cls_r + reg_r, dcls + dreg)).backward()

    eng, p, g = make_engine(sd)
    ctx = {}
    cls, reg = eng._heads_fwd([nhwc(f) for f in feats], ctx)
    for o, r in zip(cls, cls_r):
        assert tuple(o.shape[3:]) == (9, 80)
        assert rel_l2(nchw(o.reshape(o.shape[0], o.shape[1], o.shape[2], -1)), r.detach()) <= 1e-5
    for o, r in zip(reg, reg_r):
        assert rel_l2(nchw(o.reshape(o.shape[0], o.shape[1], o.shape[2], -1)), r.detach()) <= 1e-5
    eng.ctx = ctx
    from sacv_amd import ops
    probs = ctx['cls_probs']
    dz_cls = [ops.eltwise(ops.EW_SCALE_SIGMOID_BWD, nhwc(d), pr, scalar=1.0) for d, pr in zip(dcls, probs)]
    dfeat = eng._tower_bwd(dz_cls, 'cls_head.cls_head', 'cls_head.cls_out')
    dfeat = eng._tower_bwd([nhwc(d) for d in dreg], 'reg_head.reg_head', 'reg_head.reg_out', accum_into=dfeat)
    eng._side_join()
    # five stacked ReLU layers with O(10) activations: a few masks flip on rounding -> 5e-4
    for d, fr in zip(dfeat, frs):
        assert rel_l2(nchw(d), fr.grad) <= 5e-4
    for k, pr in params.items():
        assert rel_l2(g[k].cpu(), pr.grad) <= 5e-4, k


def test_stem_and_pool_bwd():
    """conv1 7x7/s2 + BN + ReLU + maxpool backward chain as the executor runs it."""
    from sacv_amd import ops
    sd = {'backbone.conv1.layer.0.weight': rnd(1, 64, 3, 7, 7, std=0.1)}
    bn_entries(sd, 'backbone.conv1', 64, 3)
    x = torch.from_numpy(synth.images(2, 2, 61, 75))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    state = {k: v.clone() for k, v in sd.items()}
    state.update(params)
    z = omodel._cba(state, 'backbone.conv1', x, 2, 3, True, True)
    out_r = F.max_pool2d(z, 3, 2, 1)
    dout = rnd(5, *out_r.shape)
    out_r.backward(dout)
    eng, p, g = make_engine(sd)
    ho, wo = z.shape[2:]
    col = ops.im2col_stem(x.cuda(), ho, wo)
    y = ops.conv_gemm([col], ops.pack_weight(p['backbone.conv1.layer.0.weight'], ops.PACK_STEM), 64, 1, 1, 1, 0, 0, [(ho, wo)])[0]
    zz, c = eng._bn_fwd(col, y, 'backbone.conv1', 7, 2, 3, True, True, None, True)
    pooled, idx = ops.maxpool_fwd(zz)
    assert rel_l2(nchw(pooled), out_r.detach()) <= 1e-5
    dpool = ops.maxpool_bwd(nhwc(dout), idx, tuple(zz.shape))
    dy, _ = ops.bn_train_bwd(dpool, None, c.y, c.mean, c.invstd, p['backbone.conv1.layer.1.weight'],
                             g['backbone.conv1.layer.1.weight'], g['backbone.conv1.layer.1.bias'],
                             relu_beta=p['backbone.conv1.layer.1.bias'])  # as engine._backbone_bwd does
    ops.conv_wgrad([dy], [c.x], g['backbone.conv1.layer.0.weight'], 3, 1, 1, 1, 0, cin_pad=ops.STEM_KPAD, dw_taps=49)
    for k, pr in params.items():
        assert rel_l2(g[k].cpu(), pr.grad) <= 1e-4, k


@pytest.mark.parametrize('kind', ['retina', 'fcos', 'retina18'])
def test_eval_forward_with_folded_batchnorm(kind):
    """Evaluation passes apply BatchNorm (running statistics) as a scale / shift in the convolution epilogue and
    accumulate the last conv of a residual block into the shortcut's buffer (engine.fold_bn): against the same pass with
    conv + bn_apply, and against the CPU oracle in eval mode.  Bottleneck and BasicBlock trunks; running statistics
    away from (0, 1) so that the fold has something to do."""
    from sacv_amd.detection import models
    fcos = kind == 'fcos'
    bb = 'resnet18backbone' if kind == 'retina18' else 'resnet50backbone'
    sd = omodel.init_retinanet_state(0, fcos=fcos, backbone=bb)
    for k in list(sd):
        if k.endswith('running_mean'):
            sd[k] = torch.from_numpy(synth.normal(len(k), tuple(sd[k].shape), std=0.2))
        elif k.endswith('running_var'):
            sd[k] = torch.from_numpy(synth.uniform(len(k) + 1, tuple(sd[k].shape), 0.5, 2.0))
        elif k.endswith('layer.1.weight'):
            sd[k] = torch.from_numpy(synth.uniform(len(k) + 2, tuple(sd[k].shape), 0.5, 1.5))
        elif k.endswith('layer.1.bias'):
            sd[k] = torch.from_numpy(synth.normal(len(k) + 3, tuple(sd[k].shape), std=0.1))
    name = bb.replace('backbone', '') + ('_fcos' if fcos else '_retinanet')
    net = models.__dict__[name]()
    net.load_state_dict(sd)
    net = net.cuda().eval()
    x = torch.tensor(synth.images(77, 2, 224, 288))
    outs = {}
    with torch.no_grad():
        for fold in (True, False):
            net._engine.fold_bn = fold
            outs[fold] = [[o.clone() for o in grp] for grp in net(x.cuda())]
        ref = (omodel.fcos_forward if fcos else omodel.retinanet_forward)({k: v.clone() for k, v in sd.items()}, x,
                                                                          training=False)
    for ga, gb, gr in zip(outs[True], outs[False], ref):
        for a, b, r in zip(ga, gb, gr):
            scale = max(1.0, float(r.abs().max()))
            assert float((a - b).abs().max()) <= 2e-5 * scale
            assert float((a.cpu() - r).abs().max()) <= 2e-4 * scale
