"""GPU parity of every HIP kernel, called through the C ABI, against the CPU oracle.

Integer / index results (anchors, labels, matched boxes, detections) are compared bit-exact;
fp32 layer kernels against a plain torch-CPU fp32 reference of the same op with the tolerance
written at each assert.  Run on the MI355X box with `pytest -m gpu`.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import anchors as oanchors, assign as oassign, decode as odecode, losses as olosses, synth
from tests import golden_inputs as gi


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    from sacv_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True, params=['fp32_pipe', 'bf16x3_pipe'])
def matrix_pipe(request, ops):
    """Every test of this module runs twice: sacv_conv2d_gemm on the fp32 matrix pipe, and with the operands split into
    three bf16 terms at the LDS store (six products on the bf16 pipe, fp32 accumulators; csrc/conv_gemm.hip SP) -- the
    same tolerances: the split form carries the fp32 pipe's error."""
    prev = ops.launch_share(nt_split=int(request.param == 'bf16x3_pipe'))
    yield request.param
    ops.launch_share(*prev)


def nhwc(t):  # NCHW cpu tensor -> NHWC cuda tensor
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):  # NHWC cuda tensor -> NCHW cpu tensor
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(seed, *shape, std=1.0):
    return torch.from_numpy(synth.normal(seed, shape, std))


# ----------------------------------------------------------------------------------------- anchors / assign
@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1', 'S2'])
def test_anchors_bit_exact(ops, tag):
    fs = synth.feature_sizes(*gi.SHAPES[tag])
    got = ops.retina_anchors(fs, oanchors.DEFAULT_AREAS, oanchors.DEFAULT_RATIOS, oanchors.DEFAULT_SCALES,
                             oanchors.DEFAULT_STRIDES, 'cuda').cpu().numpy()
    ref = oanchors.retina_anchors(fs)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    pts = ops.fcos_points(fs, oanchors.DEFAULT_STRIDES, 'cuda').cpu().numpy()
    assert np.array_equal(pts, oanchors.fcos_positions(fs))


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
def test_retina_assign_bit_exact_labels(ops, tag, golden_dir):
    H, W = gi.SHAPES[tag]
    meta = json.load(open(os.path.join(golden_dir, 'assign.json')))
    anc = oanchors.retina_anchors(synth.feature_sizes(H, W))
    anc_d = torch.from_numpy(anc).cuda()
    cases = gi.annots_cases(H, W)
    cases['none'] = synth.annotations(1, 2, H, W, boxes_per_image=0)
    for cname, ann in cases.items():
        for blt in ('SmoothL1', 'GIoU'):
            tg, counts = ops.retina_assign(anc_d, torch.from_numpy(ann).cuda(), smooth_l1=blt == 'SmoothL1')
            tg = tg.cpu().numpy()
            ref = oassign.retina_assign(anc, ann, blt)
            assert np.array_equal(tg[..., 4], ref[..., 4]), (tag, cname, blt)  # labels: bit-exact
            pos = ref[..., 4] > 0
            # matched GT / box targets: exact for xyxy mode, 1e-6 for log/div targets
            if blt == 'GIoU':
                assert np.array_equal(tg[..., :4], ref[..., :4])
            else:
                np.testing.assert_allclose(tg[..., :4], ref[..., :4], rtol=1e-6, atol=1e-6)
            c = counts.cpu().numpy()
            assert c[0] == pos.sum() and c[1] == (ref[..., 4] >= 0).sum()
            key = f'retina_{tag}_{cname}_{blt}'
            if key in meta:  # and against the reference-generated fixture
                import hashlib
                assert hashlib.sha256(tg[..., 4].astype(np.int8).tobytes()).hexdigest() == meta[key]['label_sha256']


# ----------------------------------------------------------------------------------------- losses
def _oracle_loss(crit, preds_np, ann):
    preds = [[torch.tensor(p, requires_grad=True) for p in lv] for lv in preds_np]
    ld = crit(preds, torch.tensor(ann))
    total = sum(ld.values())
    grads = None
    if total.requires_grad:
        total.backward()
        grads = [[(p.grad if p.grad is not None else torch.zeros_like(p)) for p in lv] for lv in preds]
    return {k: float(v) for k, v in ld.items()}, grads


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
@pytest.mark.parametrize('blt', ['SmoothL1', 'GIoU', 'IoU', 'DIoU', 'CIoU', 'EIoU'])
def test_retina_loss_and_grads(ops, tag, blt):
    from sacv_amd.detection import losses
    H, W = gi.SHAPES[tag]
    cases = gi.annots_cases(H, W)
    cases['none'] = synth.annotations(1, 2, H, W, boxes_per_image=0)
    if tag == 'S1':
        cases = {k: cases[k] for k in ('g8', 'one_empty')}
    cls, reg = gi.synth_preds_retina(11, 2, H, W)
    crit = losses.__dict__['RetinaLoss'](box_loss_type=blt)
    ocrit = olosses.RetinaLossOracle(box_loss_type=blt)
    for cname, ann in cases.items():
        ref, rgrads = _oracle_loss(ocrit, [cls, reg], ann)
        cp = [torch.tensor(c).cuda().requires_grad_(True) for c in cls]
        rp = [torch.tensor(r).cuda().requires_grad_(True) for r in reg]
        ld = crit([cp, rp], torch.tensor(ann).cuda())
        assert set(ld) == {'cls_loss', 'reg_loss'}
        for k in ld:
            # fp32 loss tolerance from BASELINE.json: 1e-4
            assert abs(float(ld[k]) - ref[k]) <= 1e-4 * max(1.0, abs(ref[k])), (tag, cname, blt, k)
        total = sum(ld.values())
        if rgrads is None:
            assert not total.requires_grad  # zero-positive branch: plain zeros (losses.py:234-235)
            continue
        total.backward()
        for got, want in zip(cp + rp, rgrads[0] + rgrads[1]):
            g = got.grad.cpu()
            scale = float(want.abs().max()) + 1e-12
            assert float((g - want).abs().max()) <= 2e-5 * scale + 1e-9, (tag, cname, blt)


# ----------------------------------------------------------------------------------------- convolution
CONV_CASES = [
    # (N, Cin, H, W, Cout, k, stride, pad)
    (2, 64, 25, 42, 64, 3, 1, 1),
    (2, 64, 19, 23, 256, 1, 1, 0),
    (2, 256, 13, 21, 64, 1, 1, 0),
    (2, 128, 26, 31, 128, 3, 2, 1),
    (2, 256, 25, 42, 512, 1, 2, 0),
    (2, 256, 13, 21, 256, 3, 1, 1),
    (1, 256, 7, 11, 720, 3, 1, 1),
    (2, 256, 7, 11, 36, 3, 1, 1),
    (2, 512, 9, 10, 256, 3, 2, 1),
    (2, 32, 8, 9, 96, 3, 1, 1),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(ops, case):
    n, cin, h, w, cout, k, s, p = case
    x = rnd(1, n, cin, h, w)
    wt = rnd(2, cout, cin, k, k, std=(1.0 / (cin * k * k))**0.5)
    b = rnd(3, cout)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, s, p)
    dy = rnd(4, *yr.shape)
    yr.backward(dy)
    ho, wo = yr.shape[2:]
    xd, wd_ = nhwc(x), wt.cuda()
    y = ops.conv_gemm([xd], ops.pack_weight(wd_, ops.PACK_FWD), cout, k, k, s, p, 0, [(ho, wo)], bias=b.cuda())[0]
    # fp32 MFMA = exact fp32 fma chain; only the summation order differs from oneDNN: 2e-5 relative
    tol = 2e-5 * float(yr.abs().max())
    assert float((nchw(y) - yr.detach()).abs().max()) <= tol
    dyd = nhwc(dy)
    dx = ops.conv_gemm([dyd], ops.pack_weight(wd_, ops.PACK_DGRAD), cin, k, k, s, p, 1, [(h, w)])[0] \
        if not (k == 1 and s > 1) else None
    if dx is None:
        dx = torch.zeros((n, h, w, cin), device='cuda')
        ops.conv_gemm([dyd], ops.pack_weight(wd_, ops.PACK_DGRAD), cin, 1, 1, 1, 0, 0, [(ho, wo)],
                      flags=ops.CF_ACCUM, outs=[dx], remap=(s, h, w))
    assert float((nchw(dx) - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max())
    dw = torch.empty_like(wd_)
    ops.conv_wgrad([dyd], [xd], dw, cin, k, k, s, p)
    assert float((dw.cpu() - wr.grad).abs().max()) <= 5e-5 * float(wr.grad.abs().max())


def test_conv_split_plan_stays_inside_workspace(ops):
    """Regression: 392 output tiles on 512 persistent workgroups (layer4.0.conv2 of the bs-256 classifier).  The
    split-K planner once preferred 5 K-slices per tile = 1960 partial slabs for a 512-slab workspace."""
    x = rnd(1, 256, 512, 14, 14)
    wt = rnd(2, 512, 512, 3, 3, std=(1.0 / (512 * 9))**0.5)
    yr = F.conv2d(x, wt, None, 2, 1)
    guard = torch.zeros(1 << 20, device='cuda')  # likely neighbour of a freshly grown workspace
    y = ops.conv_gemm([nhwc(x)], ops.pack_weight(wt.cuda(), ops.PACK_FWD), 512, 3, 3, 2, 1, 0, [(7, 7)])[0]
    assert float((nchw(y) - yr).abs().max()) <= 2e-5 * float(yr.abs().max())
    assert float(guard.abs().max()) == 0.0


def test_channels_last_weight_paths(ops):
    """A channels_last-stored weight is its own forward panel (zero copy); its data-gradient panel comes from the
    batched one-launch pack; the weight gradient is written in the same channels_last order."""
    wt = rnd(2, 96, 32, 3, 3)
    w_cl = wt.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)  # logical OIHW, stored OHWI
    w_pl = wt.cuda()
    assert ops.weight_layout(w_cl) == 1 and ops.weight_layout(w_pl) == 0
    pf = ops.fwd_panel(w_cl)
    assert pf.data_ptr() == w_cl.data_ptr()
    assert torch.equal(pf.reshape(96, -1), ops.pack_weight(w_pl, ops.PACK_FWD))
    ref_d = ops.pack_weight(w_pl, ops.PACK_DGRAD)
    assert torch.equal(ops.pack_weight(w_cl, ops.PACK_DGRAD), ref_d)
    w2 = rnd(3, 40, 64, 1, 1).cuda()
    panels = ops.DgradPanels([w_cl, w2, w_pl])
    panels.refresh()
    assert torch.equal(panels.get(w_cl), ref_d) and torch.equal(panels.get(w_pl), ref_d)
    assert torch.equal(panels.get(w2), ops.pack_weight(w2, ops.PACK_DGRAD))
    x, dy = nhwc(rnd(4, 2, 32, 9, 11)), nhwc(rnd(5, 2, 96, 9, 11))
    dw_pl = torch.empty_like(w_pl)
    dw_cl = torch.empty_like(w_cl)
    assert dw_cl.stride() == w_cl.stride()
    ops.conv_wgrad([dy], [x], dw_pl, 32, 3, 3, 1, 1)
    ops.conv_wgrad([dy], [x], dw_cl, 32, 3, 3, 1, 1)
    assert torch.equal(dw_cl, dw_pl)


def test_conv_epilogues_and_grouping(ops):
    """relu / sigmoid / mask / accumulate epilogues and a 5-problem grouped launch (shared weights)."""
    cin, cout, k = 64, 96, 3
    wt = rnd(5, cout, cin, k, k, std=0.05)
    b = rnd(6, cout, std=0.1)
    sizes = [(13, 21), (7, 11), (4, 6), (2, 3), (1, 2)]
    xs = [rnd(10 + i, 2, cin, h, w) for i, (h, w) in enumerate(sizes)]
    wp = ops.pack_weight(wt.cuda(), ops.PACK_FWD)
    ys = ops.conv_gemm([nhwc(x) for x in xs], wp, cout, k, k, 1, 1, 0, sizes, bias=b.cuda(), flags=ops.CF_RELU)
    for x, y in zip(xs, ys):
        ref = F.relu(F.conv2d(x, wt, b, 1, 1))
        assert float((nchw(y) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    ys = ops.conv_gemm([nhwc(x) for x in xs], wp, cout, k, k, 1, 1, 0, sizes, bias=b.cuda(), flags=ops.CF_SIGMOID)
    for x, y in zip(xs, ys):
        ref = torch.sigmoid(F.conv2d(x, wt, b, 1, 1))
        assert float((nchw(y) - ref).abs().max()) <= 2e-6
    # grouped wgrad (sum over problems) and masked / accumulated dgrad
    dys = [rnd(30 + i, 2, cout, h, w) for i, (h, w) in enumerate(sizes)]
    wr = wt.clone().requires_grad_(True)
    xrs = [x.clone().requires_grad_(True) for x in xs]
    tot = sum((F.conv2d(x, wr, None, 1, 1) * dy).sum() for x, dy in zip(xrs, dys))
    tot.backward()
    dw = torch.empty_like(wt).cuda()
    ops.conv_wgrad([nhwc(d) for d in dys], [nhwc(x) for x in xs], dw, cin, k, k, 1, 1)
    assert float((dw.cpu() - wr.grad).abs().max()) <= 5e-5 * float(wr.grad.abs().max())
    wdp = ops.pack_weight(wt.cuda(), ops.PACK_DGRAD)
    masks = [rnd(50 + i, 2, cin, h, w) for i, (h, w) in enumerate(sizes)]
    dxs = ops.conv_gemm([nhwc(d) for d in dys], wdp, cin, k, k, 1, 1, 1, sizes, flags=ops.CF_MASK,
                        masks=[nhwc(m) for m in masks])
    for xr, m, dx in zip(xrs, masks, dxs):
        ref = xr.grad * (m > 0)
        assert float((nchw(dx) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    base = [rnd(70 + i, 2, cin, h, w) for i, (h, w) in enumerate(sizes)]
    outs = [nhwc(t) for t in base]
    ops.conv_gemm([nhwc(d) for d in dys], wdp, cin, k, k, 1, 1, 1, sizes, flags=ops.CF_ACCUM, outs=outs)
    for xr, t, o in zip(xrs, base, outs):
        ref = xr.grad + t
        assert float((nchw(o) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize('n,cin,h,w,cout,k', [(2, 64, 37, 41, 256, 1), (2, 128, 19, 23, 128, 3), (2, 256, 50, 84, 64, 1)])
def test_conv_stats_epilogue(ops, n, cin, h, w, cout, k):
    """CF_STATS: per-M-tile column sums / sums of squares of the conv output (BatchNorm statistics and
    bias gradients come from here), for whole tiles and for K-split tiles (fix-up kernel)."""
    x = rnd(1, n, cin, h, w)
    wt = rnd(2, cout, cin, k, k, std=(1.0 / (cin * k * k))**0.5)
    ref = F.conv2d(x, wt, None, 1, k // 2)
    (y,), st = ops.conv_gemm([nhwc(x)], ops.pack_weight(wt.cuda(), ops.PACK_FWD), cout, k, k, 1, k // 2, 0,
                             [(h, w)], stats=True)
    assert float((nchw(y) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    rows = n * h * w
    assert st.shape == (-(-rows // 128), 2, cout)
    s1 = st[:, 0].double().sum(0).cpu()
    s2 = st[:, 1].double().sum(0).cpu()
    r1 = ref.double().sum((0, 2, 3))
    r2 = (ref.double()**2).sum((0, 2, 3))
    np.testing.assert_allclose(s1.numpy(), r1.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s2.numpy(), r2.numpy(), rtol=1e-4, atol=1e-3)
    rm, rv, nbt = torch.zeros(cout).cuda(), torch.ones(cout).cuda(), torch.zeros((), dtype=torch.long).cuda()
    mean, invstd = ops.bn_finalize_stats(st, rows, 1e-5, 0.1, rm, rv, nbt)
    np.testing.assert_allclose(mean.cpu().numpy(), ref.mean((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(invstd.cpu().numpy(), (1.0 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)).numpy(),
                               rtol=1e-4)
    out = torch.zeros(cout).cuda()
    ops.colsum_finalize(st, out)
    np.testing.assert_allclose(out.cpu().numpy(), r1.float().numpy(), rtol=1e-4, atol=1e-3)


def test_stem_conv(ops):
    """7x7/s2 stem through explicit im2col, for NCHW-contiguous and channels_last images."""
    x = torch.from_numpy(synth.images(3, 2, 70, 93))
    wt = rnd(7, 64, 3, 7, 7, std=0.05)
    xr, wr = x.clone(), wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 2, 3)
    dy = rnd(8, *yr.shape)
    yr.backward(dy)
    ho, wo = yr.shape[2:]
    wp = ops.pack_weight(wt.cuda(), ops.PACK_STEM)
    for img in (x.cuda(), x.cuda().contiguous(memory_format=torch.channels_last)):
        col = ops.im2col_stem(img, ho, wo)
        y = ops.conv_gemm([col], wp, 64, 1, 1, 1, 0, 0, [(ho, wo)])[0]
        assert float((nchw(y) - yr.detach()).abs().max()) <= 2e-5 * float(yr.abs().max())
    dw = torch.empty_like(wt).cuda()
    ops.conv_wgrad([nhwc(dy)], [col], dw, 3, 1, 1, 1, 0, cin_pad=ops.STEM_KPAD, dw_taps=49)
    assert float((dw.cpu() - wr.grad).abs().max()) <= 5e-5 * float(wr.grad.abs().max())


@pytest.mark.parametrize('hw', [(70, 93), (64, 128), (33, 31), (800, 1333)])
def test_stem_conv_without_im2col(ops, hw):
    """sacv_stem_conv_fwd (the 7x7/s2 stem reading the image itself) vs torch's conv2d, for NCHW-contiguous and
    channels_last images, odd sizes (partial tiles in both directions, output narrower than one tile), and against
    the im2col + GEMM path it replaces; the per-tile BatchNorm partial sums add up to the column sums of y."""
    H, W = hw
    x = torch.from_numpy(synth.images(3, 2, H, W))
    wt = rnd(7, 64, 3, 7, 7, std=0.05)
    yr = F.conv2d(x, wt, None, 2, 3)
    ho, wo = yr.shape[2:]
    wp = ops.pack_weight(wt.cuda(), ops.PACK_STEM)
    scale = float(yr.abs().max())
    for img in (x.cuda(), x.cuda().contiguous(memory_format=torch.channels_last)):
        y, st = ops.stem_conv(img, wp, ho, wo, stats=True)
        assert float((nchw(y) - yr).abs().max()) <= 2e-5 * scale
        y2 = ops.stem_conv(img, wp, ho, wo)
        assert torch.equal(y, y2)
        tot = st.double().sum(dim=0).cpu()
        ref = torch.stack([yr.double().sum(dim=(0, 2, 3)), (yr.double() ** 2).sum(dim=(0, 2, 3))])
        assert float((tot - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    col = ops.im2col_stem(x.cuda(), ho, wo)
    y3 = ops.conv_gemm([col], wp, 64, 1, 1, 1, 0, 0, [(ho, wo)])[0]
    assert float((y3 - y).abs().max()) <= 1e-5 * scale  # (not bit-identical: the GEMM path may split K over workgroups)


@pytest.mark.parametrize('hw', [(70, 93), (64, 128), (33, 31), (9, 200), (800, 1333)])
def test_stem_wgrad_without_im2col(ops, hw):
    """sacv_stem_conv_wgrad (the stem's weight gradient from the image and dy) vs torch's conv2d backward in fp64, for
    NCHW-contiguous and channels_last images, odd sizes (partial tiles in both directions), accumulation into dw, run to
    run bit stability, and against the im2col + conv_tn path it replaces."""
    H, W = hw
    x = torch.from_numpy(synth.images(3, 2, H, W))
    wt = rnd(7, 64, 3, 7, 7, std=0.05)
    wr = wt.double().requires_grad_(True)
    yr = F.conv2d(x.double(), wr, None, 2, 3)
    dy = rnd(8, *yr.shape)
    yr.backward(dy.double())
    ho, wo = yr.shape[2:]
    want = wr.grad
    scale = float(want.abs().max())
    dyd = nhwc(dy)
    first = None
    for img in (x.cuda(), x.cuda().contiguous(memory_format=torch.channels_last)):
        dw = torch.full((64, 3, 7, 7), float('nan'), device='cuda')
        ops.stem_conv_wgrad(img, dyd, dw)
        assert float((dw.cpu().double() - want).abs().max()) <= 2e-5 * scale
        first = dw.clone() if first is None else first
        assert torch.equal(dw, first)  # same bits for both image layouts and from run to run
    acc = rnd(9, 64, 3, 7, 7).cuda()
    got = acc.clone()
    ops.stem_conv_wgrad(x.cuda(), dyd, got, accumulate=True)
    assert float((got.cpu().double() - (acc.cpu().double() + want)).abs().max()) <= 2e-5 * scale + 1e-6
    col = ops.im2col_stem(x.cuda(), ho, wo)
    dw2 = torch.empty((64, 3, 7, 7), device='cuda')
    ops.conv_wgrad([dyd], [col], dw2, 3, 1, 1, 1, 0, cin_pad=ops.STEM_KPAD, dw_taps=49)
    assert float((dw2 - first).abs().max()) <= 2e-5 * scale


# ----------------------------------------------------------------------------------------- norm / pointwise
@pytest.mark.parametrize('shape', [(2, 64, 50, 67), (2, 256, 13, 21), (2, 2048, 4, 6), (2, 1024, 7, 11)])
@pytest.mark.parametrize('relu,res', [(True, False), (True, True), (False, False)])
def test_batchnorm_train_fwd_bwd(ops, shape, relu, res):
    n, c, h, w = shape
    y = rnd(1, *shape) * 2.0 + 0.5
    gamma, beta = rnd(2, c) * 0.1 + 1.0, rnd(3, c) * 0.1
    r = rnd(4, *shape) if res else None
    rm, rv = torch.zeros(c), torch.ones(c)
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    z = F.batch_norm(yr, rm, rv, gr, br, True, 0.1, 1e-5)
    if res:
        z = z + rr
    if relu:
        z = F.relu(z)
    dz = rnd(5, *shape)
    z.backward(dz)
    yd = nhwc(y)
    rmd, rvd, nbt = torch.zeros(c).cuda(), torch.ones(c).cuda(), torch.zeros((), dtype=torch.long).cuda()
    mean, invstd = ops.bn_train_stats(yd, 1e-5, 0.1, rmd, rvd, nbt)
    zd = ops.bn_apply(yd, mean, invstd, gamma.cuda(), beta.cuda(), nhwc(r) if res else None, relu)
    assert float((nchw(zd) - z.detach()).abs().max()) <= 1e-5 * max(1.0, float(z.abs().max()))
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-5, atol=1e-6)
    assert int(nbt) == 1
    dg, db = torch.empty(c).cuda(), torch.empty(c).cuda()
    dyd, g = ops.bn_train_bwd(nhwc(dz), zd if relu else None, yd, mean, invstd, gamma.cuda(), dg, db, want_g=True)
    assert float((nchw(dyd) - yr.grad).abs().max()) <= 2e-5 * float(yr.grad.abs().max())
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), rtol=2e-4, atol=2e-4)
    if res:
        assert float((nchw(g) - rr.grad).abs().max()) <= 1e-6
    if relu and not res:
        # the ReLU mask recomputed from y (no read of z) must reproduce the stored-z result bit for bit
        dg2, db2 = torch.empty(c).cuda(), torch.empty(c).cuda()
        dy2, g2 = ops.bn_train_bwd(nhwc(dz), None, yd, mean, invstd, gamma.cuda(), dg2, db2, want_g=True,
                                   relu_beta=beta.cuda())
        assert torch.equal(dy2, dyd) and torch.equal(g2, g) and torch.equal(dg2, dg) and torch.equal(db2, db)


@pytest.mark.parametrize('n,cin,h,w,cout,k,zmask', [(2, 256, 50, 67, 64, 1, False), (2, 64, 37, 41, 256, 1, True),
                                                     (2, 128, 26, 31, 128, 3, False), (2, 512, 13, 21, 128, 1, False)])
def test_bn_backward_first_stage_in_the_data_gradient_epilogue(ops, n, cin, h, w, cout, k, zmask):
    """sacv_conv2d_gemm_bnbwd + sacv_bn_train_bwd_partial against the unfused pair (data gradient, then the complete
    sacv_bn_train_bwd): dz bit-identical (same kernel, one more epilogue option), dgamma / dbeta / dy to fp32 summation
    noise (the first-stage sums are taken per 128-row tile instead of per row chunk; both are finished in fp64).  `cout`
    is the conv's output = the data gradient's source; its result dz [.., cin] is the upstream gradient of a BatchNorm
    layer over cin channels.  zmask: that layer's ReLU follows a residual add (mask from z), the data gradient accumulates
    into an existing tensor (the shortcut gradient)."""
    p = k // 2
    wt = rnd(2, cout, cin, k, k, std=(1.0 / (cin * k * k))**0.5)
    wd = ops.pack_weight(wt.cuda(), ops.PACK_DGRAD)
    dy = nhwc(rnd(3, n, cout, h, w))
    y = nhwc(rnd(4, n, cin, h, w))              # the BatchNorm layer's input (its conv's output)
    gamma = (1.0 + 0.2 * rnd(5, cin)).cuda()
    beta = (0.2 * rnd(6, cin)).cuda()
    mean, invstd = ops.bn_train_stats(y, 1e-5, 0.1, None, None, None)
    res = nhwc(rnd(7, n, cin, h, w)) if zmask else None
    z = ops.bn_apply(y, mean, invstd, gamma, beta, res, True)
    base = nhwc(rnd(8, n, cin, h, w)) if zmask else None
    flags = ops.CF_ACCUM if zmask else 0

    def grads():
        return torch.zeros(cin, device='cuda'), torch.zeros(cin, device='cuda')
    # unfused
    out0 = base.clone() if zmask else None
    dz0 = ops.conv_gemm([dy], wd, cin, k, k, 1, p, 1, [(h, w)], flags=flags, outs=[out0] if zmask else None)[0]
    dg0, db0 = grads()
    dx0, g0 = ops.bn_train_bwd(dz0, z if zmask else None, y, mean, invstd, gamma, dg0, db0, want_g=True,
                               relu_beta=None if zmask else beta)
    # fused
    out1 = base.clone() if zmask else None
    dz1, st = ops.conv_gemm_bnbwd(dy, wd, cin, k, k, p, (h, w), y, mean, invstd, bn_gamma=None if zmask else gamma,
                                  bn_beta=None if zmask else beta, bn_z=z if zmask else None, out=out1, flags=flags)
    assert torch.equal(dz1, dz0)
    dg1, db1 = grads()
    dx1, g1 = ops.bn_train_bwd(dz1, z if zmask else None, y, mean, invstd, gamma, dg1, db1, want_g=True,
                               relu_beta=None if zmask else beta, partial=st)
    assert torch.equal(g1, g0)  # the masked upstream gradient comes from the apply pass in both
    # first-stage sums against fp64
    gm = g0.reshape(-1, cin).double()
    xh = ((y.reshape(-1, cin).double() - mean.double()) * invstd.double())
    assert torch.allclose(st[:, 0].double().sum(0), gm.sum(0), rtol=1e-5, atol=1e-4 * float(gm.abs().sum(0).max()))
    assert torch.allclose(st[:, 1].double().sum(0), (gm * xh).sum(0), rtol=1e-5, atol=1e-4 * float(gm.abs().sum(0).max()))
    for a, b in ((dg1, dg0), (db1, db0)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6
    assert float((dx1 - dx0).abs().max()) <= 1e-5 * float(dx0.abs().max())


def test_batchnorm_eval(ops):
    y = rnd(1, 2, 64, 9, 11)
    gamma, beta, rm, rv = rnd(2, 64) + 1, rnd(3, 64), rnd(4, 64), rnd(5, 64).abs() + 0.5
    ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, False, 0.1, 1e-5))
    z = ops.bn_apply(nhwc(y), rm.cuda(), rv.cuda(), gamma.cuda(), beta.cuda(), None, True, stat_is_var=True)
    assert float((nchw(z) - ref).abs().max()) <= 1e-5


def test_maxpool(ops):
    x = F.relu(rnd(1, 2, 64, 31, 45))  # many exact zeros: ties
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = rnd(2, *yr.shape)
    yr.backward(dy)
    y, idx = ops.maxpool_fwd(nhwc(x))
    assert torch.equal(nchw(y), yr.detach())
    dx = ops.maxpool_bwd(nhwc(dy), idx, (2, 31, 45, 64))
    # positions with x == 0 may receive a tie's gradient at a different zero: compare where x > 0,
    # and total gradient mass per window is conserved
    m = x > 0
    assert float(((nchw(dx) - xr.grad) * m).abs().max()) <= 1e-6
    np.testing.assert_allclose(float(dx.sum()), float(dy.sum()), rtol=1e-4)


@pytest.mark.parametrize('src,dst', [((4, 6), (7, 11)), ((25, 42), (50, 84)), ((13, 21), (25, 42)), ((1, 1), (2, 3))])
def test_upsample_add(ops, src, dst):
    top = rnd(1, 2, 64, *src)
    lat = rnd(2, 2, 64, *dst)
    tr, lr_ = top.clone().requires_grad_(True), lat.clone().requires_grad_(True)
    out = F.interpolate(tr, size=dst, mode='bilinear', align_corners=True) + lr_
    dout = rnd(3, *out.shape)
    out.backward(dout)
    o = ops.upsample_add_fwd(nhwc(top), nhwc(lat))
    assert float((nchw(o) - out.detach()).abs().max()) <= 1e-5
    base = rnd(4, 2, 64, *src)
    dtop = nhwc(base)
    ops.upsample_bwd(nhwc(dout), dtop, accumulate=True)
    assert float((nchw(dtop) - (tr.grad + base)).abs().max()) <= 2e-5 * max(1.0, float(tr.grad.abs().max()))


def test_colsum_and_eltwise(ops):
    x = rnd(1, 3001, 720)
    out = torch.zeros(720).cuda()
    ops.colsum(x.cuda(), 720, out)
    np.testing.assert_allclose(out.cpu().numpy(), x.double().sum(0).float().numpy(), rtol=1e-5, atol=1e-4)
    for r, c in ((33400, 256), (257, 8), (5000, 36), (131, 1000), (70, 2048), (999, 5), (1, 64)):
        x = rnd(7, r, c)
        out = torch.full((c,), 1.0).cuda()
        ops.colsum(x.cuda(), c, out, accumulate=True)
        np.testing.assert_allclose(out.cpu().numpy(), 1.0 + x.double().sum(0).float().numpy(), rtol=1e-5, atol=2e-4)
    a, b = rnd(2, 1003), torch.sigmoid(rnd(3, 1003))
    got = ops.eltwise(ops.EW_SCALE_SIGMOID_BWD, a.cuda(), b.cuda(), scalar=0.5).cpu()
    np.testing.assert_allclose(got.numpy(), (a * 0.5 * b * (1 - b)).numpy(), rtol=1e-6, atol=1e-7)


def test_adamw_matches_torch(ops):
    n1, n2 = 124, 4000
    p = rnd(1, n1 + n2)
    params_cpu = [p[:n1].clone().requires_grad_(True), p[n1:].clone().requires_grad_(True)]
    opt = torch.optim.AdamW([{'params': [params_cpu[0]], 'weight_decay': 0.0},
                             {'params': [params_cpu[1]], 'weight_decay': 1e-3}], lr=1e-4)
    pd, m, v = p.clone().cuda(), torch.zeros(n1 + n2).cuda(), torch.zeros(n1 + n2).cuda()
    for step in range(1, 4):
        g = rnd(10 + step, n1 + n2)
        params_cpu[0].grad, params_cpu[1].grad = g[:n1].clone(), g[n1:].clone()
        opt.step()
        ops.adamw_step(pd, g.cuda(), m, v, n1, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1e-3, step)
    ref = torch.cat([params_cpu[0].detach(), params_cpu[1].detach()])
    np.testing.assert_allclose(pd.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)


# ----------------------------------------------------------------------------------------- decode / NMS
@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
@pytest.mark.parametrize('nms_type', ['python_nms', 'diou_python_nms', 'torch_nms'])
def test_retina_decoder(ops, tag, nms_type, golden_dir):
    from sacv_amd.detection import decode
    H, W = gi.SHAPES[tag]
    cls, reg = gi.synth_preds_decode(21, 2, H, W)
    dec = decode.__dict__['RetinaDecoder'](nms_type=nms_type, nms_threshold=0.5)
    s, c, b = dec([[torch.tensor(x).cuda() for x in cls], [torch.tensor(x).cuda() for x in reg]])
    rs, rc, rb = odecode.retina_decode(cls, reg, nms_type=nms_type, nms_threshold=0.5)
    assert isinstance(s, np.ndarray) and s.dtype == np.float32 and s.shape == (2, 100) and b.shape == (2, 100, 4)
    assert np.array_equal(s, rs) and np.array_equal(c, rc) and np.array_equal(b, rb)
    G = np.load(os.path.join(golden_dir, 'decode.npz'))
    key = f'retina_{tag}_{nms_type}'
    if key + '_scores' in G:  # outputs of the reference's own RetinaDecoder
        assert np.array_equal(s, G[key + '_scores']) and np.array_equal(b, G[key + '_boxes'])
        assert np.array_equal(c, G[key + '_classes'])


@pytest.mark.parametrize('tag', ['TINY', 'MID', 'S1'])
@pytest.mark.parametrize('nms_type', ['soft_nms', 'soft_nms_gaussian'])
def test_retina_decoder_soft_nms(ops, tag, nms_type):
    """Batched Soft-NMS (the north-star's postprocess mode; the reference has none -> PARITY UNPINNED, the checker is
    oracle/decode.py::soft_nms restating Bodla et al. 2017).  Linear decay is pure fp32 mul / sub / div with
    -ffp-contract=off: scores, classes and boxes are compared bit-exact.  Gaussian decay goes through expf, whose
    last bit may differ from NumPy's and a deep pick carries the product of hundreds of such factors (measured: 2.5e-6
    relative at pick ~500): scores to 1e-5, and the order is compared where the oracle's neighbouring scores are at
    least 2e-5 apart (closer picks could legitimately swap)."""
    from sacv_amd.detection import decode
    H, W = gi.SHAPES[tag]
    cls, reg = gi.synth_preds_decode(21, 2, H, W)
    # at MID / S1 the first 100 picks never overlap (they equal hard NMS's): ask for all 1000 so that decayed boxes are
    # picked, re-decayed and pruned
    M = 100 if tag == 'TINY' else 1000
    dec = decode.__dict__['RetinaDecoder'](nms_type=nms_type, nms_threshold=0.5, max_object_num=M)
    s, c, b = dec([[torch.tensor(x).cuda() for x in cls], [torch.tensor(x).cuda() for x in reg]])
    rs, rc, rb = odecode.retina_decode(cls, reg, nms_type=nms_type, nms_threshold=0.5, max_object_num=M)
    assert s.shape == (2, M) and b.shape == (2, M, 4) and s.dtype == np.float32
    assert (rs > 0).sum() >= 20  # the case is not vacuous
    if nms_type == 'soft_nms':
        assert np.array_equal(s, rs) and np.array_equal(c, rc) and np.array_equal(b, rb)
        hard = odecode.retina_decode(cls, reg, nms_type='python_nms', nms_threshold=0.5, max_object_num=M)
        assert not np.array_equal(hard[0], rs)  # and it is not hard NMS under another name
    else:
        np.testing.assert_allclose(s, rs, rtol=1e-5, atol=0)
        for i in range(2):
            n = int((rs[i] > 0).sum())
            gap = np.abs(np.diff(rs[i, :n], append=rs[i, n - 1] - 1.0))
            gap = np.minimum(gap, np.concatenate([[1.0], gap[:-1]]))
            firm = gap > 2e-5
            assert np.array_equal(b[i, :n][firm], rb[i, :n][firm]) and np.array_equal(c[i, :n][firm], rc[i, :n][firm])
            assert np.all(s[i, n:] == -1) and np.all(b[i, n:] == 0)
    assert np.all(np.diff(np.where(s > 0, s, 0), axis=1) <= 0)  # picks come out in descending decayed score


@pytest.mark.parametrize('num_classes', [80, 160, 81, 8])
def test_decoder_selection_limits(ops, num_classes):
    """The selection stage on its own terms (no suppression: nms_threshold above any IoU, max_object_num = topn = 1000 so
    the whole sorted top-1000 comes out): thousands of EQUAL scores straddling the top-1000 cut (the radix select must walk
    past the score bits into the index bits; ties = ascending anchor index, the oracle's stable order), fewer candidates
    than topn, exactly topn, and the three scan kernels -- 128-prediction LDS tiles (80, 8 classes), 32-prediction tiles
    (160) and the scalar one (81: rows that are not 16-byte multiples).  Bit-exact."""
    from sacv_amd.detection import decode
    H, W = gi.SHAPES['MID']
    B = 3
    fs = synth.feature_sizes(H, W)
    rng = np.random.default_rng(5 + num_classes)
    cls = [(synth.uniform(70 + l, (B, fh, fw, 9, num_classes)) * 0.04).astype(np.float32)
           for l, (fw, fh) in enumerate(fs)]
    reg = [synth.normal(80 + l, (B, fh, fw, 9, 4), std=0.3) for l, (fw, fh) in enumerate(fs)]
    n_l = [fw * fh * 9 for fw, fh in fs]
    total = sum(n_l)

    def poke(b, idx, val, klass):
        for l, n in enumerate(n_l):
            if idx < n:
                cls[l].reshape(B, n, num_classes)[b, idx, klass] = val
                return
            idx -= n
    # image 0: 400 distinct high scores + 2500 anchors at exactly 0.5 (the cut falls inside the tie) + 300 at exactly 0.25
    hot = rng.permutation(total)[:3200]
    for q, a in enumerate(hot[:400]):
        poke(0, int(a), np.float32(0.6 + 0.39 * q / 400.0), int(a) % num_classes)
    for a in hot[400:2900]:
        poke(0, int(a), np.float32(0.5), (int(a) * 7) % num_classes)
    for a in hot[2900:]:
        poke(0, int(a), np.float32(0.25), 3)
    # image 1: 37 candidates; two classes of one prediction share the maximum (first index wins)
    for q, a in enumerate(hot[:37]):
        poke(1, int(a), np.float32(0.3 + 0.01 * q), 5)
    poke(1, int(hot[0]), np.float32(0.3), 2)
    # image 2: exactly 1000 candidates, all equal
    for a in hot[:1000]:
        poke(2, int(a), np.float32(0.125), int(a) % num_classes)
    dec = decode.__dict__['RetinaDecoder'](nms_type='python_nms', nms_threshold=1.5, max_object_num=1000)
    s, c, b = dec([[torch.tensor(x).cuda() for x in cls], [torch.tensor(x).cuda() for x in reg]])
    rs, rc, rb = odecode.retina_decode(cls, reg, nms_type='python_nms', nms_threshold=1.5, max_object_num=1000)
    assert (rs[0] == 0.5).sum() == 600 and (rs[0] >= 0).sum() == 1000 and (rs[1] >= 0).sum() == 37
    assert (rs[2] == 0.125).sum() == 1000
    assert rc[1][rs[1] == np.float32(0.3)][0] == 2
    assert np.array_equal(s, rs) and np.array_equal(c, rc) and np.array_equal(b, rb)


@pytest.mark.parametrize('nms_type', ['python_nms', 'diou_python_nms', 'torch_nms', 'soft_nms'])
def test_decoder_dense_suppression(ops, nms_type):
    """Crowded scenes: every anchor of a 14 x 14 patch of P3 (and a 5 x 5 patch of P4) is a candidate, so the 1000 selected
    boxes suppress each other across all sixteen 64-box blocks of the mask, and max_object_num = 1000 makes the greedy walk
    visit every block (the default 100 stops in the first ones).  Bit-exact vs the oracle, all suppression rules."""
    from sacv_amd.detection import decode
    H, W = gi.SHAPES['MID']
    B = 2
    fs = synth.feature_sizes(H, W)
    cls = [(synth.uniform(170 + l, (B, fh, fw, 9, 80)) * 0.04).astype(np.float32) for l, (fw, fh) in enumerate(fs)]
    reg = [synth.normal(180 + l, (B, fh, fw, 9, 4), std=0.2) for l, (fw, fh) in enumerate(fs)]
    q = 0
    for b in range(B):
        for l, (y0, x0, side) in enumerate([(5 + b, 7, 14), (3, 4 + b, 5)]):
            for y in range(y0, y0 + side):
                for x in range(x0, x0 + side):
                    for a in range(9):
                        q += 1
                        cls[l][b, y, x, a, (q * 7) % 80] = np.float32(0.06 + 0.9 * ((q * 7919) % 100003) / 100003.0)
    for M in (1000, 100):
        dec = decode.__dict__['RetinaDecoder'](nms_type=nms_type, nms_threshold=0.5, max_object_num=M)
        s, c, b = dec([[torch.tensor(x).cuda() for x in cls], [torch.tensor(x).cuda() for x in reg]])
        rs, rc, rb = odecode.retina_decode(cls, reg, nms_type=nms_type, nms_threshold=0.5, max_object_num=M)
        kept = (rs >= 0).sum(axis=1)
        assert np.array_equal(c, rc) and np.array_equal(b, rb) and np.array_equal(s, rs), (nms_type, M)
        if nms_type != 'soft_nms':
            assert kept.max() < 600 and kept.min() > 20, kept  # real suppression, beyond the first blocks


@pytest.mark.parametrize('B,num_classes,topn,max_obj,hw', [(5, 80, 300, 50, (96, 128)), (1, 508, 1000, 100, (64, 64)),
                                                             (16, 4, 64, 64, (32, 32)), (3, 20, 1, 1, (96, 128))])
def test_decoder_argument_corners(ops, B, num_classes, topn, max_obj, hw):
    """Corners of the decoder's argument space: odd batch sizes up to 16 images, topn / max_object_num far below the
    defaults (down to 1), the widest class count the 32-prediction tiles take (508), 4 classes (one float4 per row),
    pyramids whose upper levels are single pixels (a level smaller than one tile), images without candidates."""
    from sacv_amd.detection import decode
    H, W = hw
    fs = synth.feature_sizes(H, W)
    cls = [(synth.uniform(300 + l, (B, fh, fw, 9, num_classes)) * 0.04).astype(np.float32) for l, (fw, fh) in enumerate(fs)]
    reg = [synth.normal(310 + l, (B, fh, fw, 9, 4), std=0.25) for l, (fw, fh) in enumerate(fs)]
    q = 0
    for l, c in enumerate(cls):
        flat = c.reshape(B, -1, num_classes)
        for b in range(B):
            if b == 1:
                continue  # image 1: nothing above the threshold
            for a in range(b % 3, flat.shape[1], 3 + b % 2):
                q += 1
                flat[b, a, (q * 5) % num_classes] = np.float32(0.06 + 0.9 * ((q * 7919) % 100003) / 100003.0)
    dec = decode.__dict__['RetinaDecoder'](topn=topn, max_object_num=max_obj, nms_threshold=0.5)
    s, c, b = dec([[torch.tensor(x).cuda() for x in cls], [torch.tensor(x).cuda() for x in reg]])
    rs, rc, rb = odecode.retina_decode(cls, reg, topn=topn, max_object_num=max_obj, nms_threshold=0.5)
    assert s.shape == (B, max_obj) and b.shape == (B, max_obj, 4)
    assert np.array_equal(s, rs) and np.array_equal(c, rc) and np.array_equal(b, rb)
    if B > 1:
        assert (c[1] == -1).all() and (c[0] > -1).any()


@pytest.mark.parametrize('fcos', [False, True])
def test_decoder_dense_candidate_lists(ops, fcos):
    """min_score_threshold = 0: EVERY prediction is a candidate (15 354 / 1 706 per image at 256 x 320; more than the
    8 192 keys the selection keeps in registers for RetinaNet, so it walks the segments instead), with hundreds of exactly
    equal scores among them (maxima of 24-bit uniform noise): top-1000, order (ties by ascending prediction index) and
    NMS result bit-exact vs the oracle."""
    from sacv_amd.detection import decode
    H, W = gi.SHAPES['MID']
    preds = gi.synth_preds_decode(33, 2, H, W, fcos=fcos)
    for p_ in preds[0]:
        p_ *= 0.0
        p_ += (synth.uniform(p_.size % 977, p_.shape) * 0.04).astype(np.float32) + np.float32(1e-3)
    name = 'FCOSDecoder' if fcos else 'RetinaDecoder'
    for M in (100, 1000):
        dec = decode.__dict__[name](min_score_threshold=0.0, max_object_num=M)
        s, c, b = dec([[torch.tensor(x).cuda() for x in grp] for grp in preds])
        rs, rc, rb = (odecode.fcos_decode if fcos else odecode.retina_decode)(*preds, min_score_threshold=0.0,
                                                                             max_object_num=M)
        assert np.array_equal(s, rs) and np.array_equal(c, rc) and np.array_equal(b, rb), (fcos, M)
    assert (rs[0] >= 0).sum() > 100


def test_fcos_decoder_soft_nms_and_edge_cases(ops):
    """FCOS front end + Soft-NMS; an image without any score above the threshold; max_object_num larger than the
    number of survivors; a prune level that removes everything a decay touches (== hard suppression at >= thr)."""
    from sacv_amd.detection import decode
    H, W = gi.SHAPES['MID']
    cls, reg, ctr = gi.synth_preds_decode(23, 2, H, W, fcos=True)
    f = lambda lst: [torch.tensor(x).cuda() for x in lst]
    dec = decode.__dict__['FCOSDecoder'](nms_type='soft_nms', nms_threshold=0.6)
    s, c, b = dec([f(cls), f(reg), f(ctr)])
    rs, rc, rb = odecode.fcos_decode(cls, reg, ctr, nms_type='soft_nms', nms_threshold=0.6)
    assert np.array_equal(s, rs) and np.array_equal(c, rc) and np.array_equal(b, rb)
    # image 1 has nothing above the score threshold
    cls0 = [x.copy() for x in cls]
    for x in cls0:
        x[1] = 1e-4  # FCOS score = sqrt(cls * centerness) <= 0.01 < 0.05
    s, c, b = dec([f(cls0), f(reg), f(ctr)])
    rs, rc, rb = odecode.fcos_decode(cls0, reg, ctr, nms_type='soft_nms', nms_threshold=0.6)
    assert np.array_equal(s, rs) and np.all(s[1] == -1) and np.all(c[1] == -1) and np.all(b[1] == 0)
    # few survivors, many slots; prune level 0.5 drops every decayed box whose score falls to <= 0.5
    dec = decode.__dict__['FCOSDecoder'](nms_type='soft_nms', nms_threshold=0.6, max_object_num=300,
                                          min_score_threshold=0.3, soft_nms_prune_score=0.5)
    s, c, b = dec([f(cls), f(reg), f(ctr)])
    rs, rc, rb = odecode.fcos_decode(cls, reg, ctr, nms_type='soft_nms', nms_threshold=0.6, max_object_num=300,
                                     min_score_threshold=0.3, soft_nms_prune_score=0.5)
    assert s.shape == (2, 300) and np.array_equal(s, rs) and np.array_equal(c, rc) and np.array_equal(b, rb)
    assert np.all(s[s > 0] > 0.5)


# ----------------------------------------------------------------------------------------- data boundary
def test_device_collater_and_eval_postprocess(ops):
    """device-side Normalize + DetectionCollater and the evaluation box post-steps, bit-exact
    against the NumPy restatement of the reference's host code."""
    from sacv_amd.detection import common
    rng = np.random.RandomState(0)
    samples = []
    for h, w, n in ((37, 51, 3), (64, 40, 0), (5, 66, 7)):
        samples.append({'image': rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8),
                        'annots': rng.rand(n, 5).astype(np.float32) * 30, 'scale': 0.5 + rng.rand(), 'size': (h * 2, w * 2)})
    col = common.DetectionCollater(resize=40, resize_type='retina_style', max_annots_num=10)  # R = 67
    out = col(samples)
    ref_img, ref_ann = odecode.collate(samples, resize=40, max_annots_num=10)
    img = out['image']
    assert tuple(img.shape) == (3, 3, 67, 67) and img.is_contiguous(memory_format=torch.channels_last)
    assert np.array_equal(img.permute(0, 2, 3, 1).cpu().numpy(), ref_img)
    assert np.array_equal(out['annots'].numpy(), ref_ann)
    assert out['scale'].dtype == np.float32 and out['size'].shape == (3, 2)
    boxes = (rng.rand(3, 100, 4).astype(np.float32) * 200 - 20)
    boxes[:, 60:] = 0
    got = common.eval_postprocess(torch.tensor(boxes).cuda(), out['scale'], out['size']).cpu().numpy()
    assert np.array_equal(got, odecode.eval_postprocess(boxes, out['scale'], out['size']))
