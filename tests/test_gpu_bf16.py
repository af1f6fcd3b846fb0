"""bf16 matrix-core path (the autocast ResNet50 recipe): kernel parity against fp32 references on
bf16-rounded inputs (products of bf16 values are exact in fp32, so the only differences are the fp32 summation order
and the final rounding of bf16 outputs: 2^-8 relative), and the end-to-end step against the reference model run under
torch.autocast('cpu', bfloat16) (tests/golden/cls_bf16.*) with the fp32 run of the same inputs as the yardstick.

Tolerances: bf16 outputs |err| <= 2^-8 |ref| + 2e-3 max|ref| (rounding + accumulation order over K <= 4608); fp32
outputs (logits, weight gradients) 2e-3 of the tensor's scale; end-to-end loss within max(3e-2, 2x the distance
of the reference's own bf16 run from its fp32 run)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth
from tests import golden_inputs as gi

BF = torch.bfloat16


def rnd(seed, *shape, std=1.0):
    return torch.tensor(synth.normal(seed, shape, std=std))


def q(t):
    """bf16 rounding of an fp32 tensor, kept as fp32."""
    return t.to(BF).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def close_bf16(got, ref, what):
    got, ref = got.float().cpu(), ref.float()
    tol = (2.0**-8) * ref.abs() + 2e-3 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - ref).abs().max()), float(ref.abs().max()))


@pytest.mark.parametrize('n,h,w,cin,cout,k,s', [(2, 13, 17, 64, 64, 3, 1), (1, 9, 11, 128, 256, 1, 1),
                                                (2, 14, 14, 72, 136, 3, 2), (3, 8, 8, 256, 64, 1, 2),
                                                (2, 7, 7, 512, 512, 3, 1), (4, 1, 1, 2048, 1000, 1, 1)])
def test_conv_bf16_forward_dgrad_wgrad(n, h, w, cin, cout, k, s):
    from sacv_amd import ops
    from sacv_amd.engine import DetectorEngine
    p = k // 2
    x = q(rnd(1, n, cin, h, w))
    wt = q(rnd(2, cout, cin, k, k, std=(2.0 / (cin * k * k))**0.5))
    bias = rnd(3, cout)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, s, p)
    dy = q(rnd(4, *y_ref.shape))
    y_ref.backward(dy)
    xd = nhwc(x).cuda().to(BF)
    wp = wt.permute(0, 2, 3, 1).contiguous().view(cout, -1).cuda().to(BF)       # [Cout][taps*Cin]
    ho, wo = y_ref.shape[2], y_ref.shape[3]
    # forward (+ fused statistics of the rounded outputs)
    (y,), st = ops.conv_gemm([xd], wp, cout, k, k, s, p, 0, [(ho, wo)], stats=True)
    assert y.dtype == BF
    close_bf16(y.permute(0, 3, 1, 2), y_ref.detach(), 'forward')
    yf = y.float().view(-1, cout)
    torch.testing.assert_close(st[:, 0].sum(0).cpu(), yf.sum(0).cpu(), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(st[:, 1].sum(0).cpu(), (yf * yf).sum(0).cpu(), rtol=1e-4, atol=1e-2)
    # fp32 destination + bias + accumulate
    base = torch.ones((n, ho, wo, cout), device='cuda')
    y32 = ops.conv_gemm([xd], wp, cout, k, k, s, p, 0, [(ho, wo)], bias=bias.cuda(),
                        flags=ops.CFH_OUT_F32 | ops.CF_ACCUM, outs=[base])[0]
    ref32 = (y_ref.detach() + bias.view(1, -1, 1, 1) + 1.0).permute(0, 2, 3, 1)
    assert float((y32.cpu() - ref32).abs().max()) <= 2e-3 * float(ref32.abs().max())
    # data gradient through the engine's dispatcher (stride-2 convs run as parity classes / strided scatter)
    eng = DetectorEngine({}, {})
    eng.amp = True
    wd = wt.permute(1, 2, 3, 0).contiguous().view(cin, -1).cuda().to(BF)          # [Cin][taps*Cout]
    eng._dgrad_panel = lambda _w: wd
    dyd = nhwc(dy).cuda().to(BF)
    dx = eng._dgrad(dyd, wt.cuda(), (n, h, w, cin), k, s, p)
    assert dx.dtype == BF
    close_bf16(dx.permute(0, 3, 1, 2), xr.grad, 'dgrad')
    # weight gradient (fp32 result), OIHW and channels_last destinations
    dw = torch.empty((cout, cin, k, k), device='cuda')
    ops.conv_wgrad([dyd], [xd], dw, cin, k, k, s, p)
    assert float((dw.cpu() - wr.grad).abs().max()) <= 2e-3 * float(wr.grad.abs().max()), 'wgrad'
    if k > 1:
        dwl = torch.empty((cout, k, k, cin), device='cuda').permute(0, 3, 1, 2)
        ops.conv_wgrad([dyd], [xd], dwl, cin, k, k, s, p)
        assert float((dwl.cpu() - wr.grad).abs().max()) <= 2e-3 * float(wr.grad.abs().max()), 'wgrad channels_last'


def test_linear_gemm_dma_kernel_ragged_edges():
    """Stride-1 1x1 problems with >= 300 tiles of 256 x 128 go to gemm_nt_bf16_dma_kernel (csrc/gemm_bf16.hip): M and N
    that are not multiples of the tile (clamped rows, masked stores), all epilogue options, per-128-row statistics."""
    from sacv_amd import ops
    m, k, n = 256 * 20 + 37, 192, 128 * 15 + 8  # 21 x 16 = 336 tiles
    x = q(rnd(11, 1, k, 1, m))                       # NCHW [1, k, 1, m]
    wt = q(rnd(12, n, k, 1, 1, std=(2.0 / k)**0.5))
    bias = rnd(13, n)
    y_ref = F.conv2d(x, wt)
    xd = nhwc(x).cuda().to(BF)
    wp = wt.view(n, k).cuda().to(BF)
    (y,), st = ops.conv_gemm([xd], wp, n, 1, 1, 1, 0, 0, [(1, m)], stats=True)
    close_bf16(y.permute(0, 3, 1, 2), y_ref, 'gemm forward')
    assert st.shape[0] == -(-m // 128)
    yf = y.float().view(-1, n)
    for r in (0, 7, st.shape[0] - 1):  # per-granule sums, incl. the ragged last one
        blk = yf[r * 128:(r + 1) * 128]
        torch.testing.assert_close(st[r, 0].cpu(), blk.sum(0).cpu(), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(st[r, 1].cpu(), (blk * blk).sum(0).cpu(), rtol=1e-4, atol=1e-2)
    base = torch.full((1, 1, m, n), 0.5, device='cuda')
    y32 = ops.conv_gemm([xd], wp, n, 1, 1, 1, 0, 0, [(1, m)], bias=bias.cuda(),
                        flags=ops.CFH_OUT_F32 | ops.CF_ACCUM, outs=[base])[0]
    ref32 = (y_ref + bias.view(1, -1, 1, 1) + 0.5).permute(0, 2, 3, 1)
    assert float((y32.cpu() - ref32).abs().max()) <= 2e-3 * float(ref32.abs().max())
    # the memory right behind the outputs is untouched (rows >= M are computed from clamped rows but never stored)
    guard = torch.full((m * n + 4096,), 7.0, device='cuda', dtype=BF)
    out = guard[:m * n].view(1, 1, m, n)
    ops.conv_gemm([xd], wp, n, 1, 1, 1, 0, 0, [(1, m)], outs=[out])
    assert bool((guard[m * n:] == 7.0).all())


def test_bn_and_pooling_bf16_match_the_fp32_kernels_on_rounded_data():
    from sacv_amd import ops
    n, h, w, c = 3, 9, 7, 72
    y = q(rnd(5, n, h, w, c)) * 1.7 + 0.3
    y = q(y)
    res = q(rnd(6, n, h, w, c))
    dz = q(rnd(7, n, h, w, c))
    gamma, beta = (1.0 + 0.2 * rnd(8, c)), 0.2 * rnd(9, c)
    mean, var = y.view(-1, c).mean(0), y.view(-1, c).var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    G = lambda t: t.cuda()
    for residual, relu in ((None, True), (res, True), (None, False)):
        z32 = ops.bn_apply(G(y), G(mean), G(invstd), G(gamma), G(beta), G(residual) if residual is not None else None,
                           relu)
        zh = ops.bn_apply(G(y).to(BF), G(mean), G(invstd), G(gamma), G(beta),
                          G(residual).to(BF) if residual is not None else None, relu)
        assert zh.dtype == BF and torch.equal(zh, z32.to(BF))          # same fp32 arithmetic, one final rounding
        dg32, db32, dgh, dbh = (torch.zeros(c, device='cuda') for _ in range(4))
        zq = z32.to(BF)
        recompute = relu and residual is None
        dy32, g32 = ops.bn_train_bwd(G(dz), zq.float() if (relu and not recompute) else None, G(y), G(mean),
                                     G(invstd), G(gamma), dg32, db32, want_g=True,
                                     relu_beta=G(beta) if recompute else None)
        dyh, gh = ops.bn_train_bwd(G(dz).to(BF), zq if (relu and not recompute) else None, G(y).to(BF), G(mean),
                                   G(invstd), G(gamma), dgh, dbh, want_g=True,
                                   relu_beta=G(beta) if recompute else None)
        assert dyh.dtype == BF and gh.dtype == BF
        if recompute:  # the mask is recomputed from the ROUNDED forward value: compare against the stored-z mask
            dy32, g32 = ops.bn_train_bwd(G(dz), zq.float(), G(y), G(mean), G(invstd), G(gamma), dg32, db32,
                                         want_g=True)
        torch.testing.assert_close(dgh, dg32, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dbh, db32, rtol=1e-5, atol=1e-5)
        assert torch.equal(dyh, dy32.to(BF)) and torch.equal(gh, g32.to(BF))
    x = q(rnd(10, 2, 15, 13, 64))
    yp32, i32 = ops.maxpool_fwd(G(x), 3, 2, 1)
    yph, ih = ops.maxpool_fwd(G(x).to(BF), 3, 2, 1)
    assert torch.equal(yph, yp32.to(BF)) and torch.equal(ih, i32)
    dyp = q(rnd(11, *yp32.shape))
    assert torch.equal(ops.maxpool_bwd(G(dyp).to(BF), ih, tuple(x.shape), 3, 2, 1),
                       ops.maxpool_bwd(G(dyp), i32, tuple(x.shape), 3, 2, 1).to(BF))
    a32 = ops.avgpool_fwd(G(x))
    ah = ops.avgpool_fwd(G(x).to(BF))
    assert ah.dtype == BF and float((ah.float() - a32).abs().max()) <= 2.0**-8 * float(a32.abs().max())
    da = q(rnd(12, 2, 1, 1, 64))
    assert torch.equal(ops.avgpool_bwd(G(da).to(BF), tuple(x.shape)), ops.avgpool_bwd(G(da), tuple(x.shape)).to(BF))
    t = rnd(13, 1001)
    assert torch.equal(ops.cast_bf16(G(t)).cpu(), t.to(BF)) and torch.equal(ops.cast_f32(G(t).to(BF)).cpu(), q(t))


@pytest.mark.parametrize('name', ['cls_64', 'cls_224', 'cls_64_b32'])
def test_resnet50_amp_step_against_reference_autocast(golden_dir, name):
    from sacv_amd.classification import backbones, losses
    meta = json.load(open(os.path.join(golden_dir, 'cls_bf16.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'cls_bf16.npz'))
    B, R, nc = meta['B'], meta['R'], meta['num_classes']
    net = backbones.__dict__['resnet50'](num_classes=nc)
    net.load_state_dict(omodel.init_resnet50_cls_state(0, nc), strict=True)
    net = net.cuda().train().set_amp(True)
    crit = losses.__dict__['CELoss']()
    x = torch.tensor(synth.images(11, B, R, R)).cuda()
    lab = torch.tensor(gi.labels_for(23, B, nc)).cuda()
    logits = net(x)
    assert logits.dtype == torch.float32
    loss = crit(logits, lab)
    loss.backward()
    ref_gap = abs(meta['loss'] - meta['loss_fp32'])          # what bf16 costs the reference itself
    gap = abs(float(loss) - meta['loss_fp32'])
    print(f'{name}: loss gpu-bf16 {float(loss):.5f} ref-bf16 {meta["loss"]:.5f} ref-fp32 {meta["loss_fp32"]:.5f}')
    assert abs(float(loss) - meta['loss']) <= max(3e-2, 2 * ref_gap) * max(1.0, meta['loss'])
    assert gap <= max(3e-2, 3 * ref_gap) * max(1.0, meta['loss_fp32'])
    l32, lbf = arrs[f'{name}_logits_fp32'], arrs[f'{name}_logits']
    ref_err = np.abs(lbf - l32).max() / np.abs(l32).max()
    err = np.abs(logits.detach().cpu().numpy() - l32).max() / np.abs(l32).max()
    print(f'  logits vs fp32: gpu-bf16 {err:.3e}, reference-bf16 {ref_err:.3e}')
    assert err <= max(3e-2, 2.5 * ref_err)
    rel_g, rel_r = [], []
    for k, p in net.named_parameters():
        n32 = meta['grad_norm_fp32'][k]
        rel_g.append(abs(float(p.grad.double().norm()) - n32) / (n32 + 1e-12))
        rel_r.append(abs(meta['grad_norm'][k] - n32) / (n32 + 1e-12))
    print(f'  grad norms vs fp32: gpu-bf16 median {np.median(rel_g):.3e} max {max(rel_g):.3e}; '
          f'reference-bf16 median {np.median(rel_r):.3e} max {max(rel_r):.3e}')
    # (cls_64: 4 images of 64x64 leave 2x2 pixels at layer4 -- BatchNorm over 16 values amplifies any rounding)
    assert float(np.median(rel_g)) <= max(6e-2, 2.5 * float(np.median(rel_r)))
    assert max(rel_g) <= max(0.25, 2.5 * max(rel_r))


def test_resnet50_amp_trainer_bs256():
    """The recipe's batch (256 x 3 x 224 x 224) through the bf16 trainer: finite decreasing loss over 3 SGD steps at a
    linear-regime rate, fp32 master weights move, bf16 and fp32 first-step losses agree to bf16 accuracy."""
    from sacv_amd.classification import backbones, losses
    from sacv_amd.train import ClassificationTrainer
    B, R = 256, 224
    x = torch.cat([torch.tensor(synth.images(300 + i, 32, R, R)) for i in range(8)]).cuda()
    lab = torch.tensor(gi.labels_for(41, B, 1000)).cuda()
    out = {}
    for amp in (False, True):
        net = backbones.__dict__['resnet50'](num_classes=1000)
        net.load_state_dict(omodel.init_resnet50_cls_state(0, 1000), strict=True)
        net = net.cuda().train()
        tr = ClassificationTrainer(net, losses.__dict__['CELoss'](), lr=0.01, momentum=0.9, weight_decay=1e-4,
                                   use_amp=amp)
        p0 = net.arena.params.clone()
        ls = [tr.finish_step(tr.step(x, lab))['loss'] for _ in range(3)]
        assert all(np.isfinite(ls)) and ls[2] < ls[0], (amp, ls)
        assert float((net.arena.params - p0).abs().max()) > 0 and net.arena.params.dtype == torch.float32
        out[amp] = ls
        del net, tr
        torch.cuda.empty_cache()
    print('bs256 losses fp32', out[False], 'bf16', out[True])
    assert abs(out[True][0] - out[False][0]) <= 2e-2 * out[False][0]


@pytest.mark.parametrize('n,h,w,cin', [(40, 56, 56, 64), (70, 28, 61, 128), (600, 14, 14, 64)])
def test_halo_resident_3x3_kernel(n, h, w, cin):
    """3x3 / stride 1 / pad 1 problems with 64 output channels and >= 512 tiles of 256 padded positions go to
    conv3x3_halo_bf16_kernel (csrc/conv3x3_halo_bf16.hip): forward with the fused BatchNorm statistics and the data
    gradient against torch fp32 on the bf16-rounded operands (the bf16 output tolerance of this file), one and two
    64-channel chunks, the widest image the 384-row LDS image takes (W = 61), image / batch edges (ragged last tile)."""
    from sacv_amd import ops
    from sacv_amd._lib import lib
    cout = 64
    assert lib.sacv_abi_version() >= 2
    x = q(rnd(21, n, cin, h, w))
    wt = q(rnd(22, cout, cin, 3, 3, std=(2.0 / (cin * 9))**0.5))
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wt, None, 1, 1)
    xd = nhwc(x).cuda().to(BF)
    wp = wt.permute(0, 2, 3, 1).contiguous().view(cout, -1).cuda().to(BF)
    (y,), st = ops.conv_gemm([xd], wp, cout, 3, 3, 1, 1, 0, [(h, w)], stats=True)
    close_bf16(y.permute(0, 3, 1, 2), y_ref.detach(), 'forward')
    yf = y.float().view(-1, cout)
    # the statistics rows beyond the kernel's tiles are zero, the used ones sum to the column sums of the ROUNDED output
    tiles = -(-(n * (h + 2) * (w + 2)) // 256)
    assert st.shape[0] == -(-(n * h * w) // 128) and tiles < st.shape[0] and not bool(st[tiles:].any())
    torch.testing.assert_close(st[:, 0].sum(0).cpu(), yf.sum(0).cpu(), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(st[:, 1].sum(0).cpu(), (yf * yf).sum(0).cpu(), rtol=1e-4, atol=2e-2)
    # data gradient: dy has 64 channels, dx has `cin` -- the kernel needs 64 OUTPUT channels, so cin == 64 here
    if cin == 64:
        dy = q(rnd(23, *y_ref.shape))
        y_ref.backward(dy)
        wd = wt.permute(1, 2, 3, 0).contiguous().view(cin, -1).cuda().to(BF)      # [Cin][taps*Cout]
        dx = ops.conv_gemm([nhwc(dy).cuda().to(BF)], wd, cin, 3, 3, 1, 1, 1, [(h, w)])[0]
        close_bf16(dx.permute(0, 3, 1, 2), xr.grad, 'dgrad')


@pytest.mark.parametrize('n,cin,h,w,cout,k,zmask', [
    (2, 256, 50, 67, 64, 1, False),      # 1x1, conv_nt_bf16<128,128> with the BatchNorm-backward epilogue
    (2, 64, 37, 41, 256, 1, True),       # N = 64: the <128,64> kernel; mask from z, accumulating destination
    (2, 128, 26, 31, 128, 3, False),     # 3x3
    (1, 256, 1, 256 * 150 + 19, 128, 1, False),  # 302 tiles of 256 x 128: gemm_nt_bf16_dma_kernel, ragged last granule
    (1, 128, 1, 256 * 300 + 77, 192, 1, True),   # the same kernel, mask from z + accumulate
])
def test_bn_backward_first_stage_in_the_bf16_data_gradient_epilogue(n, cin, h, w, cout, k, zmask):
    """sacv_conv2d_gemm_bf16_bnbwd + sacv_bn_train_bwd_partial_bf16 against the unfused pair (bf16 twin of
    tests/test_gpu_kernels.py::test_bn_backward_first_stage_in_the_data_gradient_epilogue): dz bit-identical, the
    first-stage sums against fp64 sums of the same bf16 values, dgamma / dbeta / dy to summation noise (dy is bf16:
    one rounding step of its scale)."""
    from sacv_amd import ops
    p = k // 2
    wt = q(rnd(2, cout, cin, k, k, std=(1.0 / (cin * k * k))**0.5))
    wd = wt.permute(1, 2, 3, 0).contiguous().view(cin, -1).cuda().to(BF)   # [Cin][taps*Cout]
    dy = nhwc(rnd(3, n, cout, h, w)).cuda().to(BF)
    y = nhwc(rnd(4, n, cin, h, w)).cuda().to(BF)
    gamma = (1.0 + 0.2 * rnd(5, cin)).cuda()
    beta = (0.2 * rnd(6, cin)).cuda()
    yf = y.float().view(-1, cin)
    mean = yf.mean(0)
    invstd = 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5)
    res = nhwc(rnd(7, n, cin, h, w)).cuda().to(BF) if zmask else None
    z = ops.bn_apply(y, mean, invstd, gamma, beta, res, True)
    base = nhwc(rnd(8, n, cin, h, w)).cuda().to(BF) if zmask else None
    flags = ops.CF_ACCUM if zmask else 0

    def grads():
        return torch.zeros(cin, device='cuda'), torch.zeros(cin, device='cuda')
    out0 = base.clone() if zmask else None
    dz0 = ops.conv_gemm([dy], wd, cin, k, k, 1, p, 1, [(h, w)], flags=flags, outs=[out0] if zmask else None)[0]
    dg0, db0 = grads()
    dx0, g0 = ops.bn_train_bwd(dz0, z if zmask else None, y, mean, invstd, gamma, dg0, db0, want_g=True,
                               relu_beta=None if zmask else beta)
    out1 = base.clone() if zmask else None
    dz1, st = ops.conv_gemm_bnbwd(dy, wd, cin, k, k, p, (h, w), y, mean, invstd, bn_gamma=None if zmask else gamma,
                                  bn_beta=None if zmask else beta, bn_z=z if zmask else None, out=out1, flags=flags)
    assert dz1.dtype == BF and torch.equal(dz1, dz0)
    assert st.shape == (-(-(n * h * w) // 128), 2, cin)
    dg1, db1 = grads()
    dx1, g1 = ops.bn_train_bwd(dz1, z if zmask else None, y, mean, invstd, gamma, dg1, db1, want_g=True,
                               relu_beta=None if zmask else beta, partial=st)
    assert torch.equal(g1, g0)
    gm = g0.float().reshape(-1, cin).double()
    xh = (yf.double() - mean.double()) * invstd.double()
    scale = float(gm.abs().sum(0).max())
    assert torch.allclose(st[:, 0].double().sum(0), gm.sum(0), rtol=1e-5, atol=1e-4 * scale)
    assert torch.allclose(st[:, 1].double().sum(0), (gm * xh).sum(0), rtol=1e-5, atol=1e-4 * scale)
    # one granule on its own (the ragged last one included): rows r*128 .. of the same fp64 sums
    for r in (0, st.shape[0] - 1):
        blk = slice(r * 128, (r + 1) * 128)
        assert torch.allclose(st[r, 0].double(), gm[blk].sum(0), rtol=1e-4, atol=1e-4 * float(gm[blk].abs().sum(0).max()))
        assert torch.allclose(st[r, 1].double(), (gm[blk] * xh[blk]).sum(0), rtol=1e-4,
                              atol=1e-4 * float(gm[blk].abs().sum(0).max()))
    for a, b in ((dg1, dg0), (db1, db0)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6
    assert float((dx1.float() - dx0.float()).abs().max()) <= 2.0**-7 * float(dx0.float().abs().max())


def test_classifier_step_bf16_fused_bn_backward_matches_unfused():
    """ResNet50 under autocast, 8 x 96 x 96: the backward pass with BatchNorm's first stage in the data-gradient
    epilogues (engine.fuse_bn_bwd_amp) against the same pass with every layer's own first stage: identical forward
    pass, and gradients that differ by the summation order of the per-channel sums only (a changed last bit of one
    bf16 dy moves everything upstream by rounding noise: bound = 1/4 of the amp fixture test's own gate)."""
    from sacv_amd.classification import backbones, losses
    nc, B, R = 100, 8, 96
    x = torch.tensor(synth.images(11, B, R, R)).cuda()
    lab = torch.tensor(gi.labels_for(23, B, nc)).cuda()
    runs = []
    for fuse in (True, False):
        net = backbones.__dict__['resnet50'](num_classes=nc)
        net.load_state_dict(omodel.init_resnet50_cls_state(0, nc), strict=True)
        net = net.cuda().train().set_amp(True)
        net.engine.fuse_bn_bwd_amp = fuse
        loss = losses.__dict__['CELoss']()(net(x), lab)
        loss.backward()
        runs.append((float(loss), {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
        del net
    (la, ga), (lb, gb) = runs
    assert la == lb
    rel = []
    for k in ga:
        na, nb = float(ga[k].double().norm()), float(gb[k].double().norm())
        rel.append(abs(na - nb) / (nb + 1e-12))
        d = float((ga[k] - gb[k]).abs().max()) / (float(gb[k].abs().max()) + 1e-12)
        assert d <= 6e-2, (k, d)
    print('fused vs unfused bf16 BatchNorm backward: grad-norm rel diff median %.2e max %.2e' % (np.median(rel), max(rel)))
    assert float(np.median(rel)) <= 1.5e-2 and max(rel) <= 6e-2
