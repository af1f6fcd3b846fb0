"""GPU parity of the ResNet50 ImageNet classification row through the C ABI: the tail
kernels against torch fp32 references, the whole model + CELoss + SGD against the oracle and against the
fixtures the REFERENCE model produced (tests/golden/cls.*, oracle/make_golden.py section cls)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth
from tests import golden_inputs as gi


def test_avgpool_fwd_bwd():
    from sacv_amd import ops
    x = torch.tensor(synth.normal(1, (5, 7, 7, 64))).cuda()
    out = ops.avgpool_fwd(x)
    ref = x.mean(dim=(1, 2), keepdim=True)
    assert out.shape == (5, 1, 1, 64)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    dy = torch.tensor(synth.normal(2, (5, 1, 1, 64))).cuda()
    dx = ops.avgpool_bwd(dy, x.shape)
    torch.testing.assert_close(dx, (dy / 49.0).expand(5, 7, 7, 64).contiguous(), rtol=1e-6, atol=0)


@pytest.mark.parametrize('n,c', [(7, 1000), (256, 1000), (3, 40), (1, 5)])
def test_softmax_ce_matches_torch(n, c):
    from sacv_amd import ops
    logits = (torch.tensor(synth.normal(3, (n, c))) * 4).cuda()
    labels = torch.tensor(gi.labels_for(4, n, c)).cuda()
    loss, dl = ops.softmax_ce(logits, labels)
    ref_in = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(ref_in, labels, reduction='mean')
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    torch.testing.assert_close(dl, ref_in.grad, rtol=1e-4, atol=1e-7)
    loss2, dl2 = ops.softmax_ce(logits, labels, want_grad=False)
    assert dl2 is None and float(loss2) == float(loss)


def test_sgd_step_matches_torch():
    from sacv_amd import ops
    n, split = 4096, 1024
    p0 = torch.tensor(synth.normal(5, (n,)))
    p = p0.clone().cuda()
    buf = torch.zeros(n).cuda()
    a = torch.nn.Parameter(p0[:split].clone())
    b = torch.nn.Parameter(p0[split:].clone())
    opt = torch.optim.SGD([{'params': [a], 'weight_decay': 0.0}, {'params': [b], 'weight_decay': 1e-4}], lr=0.1,
                          momentum=0.9)
    for it in range(3):
        g = torch.tensor(synth.normal(6 + it, (n,)))
        a.grad, b.grad = g[:split].clone() / 2, g[split:].clone() / 2
        opt.step()
        ops.sgd_step(p, g.cuda(), buf, split, 0.1, 0.9, 0.0, 1e-4, first_step=(it == 0), grad_scale=0.5)
        torch.testing.assert_close(p.cpu(), torch.cat([a.data, b.data]), rtol=1e-6, atol=1e-7)


def build(nc):
    from sacv_amd.classification import backbones
    net = backbones.__dict__['resnet50'](num_classes=nc)
    net.load_state_dict(omodel.init_resnet50_cls_state(0, nc), strict=True)
    return net.cuda().train()


@pytest.mark.parametrize('name', ['cls_64', 'cls_224', 'cls_96_nc40'])
def test_classifier_against_reference_fixture(golden_dir, name):
    """logits, loss (1e-4), BN statistics and every parameter-gradient norm vs the reference's fp32 CPU run."""
    from sacv_amd.classification import losses
    meta = json.load(open(os.path.join(golden_dir, 'cls.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'cls.npz'))
    B, R, nc = meta['B'], meta['R'], meta['num_classes']
    net = build(nc)
    x = torch.tensor(synth.images(11, B, R, R)).cuda()
    lab = torch.tensor(gi.labels_for(23, B, nc)).cuda()
    logits = net(x)
    assert tuple(logits.shape) == (B, nc)
    ref = arrs[f'{name}_logits']
    err = float(np.abs(logits.detach().cpu().numpy() - ref).max())
    print(f'{name}: logits max abs err {err:.2e} (max |logit| {np.abs(ref).max():.2f})')
    # 53 convolutions + batch statistics over as few as 16 samples (64x64 inputs): fp32 summation-order noise
    assert err <= 3e-4 * max(1.0, float(np.abs(ref).max()))
    loss = losses.__dict__['CELoss']()(logits, lab)
    assert abs(float(loss) - meta['loss']) <= 1e-4 * max(1.0, abs(meta['loss']))
    loss.backward()
    rel = []
    for k, p in net.named_parameters():
        n = float(p.grad.double().norm())
        rel.append(abs(n - meta['grad_norm'][k]) / (meta['grad_norm'][k] + 1e-12))
        assert rel[-1] <= 5e-2, (k, rel[-1])
    assert float(np.median(rel)) <= 1e-2
    print(f'{name}: grad-norm rel err median {np.median(rel):.2e} max {max(rel):.2e}')
    g = dict(net.named_parameters())
    np.testing.assert_allclose(g['fc.bias'].grad.cpu().numpy(), arrs[f'{name}_grad_fc.bias'], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(net.state_dict()['conv1.layer.1.running_var'].cpu().numpy(),
                               arrs[f'{name}_buf_conv1.layer.1.running_var'], rtol=1e-4, atol=5e-6)


def test_classifier_gradients_vs_fp64_oracle():
    """Every gradient tensor vs the oracle in fp64, with the fp32 CPU oracle's own error as the yardstick
    (same rationale as tests/test_gpu_e2e.py::test_forward_loss_backward_vs_oracle)."""
    from sacv_amd.classification import losses
    from tests.test_cls_cpu import oracle_cls_run
    meta = dict(B=4, R=64, num_classes=1000)
    _, loss_r, params_r, _, _, _ = oracle_cls_run('cls_64', meta)
    _, _, params_64, _, _, _ = oracle_cls_run('cls_64', meta, torch.float64)
    net = build(1000)
    x = torch.tensor(synth.images(11, 4, 64, 64)).cuda().contiguous(memory_format=torch.channels_last)
    lab = torch.tensor(gi.labels_for(23, 4, 1000)).cuda()
    loss = losses.__dict__['CELoss']()(net(x), lab)
    assert abs(float(loss) - float(loss_r)) <= 1e-4 * max(1.0, abs(float(loss_r)))
    loss.backward()
    num_g = num_c = den = 0.0
    for k, p in net.named_parameters():
        exact = params_64[k].grad
        num_g += float((p.grad.double().cpu() - exact).norm())**2
        num_c += float((params_r[k].grad.double() - exact).norm())**2
        den += float(exact.norm())**2
    e_gpu, e_cpu = (num_g / den)**0.5, (num_c / den)**0.5
    print(f'classifier gradient rel-L2 vs fp64: gpu {e_gpu:.2e}, cpu fp32 {e_cpu:.2e}')
    assert e_gpu <= 2.0 * e_cpu + 1e-3


@pytest.mark.parametrize('name', ['cls_64', 'cls_224', 'cls_96_nc40'])
def test_classification_trainer_two_sgd_steps(golden_dir, name):
    """ClassificationTrainer (forward, fused CE, backward, fused SGD over the arena) for two steps of the
    recipe (lr 0.1, momentum 0.9, wd 1e-4 on ndim>1) vs the reference model + torch.optim.SGD.

    lr 0.1 on random-init weights is far outside the linear regime (the loss falls 7.0 -> 3.5 in one step), so
    the fp32 rounding noise of the first backward pass (2 % of the gradient, see
    test_classifier_gradients_vs_fp64_oracle) becomes percents of the NEXT loss: the reference's own fp32 result
    is 0.06-0.17 away from the same step evaluated in fp64 (fixture loss_step1_fp64).  Step 0 is held to 1e-4;
    step 1 to 5x the reference's own distance from fp64: that distance is ONE realisation of the rounding, and this
    engine's realisations spread as widely -- round 4, scripts/diag_cls_step1.py, cls_64, |l1 - fp64| for four summation
    orders: fp32 matrix pipe 0.23 (stream-K plan) / 0.35 (whole tiles only), bf16 pipe with split operands 0.60 / 0.17
    (reference fp32: 0.17)."""
    from sacv_amd.classification import losses
    from sacv_amd.train import ClassificationTrainer
    meta = json.load(open(os.path.join(golden_dir, 'cls.json')))[name]
    B, R, nc = meta['B'], meta['R'], meta['num_classes']
    net = build(nc)
    tr = ClassificationTrainer(net, losses.__dict__['CELoss'](), lr=0.1, momentum=0.9, weight_decay=1e-4)
    x = torch.tensor(synth.images(11, B, R, R)).cuda()
    lab = torch.tensor(gi.labels_for(23, B, nc)).cuda()
    l0 = tr.finish_step(tr.step(x, lab))['loss']
    l1 = tr.finish_step(tr.step(x, lab))['loss']
    assert abs(l0 - meta['loss']) <= 1e-4 * max(1.0, abs(meta['loss']))
    yard = abs(meta['loss_step1'] - meta['loss_step1_fp64'])
    print(f'{name}: step-1 loss gpu {l1:.4f} reference fp32 {meta["loss_step1"]:.4f} fp64 {meta["loss_step1_fp64"]:.4f}')
    assert abs(l1 - meta['loss_step1_fp64']) <= 5.0 * yard + 1e-2 * abs(meta['loss_step1_fp64'])
    rel = [abs(float(p.detach().double().norm()) - meta['param_norm_after2'][k]) /
           (meta['param_norm_after2'][k] + 1e-12) for k, p in net.named_parameters()]
    print(f'{name}: parameter norms after 2 steps vs reference: median {np.median(rel):.2e} max {max(rel):.2e}')
    assert float(np.median(rel)) <= 1e-2
    assert int(net.state_dict()['conv1.layer.1.num_batches_tracked']) == 2


def test_classification_trainer_small_lr_vs_oracle():
    """The same trainer in the linear regime (lr 1e-4; at 2e-3 the loss already falls 7.04 -> 6.4 per step): loss after the step within 1e-3 of the CPU oracle's,
    the loss decrease within 5 %, and the parameter update element-wise where the two gradients agree."""
    from sacv_amd.classification import losses
    from sacv_amd.train import ClassificationTrainer
    from tests.test_cls_cpu import oracle_cls_run
    lr, B, R, nc = 1e-4, 4, 64, 1000
    _, loss_r, params, state, x, lab = oracle_cls_run('cls_64', dict(B=B, R=R, num_classes=nc))
    P = {k: p.detach().clone() for k, p in params.items()}
    G = {k: p.grad.clone() for k, p in params.items()}
    omodel.sgd_momentum_step(P, G, {}, lr, 0.9, 1e-4, first=True)
    state.update(P)
    with torch.no_grad():
        l1_r = float(omodel.ce_loss(omodel.resnet50_cls_forward(state, x, training=True), lab))
    net = build(nc)
    tr = ClassificationTrainer(net, losses.__dict__['CELoss'](), lr=lr, momentum=0.9, weight_decay=1e-4)
    l0 = float(tr.step(x.cuda(), lab.cuda())['loss'])
    checked = 0
    for k, p in net.named_parameters():
        g, g_gpu = G[k], net.arena.grad_views[k].cpu()
        mask = (g.abs() > 1e-2 * g.abs().max()) & ((g_gpu - g).abs() * 5 < g.abs())
        d = (p.detach().cpu() - P[k]).abs()[mask]
        checked += int(mask.sum())
        if d.numel():  # |update| = lr |g|; the two gradients agree to 20 % on these elements (+ weight decay, rounding)
            assert float((d - 0.21 * lr * g.abs()[mask]).max()) <= 1.3e-7, k  # + 1 ulp of |p| ~ 1
    assert checked > 1_000_000
    l1 = float(tr.step(x.cuda(), lab.cuda())['loss'])
    # Yardstick: the same oracle step in double precision.  4 x 64 x 64 leaves 16 samples per channel in layer4's batch
    # statistics: fp32 gradients of ANY implementation are ~2 % (rel. L2) from the fp64 ones here and the loss after the
    # step moves with them -- the fp32 CPU oracle's own distance from its fp64 run sets the scale (measured round 4:
    # GPU 7.416 / 7.444 with the image-reading / im2col stem, fp32 oracle 7.439).
    _, loss64, p64, st64, x64, _ = oracle_cls_run('cls_64', dict(B=B, R=R, num_classes=nc), dtype=torch.float64)
    P64 = {k: p.detach().clone() for k, p in p64.items()}
    omodel.sgd_momentum_step(P64, {k: p.grad.clone() for k, p in p64.items()}, {}, lr, 0.9, 1e-4, first=True)
    st64.update(P64)
    with torch.no_grad():
        l1_64 = float(omodel.ce_loss(omodel.resnet50_cls_forward(st64, x64, training=True), lab))
    d_ref = abs(l1_r - l1_64)
    print(f'small-lr step: loss {l0:.5f} -> {l1:.5f}; oracle {float(loss_r):.5f} -> {l1_r:.5f}; fp64 oracle '
          f'{float(loss64):.5f} -> {l1_64:.5f}')
    assert abs(l0 - float(loss_r)) <= 1e-4 * float(loss_r)
    assert abs(l1 - l1_64) <= max(1e-3 * l1_64, 2.5 * d_ref), (l1, l1_64, l1_r)
    assert abs((l0 - l1) - (float(loss64) - l1_64)) <= max(5e-2 * abs(float(loss64) - l1_64), 2.5 * d_ref)


def test_resume_from_a_torch_sgd_checkpoint_keeps_the_momentum():
    """a torch.optim.SGD state_dict has no per-parameter 'step'; the first optimizer step after
    load_state_dict must continue from the loaded momentum buffers (not overwrite them with the gradient), i.e. equal
    torch.optim.SGD's own next step from the same state -- and a fresh trainer's first step still equals torch's."""
    from sacv_amd.classification import losses
    from sacv_amd.train import ClassificationTrainer
    nc, B, R = 40, 4, 64
    net = build(nc)
    tr = ClassificationTrainer(net, losses.__dict__['CELoss'](), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    cpu_params = [[net.arena.module_params[n].detach().cpu().clone().contiguous().requires_grad_(True)
                   for n in g['param_names']] for g in tr.param_groups]
    opt = torch.optim.SGD([{'params': cpu_params[0], 'weight_decay': 0.0},
                           {'params': cpu_params[1], 'weight_decay': 1e-4}], lr=1e-2, momentum=0.9)
    # a torch SGD checkpoint with NONZERO momentum buffers: one torch step on synthetic gradients
    for gi_, plist in enumerate(cpu_params):
        for j, p in enumerate(plist):
            p.grad = torch.tensor(synth.normal(1000 * gi_ + j, tuple(p.shape))).reshape(p.shape) * 1e-2
    opt.step()
    sd = opt.state_dict()
    assert all('step' not in st for st in sd['state'].values())
    assert float(sd['state'][0]['momentum_buffer'].abs().max()) > 0
    with torch.no_grad():  # the model continues from the parameters torch produced
        for plist, g in zip(cpu_params, tr.param_groups):
            for p, n in zip(plist, g['param_names']):
                net.arena.module_params[n].copy_(p.detach())
    tr.load_state_dict(sd)
    x = torch.tensor(synth.images(11, B, R, R)).cuda()
    lab = torch.tensor(gi.labels_for(23, B, nc)).cuda()
    tr.step(x, lab)
    for plist, g in zip(cpu_params, tr.param_groups):
        for p, n in zip(plist, g['param_names']):
            p.grad = net.arena.grad_views[n].detach().cpu().clone().contiguous()
    opt.step()
    worst_p = worst_b = 0.0
    sd2, mine = opt.state_dict(), tr.state_dict()
    idx = 0
    for plist, g in zip(cpu_params, tr.param_groups):
        for p, n in zip(plist, g['param_names']):
            worst_p = max(worst_p, float((p.detach() - net.arena.module_params[n].detach().cpu()).abs().max()))
            worst_b = max(worst_b, float((sd2['state'][idx]['momentum_buffer'] -
                                          mine['state'][idx]['momentum_buffer'].cpu()).abs().max()))
            idx += 1
    assert worst_p <= 1.5e-7 and worst_b <= 1e-6, (worst_p, worst_b)
