"""End-to-end GPU parity of the RetinaNet-R50 training step against the CPU oracle (same weights,
same inputs): forward outputs, losses (1e-4, BASELINE.json), every parameter gradient, BN running
statistics, and the reference-generated fixtures at the full 800x1333 benchmark shape."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import losses as olosses, model as omodel, synth
from tests import golden_inputs as gi


def build_gpu_model(sd):
    from sacv_amd.detection import models
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    net.load_state_dict(sd)
    return net.cuda().train()


def oracle_step(sd, images, annots, blt, dtype=torch.float32):
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    state.update(params)
    outs = omodel.retinanet_forward(state, images.to(dtype), training=True)
    ld = olosses.RetinaLossOracle(box_loss_type=blt)(outs, annots.to(dtype))
    sum(ld.values()).backward()
    return outs, ld, params, state


@pytest.mark.parametrize('tag,blt,channels_last', [('TINY', 'SmoothL1', False), ('MID', 'GIoU', True)])
def test_forward_loss_backward_vs_oracle(tag, blt, channels_last):
    """Forward outputs and losses against the fp32 CPU oracle; parameter gradients against the oracle
    evaluated in fp64 ("exact"), with the fp32 CPU oracle's own distance to fp64 as the yardstick:
    with batch statistics over a few dozen pixels and ReLU masks on near-zero pre-activations the
    backward pass is ill-conditioned (the reference's fp32 CPU path itself is 1e-2 away from fp64),
    so the bar is "as close to exact as the reference arithmetic is", not a fixed epsilon."""
    from sacv_amd.detection import losses
    H, W = gi.SHAPES[tag]
    sd = omodel.init_retinanet_state(0)
    images = torch.tensor(synth.images(5, 2, H, W))
    annots = torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8))
    outs_r, ld_r, params_r, state_r = oracle_step(sd, images, annots, blt)
    _, ld_64, params_64, _ = oracle_step(sd, images, annots, blt, torch.float64)

    net = build_gpu_model(sd)
    crit = losses.__dict__['RetinaLoss'](box_loss_type=blt)
    x = images.cuda()
    if channels_last:  # the collater hands over channels_last-strided batches (common.py:262-268)
        x = x.contiguous(memory_format=torch.channels_last)
    outs = net(x)
    assert len(outs) == 2 and len(outs[0]) == 5 and len(outs[1]) == 5
    for grp, grp_r in zip(outs, outs_r):
        for o, r in zip(grp, grp_r):
            assert tuple(o.shape) == tuple(r.shape)
            assert float((o.detach().cpu() - r.detach()).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max()))
    ld = crit(outs, annots.cuda())
    for k in ('cls_loss', 'reg_loss'):
        assert abs(float(ld[k]) - float(ld_r[k])) <= 1e-4 * max(1.0, abs(float(ld_r[k]))), k
    loss = sum(ld.values())
    loss.backward()
    # global relative L2 over all 195 gradient tensors, and per-tensor relative L2
    num_g = num_c = den = 0.0
    per_gpu, per_cpu = [], []
    for name, p in net.named_parameters():
        exact = params_64[name].grad
        dg = (p.grad.double().cpu() - exact)
        dc = (params_r[name].grad.double() - exact)
        n2 = float(exact.norm())**2
        num_g += float(dg.norm())**2
        num_c += float(dc.norm())**2
        den += n2
        per_gpu.append(float(dg.norm()) / (n2**0.5 + 1e-30))
        per_cpu.append(float(dc.norm()) / (n2**0.5 + 1e-30))
    e_gpu, e_cpu = (num_g / den)**0.5, (num_c / den)**0.5
    print(f'{tag}: gradient rel-L2 error vs fp64: gpu {e_gpu:.2e} (median tensor {np.median(per_gpu):.2e}), '
          f'reference-arithmetic cpu fp32 {e_cpu:.2e} (median tensor {np.median(per_cpu):.2e})')
    assert e_gpu <= 2.0 * e_cpu + 1e-3
    assert float(np.median(per_gpu)) <= 2.0 * float(np.median(per_cpu)) + 1e-3
    sdg = net.state_dict()
    for k in ('backbone.conv1.layer.1.running_mean', 'backbone.layer3.5.conv3.layer.1.running_var',
              'backbone.layer4.2.conv3.layer.1.running_mean'):
        np.testing.assert_allclose(sdg[k].cpu().numpy(), state_r[k].numpy(), rtol=1e-4, atol=5e-6)
    assert int(sdg['backbone.conv1.layer.1.num_batches_tracked']) == 1


def test_full_size_against_reference_fixture(golden_dir):
    """800x1333 (the benchmark shape): loss, output samples and per-parameter gradient norms produced
    by the REFERENCE model + RetinaLoss on CPU (oracle/make_golden.py, e2e_s1)."""
    from sacv_amd.detection import losses
    meta = json.load(open(os.path.join(golden_dir, 'e2e_s1.json')))['retina_S1']
    arrs = np.load(os.path.join(golden_dir, 'e2e_s1.npz'))
    H, W = gi.S1
    net = build_gpu_model(omodel.init_retinanet_state(0))
    crit = losses.__dict__['RetinaLoss']()
    outs = net(torch.tensor(synth.images(5, 2, H, W)).cuda())
    ld = crit(outs, torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8)).cuda())
    for k, v in meta['loss'].items():
        assert abs(float(ld[k]) - v) <= 1e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
    for gi_, grp in enumerate(outs):
        for l, o in enumerate(grp):
            d = gi.digest(o.detach().cpu().numpy(), 256)
            np.testing.assert_allclose(d['samples'], arrs[f'retina_S1_out{gi_}_l{l}_samples'], rtol=2e-4, atol=2e-6)
    sum(ld.values()).backward()
    # gradient norms of all 195 tensors vs the reference's fp32 CPU run.  Gates = 3 x what this run prints (round 4:
    # median 2.2e-4, max 3.6e-3 -- the stem's BatchNorm parameters, behind 53 batch-statistics layers): a regression
    # that triples the distance to the reference fails here
    rel = []
    for name, p in net.named_parameters():
        n = float(p.grad.double().norm())
        rel.append(abs(n - meta['grad_norm'][name]) / (meta['grad_norm'][name] + 1e-12))
        assert rel[-1] <= 1.2e-2, (name, rel[-1])
    assert float(np.median(rel)) <= 7e-4, float(np.median(rel))
    print('S1 grad-norm rel err: median %.2e max %.2e' % (float(np.median(rel)), max(rel)))


def test_trainer_step_matches_cpu_adamw():
    """Fused trainer steps (forward, loss, backward, AdamW over the flat arena) vs the oracle model with
    torch.optim.AdamW in the reference's two parameter groups (tools/utils.py:484-509).

    Adam normalises every element's update to ~lr whatever the gradient's scale, so elements whose
    gradient is rounding noise move by +-lr at random in BOTH implementations; the parameter check is
    therefore restricted to elements with a significant gradient (|g| > 1 % of the tensor's max), where
    the update must agree to a small fraction of lr.  (The AdamW kernel itself is checked element-exact
    against torch in test_adamw_matches_torch.)"""
    from sacv_amd.detection import losses
    from sacv_amd.train import DetectionTrainer
    H, W = gi.MID
    lr = 1e-4
    sd = omodel.init_retinanet_state(0)
    net = build_gpu_model(sd)
    trainer = DetectionTrainer(net, losses.__dict__['RetinaLoss'](), lr=lr, weight_decay=1e-3)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = {k: v.clone() for k, v in sd.items()}
    state.update(params)
    opt = torch.optim.AdamW([{'params': [p for p in params.values() if p.ndim == 1], 'weight_decay': 0.0},
                             {'params': [p for p in params.values() if p.ndim != 1], 'weight_decay': 1e-3}],
                            lr=lr)
    ocrit = olosses.RetinaLossOracle()
    for it in range(2):
        images = torch.tensor(synth.images(50 + it, 2, H, W))
        annots = torch.tensor(synth.annotations(60 + it, 2, H, W, boxes_per_image=8))
        got = trainer.step(images.cuda(), annots.cuda())
        ld = ocrit(omodel.retinanet_forward(state, images, training=True), annots)
        opt.zero_grad()
        sum(ld.values()).backward()
        if it == 0:
            grads0 = {k: p.grad.clone() for k, p in params.items()}
        opt.step()
        # step 0 starts from identical weights: 1e-4; step 1 starts from weights that differ by up to 2*lr on
        # the noise-gradient elements, which moves the loss itself: 2e-3
        tol = 1e-4 if it == 0 else 2e-3
        for k in ('cls_loss', 'reg_loss'):
            assert abs(float(got[k]) - float(ld[k])) <= tol * max(1.0, abs(float(ld[k]))), (it, k)
        if it == 0:
            checked = 0
            for name, p in net.named_parameters():
                g = grads0[name]
                g_gpu = net.arena.grad_views[name].cpu()
                # elements whose gradient is significant AND on which the two gradient computations agree
                # well (the backward is ill-conditioned at this size, see test_forward_loss_backward_vs_oracle)
                mask = (g.abs() > 1e-2 * g.abs().max()) & ((g_gpu - g).abs() * 5 < g.abs())
                d = (p.detach().cpu() - params[name].detach()).abs()[mask]
                checked += int(mask.sum())
                if d.numel():
                    assert float(d.max()) <= 0.1 * lr, (name, float(d.max()))
            assert checked > 1_000_000
    fin = trainer.finish_step(got)
    assert set(fin) == {'cls_loss', 'reg_loss', 'loss', 'optimizer_steps', 'skipped'}
    assert abs(fin['loss'] - fin['cls_loss'] - fin['reg_loss']) < 1e-6 and fin['optimizer_steps'] == 2 and not fin['skipped']


def test_trainer_skips_zero_loss_batches_like_the_reference():
    """tools/scripts.py:918-974: a batch whose loss is 0 (no positive anchor: RetinaLoss returns zeros) or inf/nan is
    skipped -- no parameter, moment or step-count update.  The next regular batch must then be the optimizer's
    FIRST step (bias corrections of t = 1), exactly as if the skipped batch had never been seen (BatchNorm running
    statistics excepted: the reference also runs the forward pass of a batch it skips)."""
    from sacv_amd.detection import losses
    from sacv_amd.train import DetectionTrainer
    H, W = gi.TINY
    sd = omodel.init_retinanet_state(0)
    net = build_gpu_model(sd)
    tr = DetectionTrainer(net, losses.__dict__['RetinaLoss'](), lr=1e-4, weight_decay=1e-3)
    images = torch.tensor(synth.images(5, 2, H, W)).cuda()
    empty = torch.tensor(synth.annotations(1, 2, H, W, boxes_per_image=0)).cuda()
    boxes = torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8)).cuda()
    p0 = net.arena.params.clone()
    fin = tr.finish_step(tr.step(images, empty))
    assert fin['skipped'] and fin['optimizer_steps'] == 0 and fin['loss'] == 0.0
    assert torch.equal(net.arena.params, p0)
    assert float(tr.exp_avg.abs().max()) == 0.0 and float(tr.exp_avg_sq.abs().max()) == 0.0
    # reference run of the same first real step with torch.optim.AdamW state at t = 1
    fin = tr.finish_step(tr.step(images, boxes))
    assert not fin['skipped'] and fin['optimizer_steps'] == 1
    g = net.arena.grads.clone()
    m = 0.1 * g
    v = 0.001 * g * g
    upd = (m / 0.1) / ((v.sqrt() / (0.001**0.5)) + 1e-8)  # t = 1: bc1 = 0.1, bc2 = 0.001
    wd = torch.zeros_like(p0)
    wd[net.arena.split:] = 1e-3
    want = p0 * (1 - 1e-4 * wd) - 1e-4 * upd
    torch.testing.assert_close(net.arena.params, want, rtol=1e-5, atol=2e-7)
    # a nan loss term skips as well
    from sacv_amd import ops
    flag = ops.loss_skip_flag(torch.tensor([0.5, float('nan')]).cuda())
    assert float(flag) == 1.0
    assert float(ops.loss_skip_flag(torch.tensor([0.5, 0.25]).cuda())) == 0.0
    assert float(ops.loss_skip_flag(torch.tensor([float('inf'), 0.25]).cuda())) == 1.0


@pytest.mark.parametrize('name,crit', [('resnet18_retinanet', 'RetinaLoss'), ('resnet101_fcos', 'FCOSLoss')])
def test_trainer_step_on_the_other_backbones(name, crit):
    """DetectionTrainer.step on a BasicBlock trunk (ResNet18) and a deep Bottleneck trunk (ResNet101; FCOS head): the
    gradient arena after the fused step equals what the class-level path (model(images) -> criterion -> backward, the
    path pinned to the reference's fixtures in test_gpu_pinned.py) leaves in a second model with the same weights, the
    loss values agree, the two AdamW groups hold every parameter, and the second step runs (stage hooks, buckets)."""
    from sacv_amd.detection import models, losses
    from sacv_amd.train import DetectionTrainer
    H, W = gi.MID
    images = torch.tensor(synth.images(31, 2, H, W)).cuda()
    annots = torch.tensor(synth.annotations(32, 2, H, W, boxes_per_image=8)).cuda()
    net = models.__dict__[name]().cuda().train()
    weights = {k: v.clone() for k, v in net.state_dict().items()}
    tr = DetectionTrainer(net, losses.__dict__[crit](), lr=1e-4, weight_decay=1e-3)
    assert sum(len(g['param_names']) for g in tr.param_groups) == len(list(net.named_parameters()))
    got = tr.finish_step(tr.step(images, annots))
    fast = net.arena.grads.clone()
    ref = models.__dict__[name]().cuda().train()
    ref.load_state_dict(weights)
    ld = losses.__dict__[crit]()(ref(images), annots)
    sum(ld.values()).backward()
    for k, v in ld.items():
        assert abs(got[k] - float(v)) <= 1e-6 * max(1.0, abs(float(v))), (k, got[k], float(v))
    d = float((fast - ref.arena.grads).abs().max())
    assert d <= 1e-6 * float(ref.arena.grads.abs().max()), d
    again = tr.finish_step(tr.step(images, annots))
    assert again['optimizer_steps'] == 2 and not again['skipped'] and np.isfinite(again['loss'])
