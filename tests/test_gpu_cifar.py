"""BASELINE.json config #1 on the GPU: resnet18cifar (BasicBlock stages, 3x3/s1 stem, no max pool) at the CIFAR-100
batch shape (32 x 3 x 32 x 32) against the reference fixtures (tests/golden/cifar.*) and the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth
from tests import golden_inputs as gi
from tests.test_cifar_cpu import oracle_cifar_run


@pytest.mark.parametrize('name', ['r18c_32', 'r18c_8'])
def test_resnet18cifar_step_against_reference_fixture(golden_dir, name):
    from sacv_amd.classification import backbones, losses
    from sacv_amd.train import ClassificationTrainer
    meta = json.load(open(os.path.join(golden_dir, 'cifar.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'cifar.npz'))
    B, R, nc = meta['B'], meta['R'], meta['num_classes']
    net = backbones.__dict__['resnet18cifar'](num_classes=nc)
    net.load_state_dict(omodel.init_resnet_cls_state(0, nc), strict=True)
    net = net.cuda().train()
    crit = losses.__dict__['CELoss']()
    x = torch.tensor(synth.images(31, B, R, R)).cuda()
    lab = torch.tensor(gi.labels_for(37, B, nc)).cuda()
    logits = net(x)
    loss = crit(logits, lab)
    assert abs(float(loss) - meta['loss']) <= 1e-4 * max(1.0, meta['loss'])
    ref = arrs[f'{name}_logits']
    assert float(np.abs(logits.detach().cpu().numpy() - ref).max()) <= 3e-4 * max(1.0, float(np.abs(ref).max()))
    loss.backward()
    # gradients: as close to fp64 as the reference's fp32 arithmetic is (BatchNorm over few pixels at layer4: 4x4)
    _, _, p64, _, _, _ = oracle_cifar_run(meta, torch.float64)
    _, _, p32, _, _, _ = oracle_cifar_run(meta)
    num_g = num_c = den = 0.0
    for k, p in net.named_parameters():
        exact = p64[k].grad
        num_g += float((p.grad.double().cpu() - exact).norm())**2
        num_c += float((p32[k].grad.double() - exact).norm())**2
        den += float(exact.norm())**2
    e_gpu, e_cpu = (num_g / den)**0.5, (num_c / den)**0.5
    print(f'{name}: gradient rel-L2 vs fp64: gpu {e_gpu:.2e}, reference-arithmetic cpu fp32 {e_cpu:.2e}')
    assert e_gpu <= 2.0 * e_cpu + 1e-3
    rel = [abs(float(p.grad.double().norm()) - meta['grad_norm'][k]) / (meta['grad_norm'][k] + 1e-12)
           for k, p in net.named_parameters()]
    assert float(np.median(rel)) <= 1e-2, float(np.median(rel))
    for k in ('fc.bias', 'conv1.layer.0.weight', 'layer2.0.downsample_conv.layer.0.weight'):
        want = arrs[f'{name}_grad_{k}'].astype(np.float64)
        got = dict(net.named_parameters())[k].grad.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(got - want) <= 3e-2 * np.linalg.norm(want) + 1e-7, k
    np.testing.assert_allclose(net.state_dict()['conv1.layer.1.running_var'].cpu().numpy(),
                               arrs[f'{name}_buf_conv1.layer.1.running_var'], rtol=1e-4, atol=1e-6)
    # two trainer steps of the fixture's SGD (lr 0.01, momentum 0.9, wd 5e-4 on ndim > 1)
    net2 = backbones.__dict__['resnet18cifar'](num_classes=nc)
    net2.load_state_dict(omodel.init_resnet_cls_state(0, nc), strict=True)
    net2 = net2.cuda().train()
    tr = ClassificationTrainer(net2, crit, lr=0.01, momentum=0.9, weight_decay=5e-4)
    l0 = tr.finish_step(tr.step(x, lab))['loss']
    l1 = tr.finish_step(tr.step(x, lab))
    assert abs(l0 - meta['loss']) <= 1e-4 * meta['loss']
    assert abs(l1['loss'] - meta['loss_step1']) <= 2e-2 * meta['loss_step1'], (l1, meta['loss_step1'])
    assert l1['optimizer_steps'] == 2
    for k, p in net2.named_parameters():
        want = meta['param_norm_after2'][k]
        # A parameter that starts at 0 (BatchNorm biases, fc.bias) IS its gradient after two steps, and this batch of 8 pins
        # gradients to ~1e-2 in fp32 arithmetic (the median gate above; 3e-2 per tensor): those get the gradient gate.
        # Measured round 4 with the bf16-pipe kernels: 2.3e-3 (conv1's bias), 6.4e-3 (layer1.1.conv1's).
        zero_init = k.endswith('layer.1.bias') or k == 'fc.bias'
        assert abs(float(p.detach().double().norm()) - want) <= (2e-2 if zero_init else 2e-3) * want + 2e-5, k
    net2.eval()  # the fixture's eval logits: after both optimizer steps, BN running statistics of two forwards
    with torch.no_grad():
        ev = net2(x).cpu().numpy()
    ref = arrs[f'{name}_eval_logits']
    assert float(np.abs(ev - ref).max()) <= 2e-2 * max(1.0, float(np.abs(ref).max()))
