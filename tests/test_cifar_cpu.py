"""BASELINE.json config #1 (ResNet18 CIFAR-100), CPU side: the general ResNet oracle (oracle/model.py
resnet_cls_forward) pinned against the reference resnet18cifar run by oracle/make_golden.py (section cifar), and
the mirrors' state_dict contracts."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel, synth
from tests import golden_inputs as gi


def oracle_cifar_run(meta, dtype=torch.float32):
    B, R, nc = meta['B'], meta['R'], meta['num_classes']
    sd = omodel.init_resnet_cls_state(0, nc)
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    state.update(params)
    x = torch.tensor(synth.images(31, B, R, R)).to(dtype)
    lab = torch.tensor(gi.labels_for(37, B, nc))
    logits = omodel.resnet_cls_forward(state, x, training=True)
    loss = omodel.ce_loss(logits, lab)
    loss.backward()
    return logits, loss, params, state, x, lab


@pytest.mark.parametrize('name', ['r18c_32', 'r18c_8'])
def test_oracle_resnet18cifar_matches_reference(golden_dir, name):
    meta = json.load(open(os.path.join(golden_dir, 'cifar.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'cifar.npz'))
    assert gi.labels_for(37, meta['B'], meta['num_classes']).tolist() == meta['labels']
    logits, loss, params, state, x, lab = oracle_cifar_run(meta)
    np.testing.assert_allclose(float(loss), meta['loss'], rtol=1e-5)
    np.testing.assert_allclose(logits.detach().numpy(), arrs[f'{name}_logits'], rtol=1e-4, atol=1e-5)
    for k, p in params.items():
        np.testing.assert_allclose(float(p.grad.double().norm()), meta['grad_norm'][k], rtol=2e-3, atol=1e-7,
                                   err_msg=k)
    for k in ('fc.bias', 'conv1.layer.0.weight'):
        np.testing.assert_allclose(params[k].grad.numpy(), arrs[f'{name}_grad_{k}'], rtol=5e-3, atol=2e-5)


@pytest.mark.parametrize('arch', ['resnet18cifar', 'resnet34cifar'])
def test_cifar_mirror_state_dict_contract(golden_dir, arch):
    from sacv_amd.classification import backbones
    c = json.load(open(os.path.join(golden_dir, 'cifar.json')))[arch + '_contract']
    m = backbones.__dict__[arch](num_classes=100)
    sd = m.state_dict()
    assert [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()] == c['keys']
    assert [[k, list(v.shape)] for k, v in m.named_parameters()] == c['params']
    assert sum(p.numel() for p in m.parameters()) == c['n_params']
    if arch == 'resnet18cifar':
        m.load_state_dict(omodel.init_resnet_cls_state(0, 100), strict=True)
    with pytest.raises(RuntimeError, match='MI355X'):
        m(torch.zeros(1, 3, 32, 32))
