"""ResNet50 ImageNet classification row, CPU side: the oracle restatement (oracle/model.py
resnet50_cls_forward / ce_loss / sgd_momentum_step) pinned against the reference model run by
oracle/make_golden.py (section cls), and the mirror's state_dict contract."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import model as omodel, synth
from tests import golden_inputs as gi


def oracle_cls_run(name, meta, dtype=torch.float32, steps=1):
    B, R, nc = meta['B'], meta['R'], meta['num_classes']
    sd = omodel.init_resnet50_cls_state(0, nc)
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    state.update(params)
    x = torch.tensor(synth.images(11, B, R, R)).to(dtype)
    lab = torch.tensor(gi.labels_for(23, B, nc))
    logits = omodel.resnet50_cls_forward(state, x, training=True)
    loss = omodel.ce_loss(logits, lab)
    loss.backward()
    return logits, loss, params, state, x, lab


def test_labels_fixture(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, 'cls.json')))
    for name in ('cls_64', 'cls_224', 'cls_96_nc40'):
        m = meta[name]
        assert gi.labels_for(23, m['B'], m['num_classes']).tolist() == m['labels']


@pytest.mark.parametrize('name', ['cls_64', 'cls_96_nc40'])
def test_oracle_classifier_matches_reference(golden_dir, name):
    meta = json.load(open(os.path.join(golden_dir, 'cls.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'cls.npz'))
    logits, loss, params, state, x, lab = oracle_cls_run(name, meta)
    np.testing.assert_allclose(float(loss), meta['loss'], rtol=1e-5)
    np.testing.assert_allclose(logits.detach().numpy(), arrs[f'{name}_logits'], rtol=1e-4, atol=1e-5)
    for k, p in params.items():
        np.testing.assert_allclose(float(p.grad.double().norm()), meta['grad_norm'][k], rtol=2e-3, atol=1e-7,
                                   err_msg=k)
    for k in ('fc.bias', 'conv1.layer.1.weight'):
        np.testing.assert_allclose(params[k].grad.numpy(), arrs[f'{name}_grad_{k}'], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(state['conv1.layer.1.running_var'].numpy(),
                               arrs[f'{name}_buf_conv1.layer.1.running_var'], rtol=1e-5, atol=1e-7)
    # the recipe's SGD: one oracle step, then the loss the reference saw at its second step
    P = {k: p.detach().clone() for k, p in params.items()}
    G = {k: p.grad.clone() for k, p in params.items()}
    bufs = {}
    omodel.sgd_momentum_step(P, G, bufs, 0.1, 0.9, 1e-4, first=True)
    state.update(P)
    with torch.no_grad():
        loss1 = omodel.ce_loss(omodel.resnet50_cls_forward(state, x, training=True), lab)
    np.testing.assert_allclose(float(loss1), meta['loss_step1'], rtol=2e-3)


def test_classifier_state_dict_contract(golden_dir):
    from sacv_amd.classification import backbones, losses
    c = json.load(open(os.path.join(golden_dir, 'cls.json')))['contract']
    m = backbones.__dict__['resnet50'](num_classes=1000)
    sd = m.state_dict()
    assert [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()] == c['keys']
    assert [[k, list(p.shape)] for k, p in m.named_parameters()] == c['params']
    assert sum(p.numel() for p in m.parameters()) == c['n_params'] == 25557032
    # oracle init loads strictly; no CPU execution path
    m.load_state_dict(omodel.init_resnet50_cls_state(0, 1000), strict=True)
    with pytest.raises(RuntimeError, match='MI355X'):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match='MI355X'):
        losses.__dict__['CELoss']()(torch.zeros(2, 10), torch.zeros(2, dtype=torch.long))
