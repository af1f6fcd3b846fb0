"""Pins oracle/model.py (torch-CPU functional RetinaNet/FCOS) + oracle losses end to end
against the reference model run by oracle/make_golden.py (same weights, same inputs)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import losses as olosses, model as omodel, synth
from tests import golden_inputs as gi


def run_oracle_e2e(H, W, fcos, box_loss_type='SmoothL1', num_classes=80, backbone='resnet50backbone'):
    sd = omodel.init_retinanet_state(0, num_classes=num_classes, fcos=fcos, backbone=backbone)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = dict(sd)
    state.update(params)
    x = torch.tensor(synth.images(5, 2, H, W))
    ann = torch.tensor(synth.annotations(3, 2, H, W, boxes_per_image=8))
    if num_classes != 80:
        ann[..., 4] = torch.where(ann[..., 4] >= 0, ann[..., 4] % num_classes, ann[..., 4])
    if fcos:
        outs = omodel.fcos_forward(state, x, training=True)
        crit = olosses.FCOSLossOracle()
    else:
        outs = omodel.retinanet_forward(state, x, training=True, num_classes=num_classes)
        crit = olosses.RetinaLossOracle(box_loss_type=box_loss_type)
    ld = crit(outs, ann)
    sum(ld.values()).backward()
    return outs, ld, params, state


CASES = [('retina_TINY', gi.TINY, False, 'SmoothL1'), ('retina_MID_giou', gi.MID, False, 'GIoU'),
         ('fcos_TINY', gi.TINY, True, None), ('fcos_MID', gi.MID, True, None)]


@pytest.mark.parametrize('name,shape,fcos,blt', CASES)
def test_oracle_model_matches_reference(golden_dir, name, shape, fcos, blt):
    meta = json.load(open(os.path.join(golden_dir, 'e2e.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'e2e.npz'))
    outs, ld, params, state = run_oracle_e2e(shape[0], shape[1], fcos, blt)
    for k, v in ld.items():
        np.testing.assert_allclose(float(v), meta['loss'][k], rtol=1e-5)
    for gi_, grp in enumerate(outs):
        for l, o in enumerate(grp):
            d = gi.digest(o.detach().numpy(), 256)
            np.testing.assert_allclose(d['samples'], arrs[f'{name}_out{gi_}_l{l}_samples'], rtol=1e-4, atol=1e-6)
    for k, p in params.items():
        np.testing.assert_allclose(float(p.grad.double().norm()), meta['grad_norm'][k], rtol=2e-3, atol=1e-7,
                                   err_msg=k)
    for k in ('backbone.conv1.layer.1.running_mean', 'backbone.conv1.layer.1.running_var'):
        np.testing.assert_allclose(state[k].numpy(), arrs[f'{name}_buf_{k}'], rtol=1e-5, atol=1e-7)


def test_oracle_model_on_the_voc_canvas(golden_dir):
    """VOC recipe shape (640 x 640, 20 classes; fixture section e2e_other): the restatement is not tied to 80 classes
    or to the 800 x 1333 aspect."""
    meta = json.load(open(os.path.join(golden_dir, 'e2e_other.json')))['retina_V640']
    arrs = np.load(os.path.join(golden_dir, 'e2e_other.npz'))
    outs, ld, params, state = run_oracle_e2e(meta['H'], meta['W'], False, num_classes=meta['num_classes'])
    assert tuple(outs[0][0].shape[1:]) == (80, 80, 9, 20)
    for k, v in ld.items():
        np.testing.assert_allclose(float(v), meta['loss'][k], rtol=1e-5)
    for gi_, grp in enumerate(outs):
        for l, o in enumerate(grp):
            d = gi.digest(o.detach().numpy(), 256)
            np.testing.assert_allclose(d['samples'], arrs[f'retina_V640_out{gi_}_l{l}_samples'], rtol=1e-4, atol=1e-6)
    for k, p in params.items():
        np.testing.assert_allclose(float(p.grad.double().norm()), meta['grad_norm'][k], rtol=2e-3, atol=1e-7,
                                   err_msg=k)


@pytest.mark.parametrize('name', ['retina_R18', 'fcos_R34', 'retina_R101'])
def test_oracle_model_on_the_other_backbones(golden_dir, name):
    """BasicBlock ResNet18 / 34 and the deeper Bottleneck backbones of the reference's factories (fixture section
    e2e_backbones, produced by the reference's own resnet18_retinanet / resnet34_fcos / resnet101_retinanet; the
    ResNet152 case runs in the GPU suite only, to keep this one short)."""
    meta = json.load(open(os.path.join(golden_dir, 'e2e_backbones.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'e2e_backbones.npz'))
    fcos = name.startswith('fcos')
    outs, ld, params, state = run_oracle_e2e(meta['H'], meta['W'], fcos, backbone=meta['resnet'] + 'backbone')
    for k, v in ld.items():
        np.testing.assert_allclose(float(v), meta['loss'][k], rtol=1e-5)
    for gi_, grp in enumerate(outs):
        for l, o in enumerate(grp):
            d = gi.digest(o.detach().numpy(), 256)
            np.testing.assert_allclose(d['samples'], arrs[f'{name}_out{gi_}_l{l}_samples'], rtol=1e-4, atol=1e-6)
    for k, p in params.items():
        np.testing.assert_allclose(float(p.grad.double().norm()), meta['grad_norm'][k], rtol=2e-3, atol=1e-7,
                                   err_msg=k)
