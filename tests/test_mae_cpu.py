"""SURVEY.md 8f-4 (ViT MAE), CPU side: the oracle restatement (oracle/mae.py) pinned against the reference model,
loss and collater run by oracle/make_golden.py (section mae), and the mirror's state_dict contract / position tables."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import mae as omae, synth


def mae_inputs(cfg, B):
    S = cfg['image']
    hwc = np.transpose(synth.images(71, B, S, S), (0, 2, 3, 1)) * 4.0 - 2.0
    images = torch.from_numpy(np.array(hwc, dtype=np.float32)).permute(0, 3, 1, 2)  # channels_last strides, as the collater
    return images


@pytest.mark.parametrize('name', ['tiny', 'base_d2', 'large_d2'])
def test_oracle_mae_matches_reference(golden_dir, name):
    meta = json.load(open(os.path.join(golden_dir, 'mae.json')))[name]
    arrs = np.load(os.path.join(golden_dir, 'mae.npz'))
    cfg, B = meta['cfg'], meta['B']
    images = mae_inputs(cfg, B)
    labels = omae.patch_labels(images, cfg['patch'], norm=True)
    assert hashlib.sha256(np.ascontiguousarray(labels.numpy()).tobytes()).hexdigest() == meta['labels_sha']
    noise = torch.tensor(arrs[f'{name}_noise'])
    sd = omae.init_mae_state(0, cfg)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items() if 'pos_embed' not in k}
    state = dict(sd)
    state.update(P)
    pred, mask, restore = omae.mae_forward(state, images, noise, cfg)
    assert np.array_equal(mask.numpy(), arrs[f'{name}_mask'])          # bit-exact masking
    assert np.array_equal(restore.numpy(), arrs[f'{name}_restore'])
    loss = omae.mse_loss(pred, labels, mask)
    np.testing.assert_allclose(float(loss), meta['loss_fp32'], rtol=2e-5)
    np.testing.assert_allclose(pred.detach().numpy().reshape(-1)[::211], arrs[f'{name}_pred_fp32'], rtol=2e-3, atol=2e-4)
    loss.backward()
    for k, p in P.items():
        np.testing.assert_allclose(float(p.grad.double().norm()), meta['grad_norm_fp32'][k], rtol=5e-3, atol=1e-7,
                                   err_msg=k)
    np.testing.assert_allclose(P['encoder.cls_token'].grad.numpy(), arrs[f'{name}_grad_encoder.cls_token'], rtol=5e-3,
                               atol=1e-6)


def test_mae_mirror_contract_and_position_tables(golden_dir):
    from sacv_amd.masked_image_modeling import models
    meta = json.load(open(os.path.join(golden_dir, 'mae.json')))
    c = meta['contract_tiny']
    cfg = meta['tiny']['cfg']
    m = models.VITMAEPretrainModel(patch_size=cfg['patch'], image_size=cfg['image'], mask_ratio=0.75,
                                   encoder_embedding_planes=cfg['E'], encoder_block_nums=cfg['Le'],
                                   encoder_head_nums=cfg['He'], decoder_embedding_planes=cfg['D'],
                                   decoder_block_nums=cfg['Ld'], decoder_head_nums=cfg['Hd'])
    assert [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in m.state_dict().items()] == c['keys']
    assert [[k, list(v.shape), bool(v.requires_grad)] for k, v in m.named_parameters()] == c['params']
    m.load_state_dict(omae.init_mae_state(0, cfg), strict=True)
    full = models.__dict__['vit_base_patch16_224_mae_pretrain_model']()
    cb = meta['contract_base']
    assert [[k, list(v.shape)] for k, v in full.state_dict().items()] == cb['keys']
    assert sum(p.numel() for p in full.parameters()) == cb['n_params']
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(full.encoder.pos_embed.detach().numpy()) == cb['pos_embed_sha']       # bit-identical sin-cos tables
    assert sha(full.decoder.pos_embed.detach().numpy()) == cb['dec_pos_embed_sha']
    assert torch.equal(omae.sincos_table(768, 14), full.encoder.pos_embed.detach())
    with pytest.raises(RuntimeError, match='MI355X'):
        full(torch.zeros(1, 3, 224, 224))
