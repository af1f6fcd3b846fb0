"""Oracle (test infrastructure): ViT MAE pre-training forward on torch-CPU, a functional restatement over a flat
{state_dict key: tensor} dictionary of
  VITMAEPretrainModelEncoder.forward / random_masking  simpleAICV/masked_image_modeling/models/vit_mae.py:160-215
  VITMAEPretrainModelDecoder.forward                   :343-372
  VITMAEPretrainModel.forward                          :411-416
  TransformerEncoderLayer / MultiHeadAttention / FeedForward   simpleAICV/classification/backbones/vit.py:58-160
  MSELoss                                              simpleAICV/masked_image_modeling/losses.py:11-23
  MAESelfSupervisedPretrainCollater labels             simpleAICV/masked_image_modeling/common.py:25-39
with the masking noise as an explicit input.  Pinned against the reference itself by tests/golden/mae.* (the reference
draws the noise with torch.rand after torch.manual_seed(s); the fixture stores those draws)."""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

from . import synth


def sincos_table(dim, grid, cls_token=True):
    """MAE's fixed 2-D sin-cos position table (vit_mae.py:100-158), same NumPy dtypes."""
    ax = np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(ax, ax), axis=0).reshape([2, 1, grid, grid])

    def axis(d, pos):
        om = np.arange(d // 2, dtype=np.float32)
        om /= d / 2.
        om = 1. / 10000**om
        o = np.einsum('m,d->md', pos.reshape(-1), om)
        return np.concatenate([np.sin(o), np.cos(o)], axis=1)

    t = np.concatenate([axis(dim // 2, g[0]), axis(dim // 2, g[1])], axis=1)
    if cls_token:
        t = np.concatenate([np.zeros([1, dim]), t], axis=0)
    return torch.from_numpy(t).float().unsqueeze(0)


def init_mae_state(seed, cfg):
    """cfg: dict(patch, image, E, Le, He, D, Ld, Hd, ratio=4).  Xavier-uniform Linear weights, zero biases, LayerNorm
    (1, 0), N(0, 0.02) tokens, sin-cos position tables; values from the portable generator, seeded per tensor name."""
    sd = {}

    def uni(key, *shape):
        return torch.from_numpy(synth.uniform(seed * 1000003 + zlib.crc32(key.encode()) + 17, shape)) * 2 - 1

    def linear(prefix, cin, cout, bias_noise=0.0):
        bound = math.sqrt(6.0 / (cin + cout))
        sd[prefix + '.weight'] = uni(prefix + '.w', cout, cin) * bound
        sd[prefix + '.bias'] = uni(prefix + '.b', cout) * bias_noise  # (reference: zeros; noise exercises the adds)

    def ln(prefix, c):
        sd[prefix + '.weight'] = torch.ones(c) + 0.1 * uni(prefix + '.g', c)
        sd[prefix + '.bias'] = 0.1 * uni(prefix + '.be', c)

    def blocks(prefix, depth, c, ratio):
        for i in range(depth):
            p = f'{prefix}.blocks.{i}'
            ln(p + '.norm1', c)
            linear(p + '.attn.qkv', c, 3 * c, 0.02)
            linear(p + '.attn.proj', c, c, 0.02)
            ln(p + '.norm2', c)
            linear(p + '.mlp.fc1', c, int(c * ratio), 0.02)
            linear(p + '.mlp.fc2', int(c * ratio), c, 0.02)

    P, S, E, D = cfg['patch'], cfg['image'], cfg['E'], cfg['D']
    ratio = cfg.get('ratio', 4)
    g = S // P
    sd['encoder.cls_token'] = torch.from_numpy(synth.normal(seed * 7 + 1, (1, 1, E), std=0.02))
    sd['encoder.pos_embed'] = sincos_table(E, g)
    k = 3 * P * P
    sd['encoder.patch_embed.proj.weight'] = (uni('pe.w', E, k) * math.sqrt(6.0 / (k + E))).view(E, 3, P, P)
    sd['encoder.patch_embed.proj.bias'] = uni('pe.b', E) * 0.02
    blocks('encoder', cfg['Le'], E, ratio)
    ln('encoder.norm', E)
    sd['decoder.mask_token'] = torch.from_numpy(synth.normal(seed * 7 + 2, (1, 1, D), std=0.02))
    sd['decoder.pos_embed'] = sincos_table(D, g)
    blocks('decoder', cfg['Ld'], D, ratio)
    ln('decoder.norm', D)
    linear('decoder.fc', D, P * P * 3, 0.02)
    linear('encoder_to_decoder', E, D, 0.02)
    return sd


def random_masking(noise, mask_ratio):
    b, n = noise.shape
    keep = int(n * (1 - mask_ratio))
    shuffle = torch.argsort(noise, dim=1)
    restore = torch.argsort(shuffle, dim=1)
    mask = torch.ones([b, n])
    mask[:, :keep] = 0
    mask = torch.gather(mask, dim=1, index=restore)
    return shuffle[:, :keep], mask, restore


def _block(sd, p, x, heads):
    b, n, c = x.shape
    h = F.layer_norm(x, (c,), sd[p + '.norm1.weight'], sd[p + '.norm1.bias'], 1e-6)
    qkv = F.linear(h, sd[p + '.attn.qkv.weight'], sd[p + '.attn.qkv.bias'])
    qkv = qkv.view(b, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = torch.unbind(qkv, dim=0)
    attn = (q @ k.transpose(-2, -1)) * (c // heads)**-0.5
    attn = attn.softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(b, n, c)
    x = x + F.linear(a, sd[p + '.attn.proj.weight'], sd[p + '.attn.proj.bias'])
    h = F.layer_norm(x, (c,), sd[p + '.norm2.weight'], sd[p + '.norm2.bias'], 1e-6)
    h = F.gelu(F.linear(h, sd[p + '.mlp.fc1.weight'], sd[p + '.mlp.fc1.bias']))
    return x + F.linear(h, sd[p + '.mlp.fc2.weight'], sd[p + '.mlp.fc2.bias'])


def mae_forward(sd, images, noise, cfg, mask_ratio=0.75):
    """-> (pred [B, n, P*P*3], mask [B, n], restore_ids)."""
    P = cfg['patch']
    x = F.conv2d(images, sd['encoder.patch_embed.proj.weight'], sd['encoder.patch_embed.proj.bias'], stride=P)
    x = x.flatten(2).transpose(1, 2)
    x = x + sd['encoder.pos_embed'][:, 1:, :]
    keep_ids, mask, restore = random_masking(noise, mask_ratio)
    x = torch.gather(x, dim=1, index=keep_ids.unsqueeze(-1).repeat(1, 1, x.shape[-1]))
    cls = (sd['encoder.cls_token'] + sd['encoder.pos_embed'][:, :1, :]).expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    for i in range(cfg['Le']):
        x = _block(sd, f'encoder.blocks.{i}', x, cfg['He'])
    x = F.layer_norm(x, (cfg['E'],), sd['encoder.norm.weight'], sd['encoder.norm.bias'], 1e-6)
    x = F.linear(x, sd['encoder_to_decoder.weight'], sd['encoder_to_decoder.bias'])
    mt = sd['decoder.mask_token'].repeat(x.shape[0], restore.shape[1] + 1 - x.shape[1], 1)
    x_ = torch.cat([x[:, 1:, :], mt], dim=1)
    x_ = torch.gather(x_, dim=1, index=restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
    x = torch.cat([x[:, :1, :], x_], dim=1) + sd['decoder.pos_embed']
    for i in range(cfg['Ld']):
        x = _block(sd, f'decoder.blocks.{i}', x, cfg['Hd'])
    x = F.layer_norm(x, (cfg['D'],), sd['decoder.norm.weight'], sd['decoder.norm.bias'], 1e-6)
    x = F.linear(x, sd['decoder.fc.weight'], sd['decoder.fc.bias'])
    return x[:, 1:, :], mask, restore


def patch_labels(images, patch, norm=True):
    b, _, h, w = images.shape
    g = h // patch
    x = images.reshape(b, 3, g, patch, g, patch)
    x = torch.einsum('nchpwq->nhwpqc', x).reshape(b, g * g, patch * patch * 3)
    if norm:
        mean = x.mean(dim=-1, keepdim=True)
        var = x.var(dim=-1, keepdim=True)
        x = (x - mean) / (var + 1e-4)**0.5
    return x


def mse_loss(pred, label, mask):
    loss = ((pred - label)**2).mean(dim=-1)
    return (loss * mask).sum() / (mask.sum() + 1e-4)
