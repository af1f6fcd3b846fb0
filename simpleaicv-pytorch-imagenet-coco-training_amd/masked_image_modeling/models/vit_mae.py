"""ViT MAE pre-training models with the reference's constructors, module tree and state_dict keys
(simpleAICV/masked_image_modeling/models/vit_mae.py:25-520), executed by the HIP engine: bf16 matrix-core GEMMs for every
Linear (patch embedding as a GEMM over the visible patches only), fp32 residual stream, LayerNorm / attention / GELU /
masked-MSE kernels of csrc/vit.hip.  The torch.nn modules are parameter containers; there is no CPU / eager path."""
import numpy as np
import torch
import torch.nn as nn

from ...detection.models._arena import ParamArena, deepcopy_without_runtime, getstate_without_runtime, refuse_sync_batchnorm
from ...mae_engine import MAEEngine

__all__ = ['vit_base_patch16_224_mae_pretrain_model', 'vit_large_patch16_224_mae_pretrain_model',
           'vit_huge_patch14_224_mae_pretrain_model', 'VITMAEPretrainModel']


def sincos_position_encoding(dim, grid_size, cls_token=True):
    """2-D sin-cos position table of MAE (get_2d_sincos_position_encoding_init, vit_mae.py:100-158): half of the
    channels encode one grid axis, half the other (the reference's meshgrid puts w first), each as sin | cos of
    pos / 10000^(2i/d).  Same NumPy dtypes as the reference so the table is bit-identical."""
    axis = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(axis, axis), axis=0).reshape([2, 1, grid_size, grid_size])

    def one_axis(d, pos):
        omega = np.arange(d // 2, dtype=np.float32)
        omega /= d / 2.
        omega = 1. / 10000**omega
        out = np.einsum('m,d->md', pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    table = np.concatenate([one_axis(dim // 2, grid[0]), one_axis(dim // 2, grid[1])], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros([1, dim]), table], axis=0)
    return table


def _block(dim, heads, ratio):
    blk = nn.Module()
    blk.norm1 = nn.LayerNorm(dim, eps=1e-6)
    blk.attn = nn.Module()
    blk.attn.qkv = nn.Linear(dim, dim * 3)
    blk.attn.proj = nn.Linear(dim, dim)
    blk.norm2 = nn.LayerNorm(dim, eps=1e-6)
    blk.mlp = nn.Module()
    blk.mlp.fc1 = nn.Linear(dim, int(dim * ratio))
    blk.mlp.fc2 = nn.Linear(int(dim * ratio), dim)
    return blk


def _init_linear_ln(module):
    for m in module.modules():
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


class _MAEFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, model, images, *params):
        pred, mask = model._engine.forward(images, noise=model._next_noise, keep=True)
        model._next_noise = None
        ctx.model = model
        b = images.shape[0]
        out = pred.view(b, -1, pred.shape[-1])[:, 1:, :]  # the cls row is dropped (vit_mae.py:369-370)
        ctx.rows_shape = pred.shape
        model._last_rows = pred
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, dout, _dmask):
        model = ctx.model
        r, k = ctx.rows_shape
        b = dout.shape[0]
        dpred = torch.zeros((b, r // b, k), device=dout.device, dtype=torch.bfloat16)
        dpred[:, 1:, :] = dout.to(torch.bfloat16)
        model._engine.backward(dpred.view(r, k))
        return (None, None) + tuple(model._arena.grad_views[name] for name, _ in model._param_list)


class VITMAEPretrainModel(nn.Module):
    __deepcopy__ = deepcopy_without_runtime
    __getstate__ = getstate_without_runtime

    def __init__(self, patch_size=16, image_size=224, mask_ratio=0.75, encoder_embedding_planes=768,
                 encoder_block_nums=12, encoder_head_nums=12, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                 decoder_embedding_planes=384, decoder_block_nums=4, decoder_head_nums=6, decoder_feedforward_ratio=4,
                 decoder_dropout_prob=0., use_gradient_checkpoint=False):
        super().__init__()
        assert image_size % patch_size == 0
        if patch_size % 8 != 0:
            # sacv_mae_patchify moves 8 consecutive pixels of a patch row per thread and the patch-embed GEMM needs
            # K = 3 * P * P in 16-byte bf16 groups: P = 14 (vit_huge_patch14) gives K = 588, which is neither
            raise NotImplementedError(f'patch_size {patch_size}: the MI355X patch-embed path needs patch_size % 8 == 0 '
                                      '(vit_base / vit_large patch16 are supported; vit_huge_patch14 is not)')
        if encoder_dropout_prob != 0. or decoder_dropout_prob != 0.:
            raise NotImplementedError('dropout > 0 (every MAE recipe of the reference uses 0)')
        if use_gradient_checkpoint:
            raise NotImplementedError('use_gradient_checkpoint: activations are kept in HBM (288 GB) instead')
        self.patch_size, self.image_size, self.mask_ratio = patch_size, image_size, mask_ratio
        self.cfg = dict(enc_dim=encoder_embedding_planes, enc_depth=encoder_block_nums, enc_heads=encoder_head_nums,
                        dec_dim=decoder_embedding_planes, dec_depth=decoder_block_nums, dec_heads=decoder_head_nums)
        n = (image_size // patch_size)**2
        E, D = encoder_embedding_planes, decoder_embedding_planes
        enc = nn.Module()
        enc.cls_token = nn.Parameter(torch.zeros(1, 1, E))
        enc.pos_embed = nn.Parameter(torch.zeros(1, n + 1, E), requires_grad=False)
        enc.patch_embed = nn.Module()
        enc.patch_embed.proj = nn.Conv2d(3, E, patch_size, stride=patch_size, padding=0, bias=True)
        enc.patch_embed.norm = nn.Identity()
        enc.blocks = nn.ModuleList([_block(E, encoder_head_nums, encoder_feedforward_ratio)
                                    for _ in range(encoder_block_nums)])
        enc.norm = nn.LayerNorm(E, eps=1e-6)
        dec = nn.Module()
        dec.mask_token = nn.Parameter(torch.zeros(1, 1, D))
        dec.pos_embed = nn.Parameter(torch.zeros(1, n + 1, D), requires_grad=False)
        dec.blocks = nn.ModuleList([_block(D, decoder_head_nums, decoder_feedforward_ratio)
                                    for _ in range(decoder_block_nums)])
        dec.norm = nn.LayerNorm(D, eps=1e-6)
        dec.fc = nn.Linear(D, patch_size * patch_size * 3)
        self.encoder, self.decoder = enc, dec
        self.encoder_to_decoder = nn.Linear(E, D)
        # initialisers of the reference (vit_mae.py:75-98, :272-285, :400-409)
        g = image_size // patch_size
        enc.pos_embed.data.copy_(torch.from_numpy(sincos_position_encoding(E, g)).float().unsqueeze(0))
        dec.pos_embed.data.copy_(torch.from_numpy(sincos_position_encoding(D, g)).float().unsqueeze(0))
        w = enc.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.normal_(enc.cls_token, std=.02)
        nn.init.normal_(dec.mask_token, std=.02)
        _init_linear_ln(self)
        self._arena = self._engine = self._param_list = None
        self._next_noise = None
        self._last_rows = None

    # ---- HIP runtime -----------------------------------------------------------------------------------------
    def prepare(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('sacv_amd VITMAEPretrainModel runs on MI355X only: move the model to the GPU (.cuda()); '
                               'there is no CPU/PyTorch fallback path')
        refuse_sync_batchnorm(self)
        self._arena = ParamArena(self)
        self._param_list = list(self._arena.order)
        tensors = {n: p.data for n, p in self.named_parameters()}
        c = self.cfg
        self._engine = MAEEngine(tensors, self._arena.grad_views, self._arena, self.patch_size, self.image_size,
                                 c['enc_dim'], c['enc_depth'], c['enc_heads'], c['dec_dim'], c['dec_depth'],
                                 c['dec_heads'], self.mask_ratio)
        return self

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        p = next(self.parameters(), None)
        if p is not None and p.device.type == 'cuda':
            self.prepare()
        return out

    def _ready(self):
        if self._arena is None or not self._arena.is_current():
            self.prepare()

    @property
    def arena(self):
        self._ready()
        return self._arena

    @property
    def engine(self):
        self._ready()
        return self._engine

    def set_noise(self, noise):
        """Masking noise [B, n] in [0, 1) for the NEXT forward (tests; default: torch.rand on the device, as the
        reference draws it inside random_masking, vit_mae.py:198)."""
        self._next_noise = noise
        return self

    @torch.compiler.disable  # config.use_compile (torch.compile(model), tools/train_detection_model.py:176-178): nothing to trace --
    # the forward pass is one call into the HIP executor; the compiled wrapper runs it as is
    def forward(self, x):
        if x.device.type != 'cuda':
            raise RuntimeError('sacv_amd VITMAEPretrainModel runs on MI355X only: inputs must be on the GPU')
        self._ready()
        x = x.float()
        b = x.shape[0]
        if torch.is_grad_enabled() and self.training:
            out, mask = _MAEFunction.apply(self, x, *[p for _, p in self._param_list])
            out._sacv_rows = self._last_rows
            return out, mask
        pred, mask = self._engine.forward(x, noise=self._next_noise, keep=False)
        self._next_noise = None
        out = pred.view(b, -1, pred.shape[-1])[:, 1:, :]
        out._sacv_rows = pred
        return out, mask

    def images_to_patch(self, images):
        """[N,3,H,W] -> [N, L, P*P*3] (vit_mae.py:418-433), on the device."""
        from ... import vit_ops as V
        return V.patch_labels(images.float(), self.patch_size, norm=False)


def vit_base_patch16_224_mae_pretrain_model(**kwargs):
    return VITMAEPretrainModel(patch_size=16, image_size=224, encoder_embedding_planes=768, encoder_block_nums=12,
                               encoder_head_nums=12, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                               decoder_embedding_planes=512, decoder_block_nums=8, decoder_head_nums=16,
                               decoder_feedforward_ratio=4, decoder_dropout_prob=0., **kwargs)


def vit_large_patch16_224_mae_pretrain_model(**kwargs):
    return VITMAEPretrainModel(patch_size=16, image_size=224, encoder_embedding_planes=1024, encoder_block_nums=24,
                               encoder_head_nums=16, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                               decoder_embedding_planes=512, decoder_block_nums=8, decoder_head_nums=16,
                               decoder_feedforward_ratio=4, decoder_dropout_prob=0., **kwargs)


def vit_huge_patch14_224_mae_pretrain_model(**kwargs):
    return VITMAEPretrainModel(patch_size=14, image_size=224, encoder_embedding_planes=1280, encoder_block_nums=32,
                               encoder_head_nums=16, encoder_feedforward_ratio=4, encoder_dropout_prob=0.,
                               decoder_embedding_planes=512, decoder_block_nums=8, decoder_head_nums=16,
                               decoder_feedforward_ratio=4, decoder_dropout_prob=0., **kwargs)
