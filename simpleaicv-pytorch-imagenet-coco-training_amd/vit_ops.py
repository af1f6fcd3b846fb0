"""Tensor-level wrappers of the ViT / MAE kernels (csrc/vit.hip) and the Linear layers as 1x1 problems of the bf16
matrix-core GEMMs (csrc/conv_bf16.hip).  Same conventions as ops.py: torch for memory and the stream only."""
import torch

from . import ops
from ._lib import lib, check
from .ops import BF16, _ptr, _stream, workspace, _chk


def random_masking(noise, keep_len):
    """noise [B,n] fp32 -> (shuffle_ids, restore_ids, keep_ids [B,keep], mask [B,n]); indices int64."""
    _chk(noise, 'noise')
    b, n = noise.shape
    dev = noise.device
    shuffle = torch.empty((b, n), device=dev, dtype=torch.int64)
    restore = torch.empty((b, n), device=dev, dtype=torch.int64)
    keep = torch.empty((b, keep_len), device=dev, dtype=torch.int64)
    mask = torch.empty((b, n), device=dev, dtype=torch.float32)
    check(lib.sacv_mae_random_masking(_ptr(noise), b, n, keep_len, _ptr(shuffle), _ptr(restore), _ptr(keep), _ptr(mask),
                                      _stream()), 'mae_random_masking')
    return shuffle, restore, keep, mask


def patchify(images, patch, ids=None):
    """images logical [B,3,H,W] fp32 (any strides) -> bf16 rows [B*keep, 3*P*P] of the patches ids [B,keep]."""
    if images.dtype != torch.float32 or images.device.type != 'cuda':
        raise ValueError('patchify: expected an fp32 CUDA(HIP) tensor')
    b, c, h, w = images.shape
    keep = ids.shape[1] if ids is not None else (h // patch) * (w // patch)
    rows = torch.empty((b * keep, c * patch * patch), device=images.device, dtype=BF16)
    sn, sc, sh, sw = images.stride()
    check(lib.sacv_mae_patchify(_ptr(images), _ptr(ids), _ptr(rows), b, c, h, w, patch, keep, sn, sc, sh, sw, _stream()),
          'mae_patchify')
    return rows


def patch_labels(images, patch, norm=True):
    b, c, h, w = images.shape
    out = torch.empty((b, (h // patch) * (w // patch), patch * patch * c), device=images.device, dtype=torch.float32)
    sn, sc, sh, sw = images.stride()
    check(lib.sacv_mae_patch_labels(_ptr(images), _ptr(out), b, c, h, w, patch, int(norm), sn, sc, sh, sw, _stream()),
          'mae_patch_labels')
    return out


def encoder_tokens(emb, cls, pos, keep_ids):
    b, keep = keep_ids.shape
    e = emb.shape[-1]
    x = torch.empty((b * (keep + 1), e), device=emb.device, dtype=torch.float32)
    check(lib.sacv_mae_encoder_tokens(_ptr(emb), _ptr(cls), _ptr(pos), _ptr(keep_ids), _ptr(x), b, keep, e, _stream()),
          'mae_encoder_tokens')
    return x


def encoder_tokens_bwd(dx, b, keep):
    e = dx.shape[-1]
    demb = torch.empty((b * keep, e), device=dx.device, dtype=BF16)
    check(lib.sacv_mae_encoder_tokens_bwd(_ptr(dx), _ptr(demb), b, keep, e, _stream()), 'mae_encoder_tokens_bwd')
    return demb


def decoder_tokens(y, mask_token, pos, restore_ids, keep):
    b, n = restore_ids.shape
    d = y.shape[-1]
    x = torch.empty((b * (n + 1), d), device=y.device, dtype=torch.float32)
    check(lib.sacv_mae_decoder_tokens(_ptr(y), _ptr(mask_token), _ptr(pos), _ptr(restore_ids), _ptr(x), b, keep, n, d,
                                      _stream()), 'mae_decoder_tokens')
    return x


def decoder_tokens_bwd(dx, shuffle_ids, keep):
    b, n = shuffle_ids.shape
    d = dx.shape[-1]
    dy = torch.empty((b * (keep + 1), d), device=dx.device, dtype=BF16)
    gmask = torch.empty((b * (n - keep), d), device=dx.device, dtype=torch.float32)
    check(lib.sacv_mae_decoder_tokens_bwd(_ptr(dx), _ptr(shuffle_ids), _ptr(dy), _ptr(gmask), b, keep, n, d, _stream()),
          'mae_decoder_tokens_bwd')
    return dy, gmask


def layernorm_fwd(x, gamma, beta, eps=1e-6):
    """x fp32 [R,C] -> (y bf16, mean, rstd)."""
    _chk(x, 'layernorm x')
    r, c = x.shape
    y = torch.empty((r, c), device=x.device, dtype=BF16)
    mean = torch.empty(r, device=x.device, dtype=torch.float32)
    rstd = torch.empty(r, device=x.device, dtype=torch.float32)
    check(lib.sacv_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), r, c, eps, _stream()),
          'layernorm_fwd')
    return y, mean, rstd


class ColsumQueue:
    """Deferred second stages of column sums (Linear bias gradients, LayerNorm dgamma / dbeta): the first-stage
    kernels write their partials into buffers of their own, `flush()` finalizes up to 16 of them per launch
    (sacv_colsum_finalize_multi) -- a transformer block's backward pass has 8, each a ~10 us launch on its own."""

    def __init__(self):
        self.jobs = []

    def add(self, partial, nblocks, stride, c, out, accumulate=False):
        self.jobs.append((partial, nblocks, stride, c, out, int(accumulate)))
        if len(self.jobs) == 16:
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        j = self.jobs
        check(lib.sacv_colsum_finalize_multi(len(j), ops.parr([p.data_ptr() for p, *_ in j]), ops.iarr([x[1] for x in j]),
                                             ops.iarr([x[2] for x in j]), ops.iarr([x[3] for x in j]),
                                             ops.parr([x[4].data_ptr() for x in j]), ops.iarr([x[5] for x in j]),
                                             _stream()), 'colsum_finalize_multi')
        self.jobs = []


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dx=None, accumulate_params=False, queue=None):
    """dx (given: accumulated into; None: created) fp32; dgamma / dbeta fp32 gradient views.  queue: a ColsumQueue
    that finalizes dgamma / dbeta later (their partials then live in a buffer of this call)."""
    r, c = x.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty((r, c), device=x.device, dtype=torch.float32)
    nfl = lib.sacv_layernorm_bwd_workspace_floats(r, c)
    if queue is None:
        ws = workspace(nfl * 4, x.device, 'ln')
        check(lib.sacv_layernorm_bwd(_ptr(_chk(dy, 'dy', BF16)), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx),
                                     _ptr(dgamma), _ptr(dbeta), r, c, int(acc), int(accumulate_params),
                                     ws.data_ptr(), _stream()), 'layernorm_bwd')
        return dx
    part = torch.empty(nfl, device=x.device, dtype=torch.float32)
    check(lib.sacv_layernorm_bwd(_ptr(_chk(dy, 'dy', BF16)), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx),
                                 None, None, r, c, int(acc), int(accumulate_params), part.data_ptr(), _stream()),
          'layernorm_bwd')
    nb = lib.sacv_layernorm_bwd_partial_rows(r)
    queue.add(part, nb, 2 * c, c, dbeta, accumulate_params)
    queue.add(part[c:], nb, 2 * c, c, dgamma, accumulate_params)
    return dx


def gelu_fwd(x):
    y = torch.empty_like(x)
    check(lib.sacv_gelu_fwd(_ptr(_chk(x, 'gelu x', BF16)), _ptr(y), x.numel(), _stream()), 'gelu_fwd')
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    check(lib.sacv_gelu_bwd(_ptr(dy), _ptr(x), _ptr(dx), x.numel(), _stream()), 'gelu_bwd')
    return dx


def colsum_bf16(x, out, accumulate=False, queue=None):
    r, c = x.shape
    nfl = lib.sacv_colsum_bf16_workspace_floats(r, c)
    if queue is None:
        ws = workspace(nfl * 4, x.device, 'colsum')
        check(lib.sacv_colsum_bf16(_ptr(_chk(x, 'colsum x', BF16)), _ptr(out), r, c, int(accumulate), ws.data_ptr(),
                                   _stream()), 'colsum_bf16')
        return out
    part = torch.empty(nfl, device=x.device, dtype=torch.float32)
    check(lib.sacv_colsum_bf16(_ptr(_chk(x, 'colsum x', BF16)), None, r, c, int(accumulate), part.data_ptr(),
                               _stream()), 'colsum_bf16')
    queue.add(part, lib.sacv_colsum_bf16_partial_rows(r), c, c, out, accumulate)
    return out


def attention_fwd(qkv, b, n, heads, mfma=None):
    """qkv bf16 [B*n, 3*C] (= [B][n][3][H][d]) -> (out bf16 [B*n, C], lse fp32 [B,H,n] or None).
    mfma: None = the matrix-core kernel whenever it covers (n, d); False = the general VALU kernel."""
    c = qkv.shape[1] // 3
    d = c // heads
    out = torch.empty((b * n, c), device=qkv.device, dtype=BF16)
    _chk(qkv, 'qkv', BF16)
    if mfma is None:
        mfma = bool(lib.sacv_attention_mfma_supported(n, d))
    if mfma:
        lse = torch.empty((b, heads, n), device=qkv.device, dtype=torch.float32)
        check(lib.sacv_attention_fwd_mfma(_ptr(qkv), _ptr(out), _ptr(lse), b, n, heads, d, float(d)**-0.5, _stream()),
              'attention_fwd_mfma')
        return out, lse
    check(lib.sacv_attention_fwd(_ptr(qkv), _ptr(out), b, n, heads, d, float(d)**-0.5, _stream()), 'attention_fwd')
    return out, None


def attention_bwd(qkv, out, dout, lse, b, n, heads):
    """lse None: the VALU kernel (recomputes the softmax itself); else the matrix-core kernel."""
    c = qkv.shape[1] // 3
    d = c // heads
    dqkv = torch.empty_like(qkv)
    _chk(dout, 'dout', BF16)
    if lse is not None:
        check(lib.sacv_attention_bwd_mfma(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), b, n, heads, d,
                                          float(d)**-0.5, _stream()), 'attention_bwd_mfma')
    else:
        check(lib.sacv_attention_bwd(_ptr(qkv), _ptr(dout), _ptr(dqkv), b, n, heads, d, float(d)**-0.5, _stream()),
              'attention_bwd')
    return dqkv


def mse_loss(pred, label, mask, want_grad=True, grad_scale=1.0):
    """pred bf16 [B*(n+1), K] (row 0 of every sample = cls token), label fp32 [B,n,K], mask [B,n] -> (loss f32[1],
    dpred bf16 or None)."""
    b, n = mask.shape
    k = pred.shape[-1]
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred) if want_grad else None
    ws = workspace(lib.sacv_mae_mse_workspace_floats(b, n) * 4, pred.device, 'mse')
    check(lib.sacv_mae_mse_loss(_ptr(_chk(pred, 'pred', BF16)), _ptr(_chk(label, 'label')), _ptr(_chk(mask, 'mask')),
                                _ptr(dpred), _ptr(loss), b, n, k, grad_scale, ws.data_ptr(), _stream()), 'mae_mse_loss')
    return loss, dpred


# ------------------------------------------------------------------------------------------ Linear = 1x1 GEMM
def linear_fwd(x, w_panel, bias, out=None, accumulate=False, out_f32=False, residual=None):
    """y[R,N] = x[R,K] @ W[N,K]^T + bias (+ residual, fp32 [R,N]) on the bf16 matrix cores.  out_f32: fp32
    destination; accumulate: y += ..."""
    r, k = x.shape
    n = w_panel.shape[0]
    flags = (ops.CFH_OUT_F32 if out_f32 else 0) | (ops.CF_ACCUM if accumulate else 0) | (ops.CF_BIAS if bias is not None else 0)
    if out is None:
        out = torch.empty((r, n), device=x.device, dtype=torch.float32 if out_f32 else BF16)
    _chk(x, 'linear x', BF16)
    _chk(w_panel, 'linear weight panel', BF16)
    _chk(out, 'linear out', torch.float32 if out_f32 else BF16)
    if residual is not None:
        _chk(residual, 'linear residual')
        if tuple(residual.shape) != (r, n):
            raise ValueError(f'linear_fwd: residual is {tuple(residual.shape)}, expected ({r}, {n})')

    def launch():
        check(lib.sacv_linear_bf16(_ptr(x), _ptr(w_panel), _ptr(out), _ptr(bias), _ptr(residual), r, n, k, flags,
                                   _stream()), 'linear_bf16')

    if ops.TIMER is not None:
        ops.TIMER.time(f'conv_nt_bf16|m0 k1 s1 cs{k} n{n} pix{r} f{flags}', 2.0 * r * n * k, launch,
                       nbytes=2.0 * (x.numel() + w_panel.numel()) + out.numel() * (4 if out_f32 else 2))
    else:
        launch()
    return out


def linear_dgrad(dy, w_dpanel, k):
    """dx[R,K] = dy[R,N] @ W[N,K]; w_dpanel = the [K][N] bf16 panel."""
    r, n = dy.shape
    return ops.conv_gemm([dy.view(1, 1, r, n)], w_dpanel, k, 1, 1, 1, 0, 1, [(1, r)])[0].view(r, k)


def linear_wgrad(dy, x, dw, db, queue=None, side=None):
    """dw[N,K] = dy^T x (fp32 gradient view), db[N] = column sums of dy (second stage through `queue` when given).
    `side(launch, tensors)`: the engine's weight-gradient stream (engine.DetectorEngine._on_side)."""
    r, n = dy.shape
    k = x.shape[1]
    launch = lambda: ops.conv_wgrad([dy.view(1, 1, r, n)], [x.view(1, 1, r, k)], dw.reshape(n, k, 1, 1), k, 1, 1, 1, 0)
    if side is not None:
        side(launch, [dy, x])
    else:
        launch()
    if db is not None:
        colsum_bf16(dy, db, queue=queue)
