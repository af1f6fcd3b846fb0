"""Tensor-level wrappers over the C ABI (libsacv_hip.so).

torch is used for device memory and the current HIP stream only; every computation below is
a hand-written gfx950 kernel reached through ctypes on raw device pointers.  All activations
are NHWC-contiguous fp32 tensors of logical shape [N, H, W, C].
"""
import os

import torch

from . import _lib
from ._lib import lib, check, iarr, farr, parr

CF_BIAS, CF_RELU, CF_SIGMOID, CF_ACCUM, CF_MASK, CF_STATS = 1, 2, 4, 8, 16, 32
CF_AFFINE = 128  # bias = [2][N] (shift, scale): an eval-mode BatchNorm folded into the convolution
CFH_OUT_F32 = 64  # bf16 convolution with an fp32 destination (the classifier's logits)
BF16 = torch.bfloat16
PACK_FWD, PACK_DGRAD, PACK_STEM = 0, 1, 2
STEM_KPAD = 160  # 7*7*3 = 147 padded to a multiple of 32


def stem_kpad(k, cin=3, granule=32):
    """GEMM-K of the im2col'ed stem convolution (k x k taps x cin), padded to a multiple of the K tile (32 fp32
    elements; 64 for the bf16 kernels)."""
    return -(-(k * k * cin) // granule) * granule


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """Raw hipStream_t of torch's current stream.  torch.cuda.current_stream() builds a Stream object through several
    Python layers (653 calls and ~1 ms of host time per RetinaNet step); the two C entry points it ends in are called
    directly when this torch build has them."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event timing of individual launches on the launch stream (torch's current stream).
    bench.py installs one as ops.TIMER to measure the conv kernels' achieved FLOP/s live."""

    def __init__(self):
        self.records = []
        self.bytes = []

    def time(self, key, flops, launch, nbytes=0.0):
        """flops / nbytes: ALGORITHMIC work of the launch (conv MACs x 2; input + output + weights touched once)."""
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = launch()
        e1.record()
        self.records.append((key, flops, e0, e1))
        self.bytes.append(nbytes)
        return out

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for (key, flops, e0, e1), nb in zip(self.records, self.bytes):
            a = agg.setdefault(key, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0))
            a['launches'] += 1
            a['flops'] += flops
            a['ms'] += e0.elapsed_time(e1)
            a['bytes'] += nb
        return agg


TIMER = None
_GEMM_WS_BYTES = None


def _chk(t, name='tensor', dtype=torch.float32):
    if t.device.type != 'cuda' or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f'{name}: expected a contiguous {dtype} CUDA(HIP) tensor, got {t.dtype} {t.device} '
                         f'contiguous={t.is_contiguous()}')
    return t


def cast_bf16(t, out=None):
    """fp32 -> bf16 (round to nearest even) on the device."""
    _chk(t, 'cast_bf16 input')
    out = out if out is not None else torch.empty(t.shape, device=t.device, dtype=BF16)
    check(lib.sacv_cast_bf16(_ptr(t), _ptr(out), t.numel(), _stream()), 'cast_bf16')
    return out


def cast_f32(t):
    _chk(t, 'cast_f32 input', BF16)
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    check(lib.sacv_cast_f32(_ptr(t), _ptr(out), t.numel(), _stream()), 'cast_f32')
    return out


def _ptr(t):
    return t.data_ptr() if t is not None else None


# ------------------------------------------------------------------------------------------ conv
def weight_layout(w):
    """0: OIHW-contiguous, 1: channels_last storage [Cout][KH][KW][Cin] (what ParamArena gives k>1 conv weights).
    1x1 weights are both; they report 0."""
    if w.device.type != 'cuda' or w.dtype != torch.float32 or w.ndim != 4:
        raise ValueError('weight: expected a 4-D fp32 CUDA(HIP) tensor')
    if w.is_contiguous():
        return 0
    if w.permute(0, 2, 3, 1).is_contiguous():
        return 1
    raise ValueError(f'weight: strides {w.stride()} are neither OIHW-contiguous nor channels_last')


def fwd_panel(w):
    """Forward GEMM panel [Cout][KH*KW*Cin] of a conv weight: zero-copy when the weight is stored channels_last
    (or is 1x1), otherwise packed."""
    co, ci, kh, kw = w.shape
    if weight_layout(w) == 1:
        return w.permute(0, 2, 3, 1)
    if kh * kw == 1:
        return w
    return pack_weight(w, PACK_FWD)


def pack_weight(w_oihw, kind, out=None, kpad=None):
    layout = weight_layout(w_oihw)
    if layout == 1:
        if kind == PACK_FWD:
            return fwd_panel(w_oihw)
        if kind != PACK_DGRAD:
            raise ValueError('pack_weight: the stem panel is packed from an OIHW weight')
    co, ci, kh, kw = w_oihw.shape
    if layout == 1:
        out = out if out is not None else torch.empty((ci, kh * kw * co), device=w_oihw.device, dtype=torch.float32)
        check(lib.sacv_pack_weight(_ptr(w_oihw), _ptr(out), co, ci, kh, kw, 3, 0, _stream()), 'pack_weight')
        return out
    if kind == PACK_STEM:
        kpad = kpad or stem_kpad(kh, ci)
        shape = (co, kpad)
    elif kind == PACK_FWD:
        shape, kpad = (co, kh * kw * ci), 0
    else:
        shape, kpad = (ci, kh * kw * co), 0
    if out is None:
        out = torch.empty(shape, device=w_oihw.device, dtype=torch.float32)
    check(lib.sacv_pack_weight(_ptr(w_oihw), _ptr(out), co, ci, kh, kw, kind, kpad, _stream()), 'pack_weight')
    return out


def _gemm_ws_bytes():
    global _GEMM_WS_BYTES
    if _GEMM_WS_BYTES is None:
        _GEMM_WS_BYTES = lib.sacv_conv2d_gemm_workspace_bytes()  # queries the CU count: needs the GPU
    return _GEMM_WS_BYTES


def conv_gemm(srcs, w_packed, n_out, kh, kw, stride, pad, mode, out_hw, bias=None, flags=0, masks=None,
              outs=None, remap=None, stats=False, ext=None):
    """Grouped implicit-GEMM conv.  srcs: list of NHWC tensors sharing the channel count.

    mode 0: forward (out_hw[i] = output H,W);  mode 1: data gradient (out_hw[i] = input H,W).
    remap = (s, Hr, Wr): single-problem strided scatter of the GEMM rows into a [N,Hr,Wr,n_out]
    tensor (used for the 1x1 stride-s data gradient), requires outs and CF_ACCUM semantics.
    """
    n = len(srcs)
    cs = srcs[0].shape[3]
    half = srcs[0].dtype == BF16
    if half:
        return _conv_gemm_bf16(srcs, w_packed, n_out, kh, kw, stride, pad, mode, out_hw, bias, flags, masks, outs,
                               remap, stats, ext)
    if outs is None:
        outs = [torch.empty((s.shape[0], hw[0], hw[1], n_out), device=s.device, dtype=torch.float32)
                for s, hw in zip(srcs, out_hw)]
    for s in srcs:
        _chk(s, 'conv src')
    if bias is not None:
        flags |= CF_BIAS
    rs, hr, wr = remap if remap else (0, 0, 0)
    stats_buf = None
    if stats:
        rows = sum(-(-(s.shape[0] * hw[0] * hw[1]) // 128) for s, hw in zip(srcs, out_hw))
        stats_buf = torch.empty((rows, 2, n_out), device=srcs[0].device, dtype=torch.float32)
        flags |= CF_STATS

    def launch():
        check(lib.sacv_conv2d_gemm(
            n, parr([s.data_ptr() for s in srcs]), parr([o.data_ptr() for o in outs]),
            parr([m.data_ptr() for m in masks]) if masks else None,
            iarr([s.shape[0] for s in srcs]), iarr([s.shape[1] for s in srcs]), iarr([s.shape[2] for s in srcs]),
            iarr([hw[0] for hw in out_hw]), iarr([hw[1] for hw in out_hw]),
            _ptr(w_packed), _ptr(bias), cs, n_out, kh, kw, stride, pad, mode, flags, rs, hr, wr,
            iarr(ext) if ext is not None else None,
            workspace(_gemm_ws_bytes(), srcs[0].device, 'conv_gemm').data_ptr(), _ptr(stats_buf), _stream()),
            'conv2d_gemm')

    if TIMER is not None:
        # ALGORITHMIC flops of the convolution (forward-equivalent MACs x 2), not the padded GEMM's
        if mode == 0:
            pix = sum(s.shape[0] * hw[0] * hw[1] for s, hw in zip(srcs, out_hw))
        else:
            pix = sum(s.shape[0] * s.shape[1] * s.shape[2] for s in srcs)
        TIMER.time(_nt_family(n_out) + f'|m{mode} k{kh} s{stride} cs{cs} n{n_out} np{n} pix{pix} f{flags}',
                   2.0 * pix * n_out * kh * kw * cs, launch,
                   nbytes=4.0 * (sum(t.numel() for t in srcs) + sum(o.numel() for o in outs) + w_packed.numel()))
    else:
        launch()
    if stats:
        return outs, stats_buf
    return outs


def conv_gemm_bnbwd(src, w_packed, n_out, kh, kw, pad, out_hw, bn_y, bn_mean, bn_invstd, bn_gamma=None, bn_beta=None,
                    bn_z=None, out=None, flags=0, mask=None, mode=1, stride=1):
    """One stride-1 data gradient (mode 1; out_hw = input H, W) whose output dz is the upstream gradient of the
    BatchNorm (+ ReLU) layer with input bn_y: returns (dz, partial) with partial [m tiles, 2, n_out] = that layer's
    backward first-stage sums (sum g, sum g * xhat) for bn_train_bwd(partial=...).  Mask from bn_z (stored output) or
    recomputed from y with bn_gamma / bn_beta.  bf16 activations: sacv_conv2d_gemm_bf16_bnbwd (no mask operand)."""
    if src.dtype == BF16:
        return _conv_gemm_bf16_bnbwd(src, w_packed, n_out, kh, kw, pad, out_hw, bn_y, bn_mean, bn_invstd, bn_gamma,
                                     bn_beta, bn_z, out, flags, mask, mode, stride)
    _chk(src, 'conv src')
    cs = src.shape[3]
    nb = src.shape[0]
    if out is None:
        out = torch.empty((nb, out_hw[0], out_hw[1], n_out), device=src.device, dtype=torch.float32)
    if tuple(bn_y.shape) != tuple(out.shape):
        raise ValueError(f'conv_gemm_bnbwd: bn_y is {tuple(bn_y.shape)}, the output {tuple(out.shape)}')
    rows = -(-(nb * out_hw[0] * out_hw[1]) // 128)
    st = torch.empty((rows, 2, n_out), device=src.device, dtype=torch.float32)
    if mask is not None:
        flags |= CF_MASK

    def launch():
        check(lib.sacv_conv2d_gemm_bnbwd(
            src.data_ptr(), out.data_ptr(), _ptr(mask), nb, src.shape[1], src.shape[2], out_hw[0], out_hw[1],
            _ptr(w_packed), None, cs, n_out, kh, kw, stride, pad, mode, flags,
            workspace(_gemm_ws_bytes(), src.device, 'conv_gemm').data_ptr(), st.data_ptr(), _ptr(_chk(bn_y, 'bn_y')),
            _ptr(bn_z), _ptr(bn_mean), _ptr(bn_invstd), _ptr(bn_gamma), _ptr(bn_beta), _stream()), 'conv2d_gemm_bnbwd')

    if TIMER is not None:
        pix = nb * (src.shape[1] * src.shape[2] if mode == 1 else out_hw[0] * out_hw[1])
        TIMER.time(_nt_family(n_out) + f'|m{mode} k{kh} s{stride} cs{cs} n{n_out} np1 pix{pix} f{flags | 256}',
                   2.0 * pix * n_out * kh * kw * cs, launch,
                   nbytes=4.0 * (src.numel() + 2 * out.numel() + w_packed.numel()))
    else:
        launch()
    return out, st


def _conv_gemm_bf16_bnbwd(src, w_packed, n_out, kh, kw, pad, out_hw, bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_z,
                          out, flags, mask, mode, stride):
    if mask is not None or (flags & ~(CF_BIAS | CF_ACCUM)):
        raise ValueError(f'bf16 conv_gemm_bnbwd: no mask epilogue, flags BIAS / ACCUM only (flags={flags})')
    _chk(src, 'conv src', BF16)
    _chk(w_packed, 'bf16 weight panel', BF16)
    cs, nb = src.shape[3], src.shape[0]
    if out is None:
        out = torch.empty((nb, out_hw[0], out_hw[1], n_out), device=src.device, dtype=BF16)
    _chk(out, 'conv dst', BF16)
    if tuple(bn_y.shape) != tuple(out.shape):
        raise ValueError(f'conv_gemm_bnbwd: bn_y is {tuple(bn_y.shape)}, the output {tuple(out.shape)}')
    _chk(bn_y, 'bn_y', BF16)
    if bn_z is not None:
        _chk(bn_z, 'bn_z', BF16)
    rows = -(-(nb * out_hw[0] * out_hw[1]) // 128)
    st = torch.empty((rows, 2, n_out), device=src.device, dtype=torch.float32)

    def launch():
        check(lib.sacv_conv2d_gemm_bf16_bnbwd(
            src.data_ptr(), out.data_ptr(), nb, src.shape[1], src.shape[2], out_hw[0], out_hw[1], _ptr(w_packed), None,
            cs, n_out, kh, kw, stride, pad, mode, flags, st.data_ptr(), bn_y.data_ptr(), _ptr(bn_z), _ptr(bn_mean),
            _ptr(bn_invstd), _ptr(bn_gamma), _ptr(bn_beta), _stream()), 'conv2d_gemm_bf16_bnbwd')

    if TIMER is not None:
        pix = nb * (src.shape[1] * src.shape[2] if mode == 1 else out_hw[0] * out_hw[1])
        TIMER.time(f'conv_nt_bf16|m{mode} k{kh} s{stride} cs{cs} n{n_out} pix{pix} f{flags | 256}',
                   2.0 * pix * n_out * kh * kw * cs, launch,
                   nbytes=2.0 * (src.numel() + 2 * out.numel() + w_packed.numel()))
    else:
        launch()
    return out, st


# ------------------------------------------------------------------------------------------ Winograd F(2x2,3x3)
WINO_MAXL = 5


def wino_tiles(nb, hw):
    """(tiles, tiles padded to the GEMM's 64-row granule) of a pyramid of [nb, h, w, .] levels."""
    t = sum(nb * ((h + 1) // 2) * ((w + 1) // 2) for h, w in hw)
    return t, -(-t // 64) * 64


def wino_pad_k(k):
    return -(-k // 32) * 32


def wino_pad_n(n):
    return -(-n // 128) * 128


def wino_operand_shape(rows, cols, split):
    """A transformed operand: fp32 [16][rows][cols], or (split) the three bf16 planes [16][3][rows][cols] whose sum is
    the fp32 value exactly -- the form the GEMM multiplies on the bf16 matrix pipe (csrc/winograd.hip header)."""
    return ((16, 3, rows, cols), torch.bfloat16) if split else ((16, rows, cols), torch.float32)


def wino_is_split(t):
    return t.dtype == torch.bfloat16


def wino_split_rows(t):
    """A split operand [16][3][rows][cols] as plain row-major planes.  In memory the 32-column chunks of a row pair lie
    side by side ([rows / 2][cols / 32][2][32], csrc/winograd.hip wino_split_off: the GEMM's 64-byte stage pieces then
    fill whole 128-byte lines); the tensor keeps the logical shape, this view undoes the interleave."""
    g, p, rows, cols = t.shape
    return t.reshape(g, p, rows // 2, cols // 32, 2, 32).permute(0, 1, 2, 4, 3, 5).reshape(g, p, rows, cols)


def wino_weights(panel, n, k, flip=False, out=None, split=False):
    """U[16][pad_n(n)][pad_k(k)] = G w G^T of a GEMM panel [n][9][k] (forward panel; the data-gradient panel with
    flip=True); zero-padded to the GEMM's 128 x 32 operand tiles."""
    if panel.numel() != n * 9 * k:
        raise ValueError(f'wino_weights: panel of {panel.numel()} elements is not [{n}][9][{k}]')
    if out is not None:
        split = wino_is_split(out)
    shape, dt = wino_operand_shape(wino_pad_n(n), wino_pad_k(k), split)
    out = out if out is not None else torch.empty(shape, device=panel.device, dtype=dt)
    if tuple(out.shape) != shape:
        raise ValueError(f'wino_weights: out is {tuple(out.shape)}, expected {shape}')
    check(lib.sacv_wino_weights(_ptr(panel), _ptr(out), n, k, int(flip), int(split), _stream()), 'wino_weights')
    return out


def wino_weights_batched(jobs, split=False):
    """jobs: list of (panel, n, k, flip) -> list of U tensors, transformed by as few launches as possible (32 jobs each)."""
    outs = []
    for p, n, k, _ in jobs:
        shape, dt = wino_operand_shape(wino_pad_n(n), wino_pad_k(k), split)
        outs.append(torch.empty(shape, device=p.device, dtype=dt))
    for b in range(0, len(jobs), 32):
        chunk, us = jobs[b:b + 32], outs[b:b + 32]
        for p, n, k, _ in chunk:
            if p.numel() != n * 9 * k:
                raise ValueError(f'wino_weights_batched: panel of {p.numel()} elements is not [{n}][9][{k}]')
        check(lib.sacv_wino_weights_batched(len(chunk), parr([_ptr(p) for p, *_ in chunk]), parr([u.data_ptr() for u in us]),
                                            iarr([n for _, n, _, _ in chunk]), iarr([k for _, _, k, _ in chunk]),
                                            iarr([int(f) for *_, f in chunk]), int(split), _stream()),
              'wino_weights_batched')
    return outs


def wino_input(srcs, out=None, tag='wino_v', split=False, keep=False):
    """V[16][Tpad][pad_k(C)] = B^T d B over all levels (NHWC fp32, one batch size, one channel count).  keep: V is a
    tensor of its own (it outlives the call: the weight gradient reads it), not a tagged scratch buffer."""
    nb, c = srcs[0].shape[0], srcs[0].shape[3]
    hw = [(s.shape[1], s.shape[2]) for s in srcs]
    for s in srcs:
        _chk(s, 'wino_input src')
    t, tpad = wino_tiles(nb, hw)
    cp = wino_pad_k(c)
    if out is not None:
        split = wino_is_split(out)
    shape, dt = wino_operand_shape(tpad, cp, split)
    if out is None and keep:
        out = torch.empty(shape, device=srcs[0].device, dtype=dt)
    elif out is None:
        nbytes = 16 * tpad * cp * (6 if split else 4)
        out = workspace(nbytes, srcs[0].device, tag)[:nbytes].view(dt).view(shape)
    if tuple(out.shape) != shape:
        raise ValueError(f'wino_input: out is {tuple(out.shape)}, expected {shape}')
    check(lib.sacv_wino_input(len(srcs), parr([s.data_ptr() for s in srcs]), nb, iarr([h for h, _ in hw]),
                              iarr([w for _, w in hw]), c, _ptr(out), tpad, int(split), _stream()), 'wino_input')
    return out


def conv3x3_wino(srcs_list, us, n_outs, biases=None, flags=None, masks_list=None, outs_list=None, stats=False,
                 mode=0, vs=None):
    """3x3 / stride 1 / pad 1 convolution of 1 or 2 independent problems (e.g. layer i of the two head towers) over
    the same pyramid geometry on the Winograd path.  srcs_list[p]: list of NHWC levels; us[p]: wino_weights(...)
    [16][n_outs[p]][C_p]; flags[p]: CF_* epilogue bits.  `mode` only labels the timer record (0 forward, 1 data
    gradient: the caller passes the flipped data-gradient U).  vs[p]: an already transformed input (wino_input).
    The operand form (fp32 / three bf16 planes) is that of us[0]; every U and V of the launch has it.
    Returns outs_list (and the per-problem CF_STATS buffers when stats=True)."""
    npb = len(srcs_list)
    first = srcs_list[0]
    nb = first[0].shape[0]
    hw = [(s.shape[1], s.shape[2]) for s in first]
    nlev = len(first)
    t, tpad = wino_tiles(nb, hw)
    flags = list(flags) if flags is not None else [0] * npb
    biases = biases if biases is not None else [None] * npb
    for p in range(npb):
        if [(s.shape[1], s.shape[2]) for s in srcs_list[p]] != hw or srcs_list[p][0].shape[0] != nb:
            raise ValueError('conv3x3_wino: the problems of one launch share the pyramid geometry')
        if biases[p] is not None:
            flags[p] |= CF_BIAS
        if stats:
            flags[p] |= CF_STATS
    if outs_list is None:
        outs_list = [[torch.empty((nb, h, w, n_outs[p]), device=first[0].device, dtype=torch.float32) for h, w in hw]
                     for p in range(npb)]
    stats_bufs = None
    if stats:
        rows = lib.sacv_wino_stats_rows(t)
        stats_bufs = [torch.empty((rows, 2, n_outs[p]), device=first[0].device, dtype=torch.float32)
                      for p in range(npb)]
    ks = [srcs_list[p][0].shape[3] for p in range(npb)]
    split = wino_is_split(us[0])

    def pad5(lst):
        return list(lst) + [0] * (WINO_MAXL - len(lst))

    def launch():
        v = [vs[p] if vs is not None and vs[p] is not None else wino_input(srcs_list[p], tag=f'wino_v{p}', split=split)
             for p in range(npb)]
        if any(wino_is_split(x) != split for x in v):
            raise ValueError('conv3x3_wino: U and V differ in operand form (fp32 / split)')
        dst = sum([pad5([o.data_ptr() for o in outs_list[p]]) for p in range(npb)], [])
        msk = None
        if masks_list is not None and any(m is not None for m in masks_list):
            msk = sum([pad5([m.data_ptr() for m in masks_list[p]]) if masks_list[p] is not None else pad5([])
                       for p in range(npb)], [])
        check(lib.sacv_wino_gemm(
            npb, nlev, nb, iarr([h for h, _ in hw]), iarr([w for _, w in hw]),
            parr([x.data_ptr() for x in v]), parr([u.data_ptr() for u in us]), iarr([tpad] * npb), iarr(ks),
            iarr(list(n_outs)), parr([_ptr(b) or 0 for b in biases]), parr(dst), parr(msk) if msk else None,
            parr([s.data_ptr() for s in stats_bufs]) if stats_bufs else None, iarr(flags), int(split), _stream()),
            'wino_gemm')

    for p in range(npb):
        want = wino_operand_shape(wino_pad_n(n_outs[p]), wino_pad_k(ks[p]), split)
        if (tuple(us[p].shape), us[p].dtype) != want:
            raise ValueError(f'conv3x3_wino: U[{p}] is {tuple(us[p].shape)} {us[p].dtype}, expected {want}')
    if TIMER is not None:
        pix = sum(nb * h * w for h, w in hw)
        fl = sum(2.0 * pix * n_outs[p] * 9 * ks[p] for p in range(npb))
        nbytes = 4.0 * sum(sum(s.numel() for s in srcs_list[p]) + sum(o.numel() for o in outs_list[p]) +
                           us[p].numel() * 9 / 16 for p in range(npb))
        TIMER.time(f'conv_nt_wino{"3" if split else ""}|m{mode} k3 s1 cs{ks[0]} n{"+".join(str(n) for n in n_outs)} '
                   f'np{nlev} pix{pix} f{flags[0]}', fl, launch, nbytes=nbytes)
    else:
        launch()
    if stats:
        return outs_list, stats_bufs
    return outs_list


def wino_wgrad(dys, v, dw, cin, accumulate=False, exact=False):
    """dw (logical OIHW view, stored OIHW or channels_last) (+)= x (*) dy of a 3x3 / stride 1 / pad 1 convolution over
    the pyramid, from v = wino_input(x) of the forward pass and the per-level output gradients dys.  A split V (three
    bf16 planes) sends the GEMMs to the bf16 matrix pipe (the transformed dy is split the same way; six products,
    fp32 accumulators); exact=True rebuilds the fp32 V from its planes and runs the fp32 pipe instead -- the same
    bits as with an fp32 V."""
    nb, cout = dys[0].shape[0], dys[0].shape[3]
    hw = [(d.shape[1], d.shape[2]) for d in dys]
    for d in dys:
        _chk(d, 'wino_wgrad dy')
    t, tpad = wino_tiles(nb, hw)
    v_split = wino_is_split(v)
    form = 0 if not v_split else (1 if exact or cout % 8 else 2)
    if tuple(v.shape) != wino_operand_shape(tpad, wino_pad_k(cin), v_split)[0]:
        raise ValueError(f'wino_wgrad: V is {tuple(v.shape)}, expected '
                         f'{wino_operand_shape(tpad, wino_pad_k(cin), v_split)[0]}')
    if tuple(dw.shape) != (cout, cin, 3, 3):
        raise ValueError(f'wino_wgrad: dw is {tuple(dw.shape)}, expected ({cout}, {cin}, 3, 3)')
    layout = weight_layout(dw)
    ws = workspace(lib.sacv_wino_wgrad_workspace_bytes(tpad, cout, cin), dys[0].device, 'wino_wgrad')

    def launch():
        check(lib.sacv_wino_wgrad(len(dys), parr([d.data_ptr() for d in dys]), nb, iarr([h for h, _ in hw]),
                                  iarr([w for _, w in hw]), cout, _ptr(v), form, tpad, cin, _ptr(dw),
                                  int(accumulate), layout,
                                  ws.data_ptr(), _stream()), 'wino_wgrad')

    if TIMER is not None:
        pix = sum(nb * h * w for h, w in hw)
        TIMER.time(f'conv_tn_wino{"3" if form == 2 else ""}|k3 s1 ci{cin} co{cout} np{len(dys)}',
                   2.0 * pix * cout * 9 * cin, launch)
    else:
        launch()
    return dw


def wino_wgrad_supported(cin):
    return wino_pad_k(cin) % 64 == 0


def _conv_gemm_bf16(srcs, w_packed, n_out, kh, kw, stride, pad, mode, out_hw, bias, flags, masks, outs, remap, stats,
                    ext):
    """bf16 activations: one problem per launch, epilogue flags BIAS / ACCUM / STATS / OUT_F32."""
    if len(srcs) != 1 or masks:
        raise ValueError('bf16 conv_gemm: one problem per launch, no mask epilogue')
    if flags & ~(CF_BIAS | CF_ACCUM | CF_STATS | CFH_OUT_F32):
        raise ValueError(f'bf16 conv_gemm: unsupported epilogue flags {flags}')
    s, hw = _chk(srcs[0], 'conv src', BF16), out_hw[0]
    _chk(w_packed, 'bf16 weight panel', BF16)
    cs = s.shape[3]
    odt = torch.float32 if (flags & CFH_OUT_F32) else BF16
    if outs is None:
        outs = [torch.empty((s.shape[0], hw[0], hw[1], n_out), device=s.device, dtype=odt)]
    _chk(outs[0], 'conv dst', odt)
    if bias is not None:
        flags |= CF_BIAS
    rs, hr, wr = remap if remap else (0, 0, 0)
    stats_buf = None
    if stats:
        rows = -(-(s.shape[0] * hw[0] * hw[1]) // 128)
        stats_buf = torch.empty((rows, 2, n_out), device=s.device, dtype=torch.float32)
        flags |= CF_STATS

    def launch():
        check(lib.sacv_conv2d_gemm_bf16(s.data_ptr(), outs[0].data_ptr(), s.shape[0], s.shape[1], s.shape[2], hw[0],
                                        hw[1], _ptr(w_packed), _ptr(bias), cs, n_out, kh, kw, stride, pad, mode, flags,
                                        rs, hr, wr, iarr(ext) if ext is not None else None, _ptr(stats_buf),
                                        _stream()), 'conv2d_gemm_bf16')

    if TIMER is not None:
        pix = s.shape[0] * (hw[0] * hw[1] if mode == 0 else s.shape[1] * s.shape[2])
        TIMER.time(f'conv_nt_bf16|m{mode} k{kh} s{stride} cs{cs} n{n_out} pix{pix} f{flags}',
                   2.0 * pix * n_out * kh * kw * cs, launch,
                   nbytes=2.0 * (s.numel() + outs[0].numel() * (2 if odt == torch.float32 else 1) + w_packed.numel()))
    else:
        launch()
    if stats:
        return outs, stats_buf
    return outs


SHARE_ALONE = (3, 2, 3, 2, 0, 3, 1, 0)


def _nt_family(n_out):
    """Timer family of a sacv_conv2d_gemm call: the tile shape the fp32 planner would take, or (a trailing 3) the
    128 x 64 split-operand kernel on the bf16 pipe (launch plan nt_split)."""
    if _share[7]:
        return 'conv_nt3_128x64'
    return 'conv_nt_128x128' if n_out > 64 else 'conv_nt_128x64'
_share = SHARE_ALONE


def current_share():
    """The launch plan in force (the tuple launch_share returns / accepts)."""
    return _share


def launch_share(nt64=0, nt128=0, wino=0, wgrad=0, cu_reserve=-1, split_policy=-1, wino_big=-1, nt_split=-1):
    """Launch plan of the following launches (include/sacv_hip.h: sacv_launch_share): workgroup slots per CU the
    persistent convolution kernels may plan for, CUs they leave free, whether remainder tiles are split along K.
    Returns the previous plan (a tuple that can be passed back as launch_share(*prev))."""
    global _share
    prev = _share
    cur = tuple(n if n > 0 else o for n, o in zip((nt64, nt128, wino, wgrad), prev[:4])) + \
        (cu_reserve if cu_reserve >= 0 else prev[4], split_policy if split_policy >= 0 else prev[5],
         wino_big if wino_big >= 0 else prev[6], nt_split if nt_split >= 0 else prev[7])
    check(lib.sacv_launch_share(*cur), 'launch_share')
    _share = cur
    return prev


def _wgrad_chunk(npix, cout, n2, taps):
    """Pixels per split of the weight-gradient GEMM (planned in the library: sacv_conv2d_wgrad_chunk)."""
    npix = [npix] if isinstance(npix, int) else list(npix)
    chunk = lib.sacv_conv2d_wgrad_chunk(len(npix), iarr(npix), cout, n2, taps)
    if chunk <= 0:
        raise ValueError('wgrad_chunk: bad problem list')
    return chunk


def _wgrad_chunk_bf16(npix, cout, n2, target_blocks=512):
    """Pixels per split of the bf16 weight-gradient GEMM (conv_tn_bf16_kernel: 128-wide tiles, 64 where a dimension is
    that narrow): tiles x splits within one co-resident wave of workgroups."""
    bm = 64 if cout <= 64 else 128
    bn = 64 if n2 <= 64 else 128
    tiles = -(-cout // bm) * -(-n2 // bn)
    chunks = max(1, min(target_blocks // tiles, -(-npix // 128)))
    chunk = -(-npix // chunks)
    return max(32, -(-chunk // 32) * 32)


_ws_cache = {}
WS_SUFFIX = ''  # set by a caller that enqueues on another stream (engine._on_side): that stream's own scratch buffers


def workspace(nbytes, device, tag='ws'):
    """Grow-only scratch buffer per tag (kernels on one stream reuse it in order)."""
    key = (tag + WS_SUFFIX, device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf


def conv_wgrad(dys, xs, dw, cin, kh, kw, stride, pad, accumulate=False, cin_pad=None, dw_taps=0):
    """dw (logical OIHW view into the gradient arena, stored OIHW or channels_last) (+)= sum over problems of
    x (*) dy."""
    n = len(dys)
    dw_layout = weight_layout(dw) if dw.ndim == 4 else 0
    cin_pad = cin_pad or cin
    cout = dys[0].shape[3]
    if dys[0].dtype == BF16:
        return _conv_wgrad_bf16(dys, xs, dw, cin, cin_pad, cout, kh, kw, stride, pad, accumulate, dw_taps, dw_layout)
    nb = [d.shape[0] for d in dys]
    ho = [d.shape[1] for d in dys]
    wo = [d.shape[2] for d in dys]
    chunk = _wgrad_chunk([b * h * w for b, h, w in zip(nb, ho, wo)], cout, kh * kw * cin_pad, kh * kw)
    nbytes = lib.sacv_conv2d_wgrad_workspace_bytes(n, iarr(nb), iarr(ho), iarr(wo), cin_pad, cout, kh, kw, chunk)
    ws = workspace(nbytes, dys[0].device, 'wgrad')
    def launch():
        check(lib.sacv_conv2d_wgrad(
            n, parr([d.data_ptr() for d in dys]), parr([x.data_ptr() for x in xs]), iarr(nb),
            iarr([x.shape[1] for x in xs]), iarr([x.shape[2] for x in xs]), iarr(ho), iarr(wo),
            _ptr(dw), ws.data_ptr(), cin, cin_pad, cout, kh, kw, stride, pad, chunk, int(accumulate), dw_taps,
            dw_layout, _stream()), 'conv2d_wgrad')

    if TIMER is not None:  # algorithmic flops of the weight gradient (+ its slab reduction, same event pair)
        taps = dw_taps if dw_taps else kh * kw
        flops = 2.0 * sum(b * h * w for b, h, w in zip(nb, ho, wo)) * cout * cin * taps
        TIMER.time(f'conv_tn{"3" if _share[7] else ""}|k{kh} s{stride} ci{cin} co{cout} np{n}', flops, launch)
    else:
        launch()
    return dw


def _conv_wgrad_bf16(dys, xs, dw, cin, cin_pad, cout, kh, kw, stride, pad, accumulate, dw_taps, dw_layout):
    if len(dys) != 1:
        raise ValueError('bf16 conv_wgrad: one segment per launch')
    dy, x = _chk(dys[0], 'dy', BF16), _chk(xs[0], 'x', BF16)
    nb, ho, wo = dy.shape[0], dy.shape[1], dy.shape[2]
    chunk = _wgrad_chunk_bf16(nb * ho * wo, cout, kh * kw * cin_pad)
    chunk = max(64, -(-chunk // 64) * 64)
    nbytes = lib.sacv_conv2d_wgrad_bf16_workspace_bytes(nb, ho, wo, cin_pad, cout, kh, kw, chunk)
    ws = workspace(nbytes, dy.device, 'wgrad')

    def launch():
        check(lib.sacv_conv2d_wgrad_bf16(dy.data_ptr(), x.data_ptr(), nb, x.shape[1], x.shape[2], ho, wo, _ptr(dw),
                                         ws.data_ptr(), cin, cin_pad, cout, kh, kw, stride, pad, chunk,
                                         int(accumulate), dw_taps, dw_layout, _stream()), 'conv2d_wgrad_bf16')

    if TIMER is not None:
        taps = dw_taps if dw_taps else kh * kw
        TIMER.time(f'conv_tn_bf16|k{kh} s{stride} ci{cin} co{cout}', 2.0 * nb * ho * wo * cout * cin * taps, launch,
                   nbytes=2.0 * (dy.numel() + x.numel()) + 4.0 * dw.numel())
    else:
        launch()
    return dw


def im2col_stem(x_nchw_logical, ho, wo, kh=7, kw=7, stride=2, pad=3, dtype=torch.float32):
    """x: logical [N,3,H,W] in ANY strides (NCHW-contiguous or channels_last) -> [N,ho,wo,STEM_KPAD]."""
    x = x_nchw_logical
    if x.device.type != 'cuda' or x.dtype != torch.float32:
        raise ValueError('im2col_stem: expected an fp32 CUDA(HIP) tensor')
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    if dtype == BF16:
        kpad = stem_kpad(kh, c, 64)
        col = torch.empty((n, ho, wo, kpad), device=x.device, dtype=BF16)
        check(lib.sacv_im2col_stem_bf16(_ptr(x), _ptr(col), n, c, h, w, sn, sc, sh, sw, ho, wo, kh, kw, stride, pad,
                                        kpad, _stream()), 'im2col_stem_bf16')
        return col
    kpad = stem_kpad(kh, c)
    col = torch.empty((n, ho, wo, kpad), device=x.device, dtype=torch.float32)
    check(lib.sacv_im2col_stem(_ptr(x), _ptr(col), n, c, h, w, sn, sc, sh, sw, ho, wo, kh, kw, stride, pad,
                               kpad, _stream()), 'im2col_stem')
    return col


def stem_conv(x_nchw_logical, wpanel, ho, wo, stats=False):
    """7x7/s2/p3 stem convolution without the im2col matrix: x logical [N,3,H,W] in any strides, wpanel the packed stem
    panel [64][Kpad] -> y [N,ho,wo,64] (and the per-tile BatchNorm partial sums [rows,2,64] when stats=True)."""
    x = x_nchw_logical
    if x.device.type != 'cuda' or x.dtype != torch.float32 or x.shape[1] != 3:
        raise ValueError('stem_conv: expected an fp32 CUDA(HIP) image batch [N,3,H,W]')
    n, _, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    y = torch.empty((n, ho, wo, 64), device=x.device, dtype=torch.float32)
    st = None
    if stats:
        st = torch.empty((lib.sacv_stem_conv_stats_rows(n, ho, wo), 2, 64), device=x.device, dtype=torch.float32)

    def launch():
        check(lib.sacv_stem_conv_fwd(_ptr(x), n, h, w, sn, sc, sh, sw, _ptr(_chk(wpanel, 'stem panel')), wpanel.shape[-1],
                                     _ptr(y), ho, wo, _ptr(st) if stats else None, _stream()), 'stem_conv_fwd')
    if TIMER is not None:
        pix = n * ho * wo
        TIMER.time(f'conv_nt_stem|m0 k7 s2 cs3 n64 np1 pix{pix} f{32 if stats else 0}', 2.0 * pix * 64 * 147, launch,
                   4.0 * (n * 3 * h * w + pix * 64 + 64 * 147))
    else:
        launch()
    return (y, st) if stats else y


def stem_conv_wgrad(x_nchw_logical, dy, dw, accumulate=False):
    """dw [64,3,7,7] (+)= weight gradient of the 7x7/s2/p3 stem convolution from the image (logical [N,3,H,W], any
    strides) and dy [N,ho,wo,64]: no im2col matrix (csrc/stem_conv.hip)."""
    x = x_nchw_logical
    if x.device.type != 'cuda' or x.dtype != torch.float32 or x.shape[1] != 3:
        raise ValueError('stem_conv_wgrad: expected an fp32 CUDA(HIP) image batch [N,3,H,W]')
    _chk(dy, 'stem_conv_wgrad dy')
    if tuple(dw.shape) != (64, 3, 7, 7) or not dw.is_contiguous():
        raise ValueError(f'stem_conv_wgrad: dw is {tuple(dw.shape)} (strides {dw.stride()}), expected contiguous [64,3,7,7]')
    n, _, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    ho, wo = dy.shape[1], dy.shape[2]
    ws = workspace(lib.sacv_stem_conv_wgrad_workspace_bytes(n, ho, wo), x.device, 'stem_wgrad')

    def launch():
        check(lib.sacv_stem_conv_wgrad(_ptr(x), n, h, w, sn, sc, sh, sw, _ptr(dy), ho, wo, _ptr(dw), ws.data_ptr(),
                                       int(accumulate), _stream()), 'stem_conv_wgrad')
    if TIMER is not None:
        TIMER.time('conv_tn_stem|k7 s2 ci3 co64 np1', 2.0 * n * ho * wo * 64 * 147, launch)
    else:
        launch()
    return dw


# ------------------------------------------------------------------------------------------ norm
def _bn_ws(r, c, device):
    return workspace(lib.sacv_bn_workspace_floats(r, c) * 4, device, 'bn')


def bn_train_stats(y, eps, momentum, running_mean, running_var, num_batches_tracked):
    c = y.shape[-1]
    r = y.numel() // c
    mean = torch.empty(c, device=y.device, dtype=torch.float32)
    invstd = torch.empty(c, device=y.device, dtype=torch.float32)
    check(lib.sacv_bn_train_stats(_ptr(y), r, c, eps, momentum, _ptr(mean), _ptr(invstd), _ptr(running_mean),
                                  _ptr(running_var), _ptr(num_batches_tracked), _bn_ws(r, c, y.device).data_ptr(),
                                  _stream()), 'bn_train_stats')
    return mean, invstd


def bn_finalize_stats(stats_buf, rows, eps, momentum, running_mean, running_var, num_batches_tracked):
    """mean / invstd (+ running statistics update) from the CF_STATS partials of the producing conv."""
    nblocks, _, c = stats_buf.shape
    mean = torch.empty(c, device=stats_buf.device, dtype=torch.float32)
    invstd = torch.empty(c, device=stats_buf.device, dtype=torch.float32)
    check(lib.sacv_bn_finalize_stats(_ptr(stats_buf), nblocks, rows, c, eps, momentum, _ptr(mean), _ptr(invstd),
                                     _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _stream()),
          'bn_finalize_stats')
    return mean, invstd


def colsum_finalize(stats_buf, out, accumulate=False):
    """out[c] (+)= sum over M tiles of the CF_STATS column sums (bias gradient)."""
    nblocks, _, c = stats_buf.shape
    check(lib.sacv_colsum_finalize(_ptr(stats_buf), nblocks, 2 * c, c, _ptr(out), int(accumulate), _stream()),
          'colsum_finalize')
    return out


def bn_apply(y, mean, invstd_or_var, gamma, beta, residual=None, relu=False, stat_is_var=False, eps=1e-5,
             out=None):
    z = out if out is not None else torch.empty_like(y)
    if y.dtype == BF16:
        check(lib.sacv_bn_apply_bf16(_ptr(y), _ptr(mean), _ptr(invstd_or_var), _ptr(gamma), _ptr(beta), _ptr(residual),
                                     _ptr(z), y.numel(), y.shape[-1], int(relu), int(stat_is_var), eps, _stream()),
              'bn_apply_bf16')
        return z
    check(lib.sacv_bn_apply(_ptr(y), _ptr(mean), _ptr(invstd_or_var), _ptr(gamma), _ptr(beta), _ptr(residual),
                            _ptr(z), y.numel(), y.shape[-1], int(relu), int(stat_is_var), eps, _stream()), 'bn_apply')
    return z


def bn_train_bwd(dz, zmask, y, mean, invstd, gamma, dgamma, dbeta, want_g=False, accumulate=False, dy_out=None,
                 relu_beta=None, partial=None):
    """relu_beta (the BN bias) with zmask=None: the ReLU mask is recomputed from y instead of read from z.
    partial: [rows, 2, C] first-stage sums from the epilogue of the data gradient that produced dz (conv_gemm_bnbwd)."""
    c = y.shape[-1]
    r = y.numel() // c
    dy = dy_out if dy_out is not None else torch.empty_like(y)
    g = torch.empty_like(y) if want_g else None
    if partial is not None:
        fn = lib.sacv_bn_train_bwd_partial_bf16 if y.dtype == BF16 else lib.sacv_bn_train_bwd_partial
        check(fn(_ptr(dz), _ptr(zmask), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(relu_beta), _ptr(dy), _ptr(g),
                 _ptr(dgamma), _ptr(dbeta), r, c, int(accumulate), _ptr(_chk(partial, 'bn partial sums')),
                 partial.shape[0], _bn_ws(r, c, y.device).data_ptr(), _stream()), 'bn_train_bwd_partial')
        return dy, g
    if y.dtype == BF16:
        check(lib.sacv_bn_train_bwd_bf16(_ptr(dz), _ptr(zmask), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                         _ptr(relu_beta), _ptr(dy), _ptr(g), _ptr(dgamma), _ptr(dbeta), r, c,
                                         int(accumulate), _bn_ws(r, c, y.device).data_ptr(), _stream()),
              'bn_train_bwd_bf16')
        return dy, g
    check(lib.sacv_bn_train_bwd(_ptr(dz), _ptr(zmask), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                _ptr(relu_beta), _ptr(dy), _ptr(g), _ptr(dgamma), _ptr(dbeta), r, c, int(accumulate),
                                _bn_ws(r, c, y.device).data_ptr(), _stream()), 'bn_train_bwd')
    return dy, g


def groupnorm_fwd(x, gamma, beta, groups=32, eps=1e-5, relu=True):
    """x: [N,H,W,C] -> (z, stats[N,G,2])."""
    n, h, w, c = x.shape
    z = torch.empty_like(x)
    stats = torch.empty((n, groups, 2), device=x.device, dtype=torch.float32)
    ws = workspace(lib.sacv_gn_workspace_floats(n, h * w, c) * 4, x.device, 'gn')
    check(lib.sacv_groupnorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(z), _ptr(stats), n, h * w, c, groups, eps,
                                 int(relu), ws.data_ptr(), _stream()), 'groupnorm_fwd')
    return z, stats


def groupnorm_bwd(x, dz, zmask, stats, gamma, dgamma, dbeta, groups=32, accumulate=False):
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    ws = workspace(lib.sacv_gn_workspace_floats(n, h * w, c) * 4, x.device, 'gn')
    check(lib.sacv_groupnorm_bwd(_ptr(x), _ptr(dz), _ptr(zmask), _ptr(stats), _ptr(gamma), _ptr(dx), _ptr(dgamma),
                                 _ptr(dbeta), n, h * w, c, groups, int(accumulate), ws.data_ptr(), _stream()),
          'groupnorm_bwd')
    return dx


# ------------------------------------------------------------------------------------------ pointwise
def maxpool_fwd(x, k=3, s=2, p=1, want_idx=True):
    n, h, w, c = x.shape
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=x.dtype)
    idx = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8) if want_idx else None
    if x.dtype == BF16:
        check(lib.sacv_maxpool_fwd_bf16(_ptr(x), _ptr(y), _ptr(idx), n, h, w, c, ho, wo, k, s, p, _stream()),
              'maxpool_fwd_bf16')
        return y, idx
    check(lib.sacv_maxpool_fwd(_ptr(x), _ptr(y), _ptr(idx), n, h, w, c, ho, wo, k, s, p, _stream()), 'maxpool_fwd')
    return y, idx


def maxpool_bwd(dy, idx, in_shape, k=3, s=2, p=1):
    n, h, w, c = in_shape
    dx = torch.empty(in_shape, device=dy.device, dtype=dy.dtype)
    if dy.dtype == BF16:
        check(lib.sacv_maxpool_bwd_bf16(_ptr(dy), _ptr(idx), _ptr(dx), n, h, w, c, dy.shape[1], dy.shape[2], k, s, p,
                                        _stream()), 'maxpool_bwd_bf16')
        return dx
    check(lib.sacv_maxpool_bwd(_ptr(dy), _ptr(idx), _ptr(dx), n, h, w, c, dy.shape[1], dy.shape[2], k, s, p,
                               _stream()), 'maxpool_bwd')
    return dx


def upsample_add_fwd(top, lateral):
    n, hs, ws_, c = top.shape
    _, hd, wd, _ = lateral.shape
    out = torch.empty_like(lateral)
    check(lib.sacv_upsample_bilinear_add_fwd(_ptr(top), _ptr(lateral), _ptr(out), n, hs, ws_, hd, wd, c, _stream()),
          'upsample_add_fwd')
    return out


def upsample_bwd(dout, dtop, accumulate):
    n, hs, ws_, c = dtop.shape
    _, hd, wd, _ = dout.shape
    check(lib.sacv_upsample_bilinear_bwd(_ptr(dout), _ptr(dtop), n, hs, ws_, hd, wd, c, int(accumulate), _stream()),
          'upsample_bwd')
    return dtop


EW_RELU, EW_RELU_BWD, EW_SIGMOID_BWD, EW_ADD, EW_SCALE, EW_SCALE_SIGMOID_BWD = 0, 1, 2, 3, 4, 5


def eltwise(op, a, b=None, out=None, scalar=1.0):
    out = out if out is not None else torch.empty_like(a)
    check(lib.sacv_eltwise(_ptr(a), _ptr(b), _ptr(out), a.numel(), op, scalar, _stream()), 'eltwise')
    return out


def colsum(x2d_rows, c, out, accumulate=False):
    r = x2d_rows.numel() // c
    ws = workspace(lib.sacv_colsum_workspace_floats(r, c) * 4, x2d_rows.device, 'colsum')
    check(lib.sacv_colsum(_ptr(x2d_rows), _ptr(out), r, c, int(accumulate), ws.data_ptr(), _stream()), 'colsum')
    return out


def fill(t, value):
    check(lib.sacv_fill(_ptr(t), t.numel(), value, _stream()), 'fill')
    return t


# ------------------------------------------------------------------------------------------ detection
def retina_anchors(feature_sizes, areas, ratios, scales, strides, device):
    """feature_sizes: [[W,H],...] -> [A_total,4] fp32 on device (bit-exact vs the NumPy reference)."""
    import numpy as np
    nlev = len(feature_sizes)
    a = len(ratios) * len(scales)
    total = sum(w * h * a for w, h in feature_sizes)
    out = torch.empty((total, 4), device=device, dtype=torch.float32)
    f32 = lambda v: [float(x) for x in np.asarray(v, dtype=np.float32).reshape(-1)]
    check(lib.sacv_retina_anchors(nlev, iarr([w for w, h in feature_sizes]), iarr([h for w, h in feature_sizes]),
                                  farr(f32(areas)), farr(f32(ratios)), len(ratios), farr(f32(scales)), len(scales),
                                  farr(f32(strides)), _ptr(out), _stream()), 'retina_anchors')
    return out


def fcos_points(feature_sizes, strides, device):
    total = sum(w * h for w, h in feature_sizes)
    out = torch.empty((total, 2), device=device, dtype=torch.float32)
    check(lib.sacv_fcos_points(len(feature_sizes), iarr([w for w, h in feature_sizes]),
                               iarr([h for w, h in feature_sizes]), farr(strides), _ptr(out), _stream()),
          'fcos_points')
    return out


def retina_assign(anchors, annots, smooth_l1=True):
    """-> (targets [B,A,5] fp32, counts int32[2] = (#positive, #label>=0))."""
    _chk(anchors, 'anchors')
    _chk(annots, 'annotations')
    b, g, _ = annots.shape
    a = anchors.shape[0]
    targets = torch.empty((b, a, 5), device=anchors.device, dtype=torch.float32)
    counts = torch.empty(2, device=anchors.device, dtype=torch.int32)
    check(lib.sacv_retina_assign(_ptr(anchors), _ptr(annots), _ptr(targets), _ptr(counts), b, a, g, int(smooth_l1),
                                 _stream()), 'retina_assign')
    return targets, counts


def fcos_assign(points, level_off, strides, mi, annots, radius=1.5, use_center_sample=True):
    """-> (targets [B,P,8], counts int32[2])."""
    _chk(points, 'points')
    _chk(annots, 'annotations')
    b, g, _ = annots.shape
    p = points.shape[0]
    targets = torch.empty((b, p, 8), device=points.device, dtype=torch.float32)
    counts = torch.empty(2, device=points.device, dtype=torch.int32)
    flat_mi = [float(v) for pair in mi for v in pair]
    check(lib.sacv_fcos_assign(len(strides), iarr(level_off), farr(strides), farr(flat_mi), _ptr(points),
                               _ptr(annots), _ptr(targets), _ptr(counts), b, p, g, radius, int(use_center_sample),
                               _stream()), 'fcos_assign')
    return targets, counts


def fcos_reg_ctr_loss(reg_levels, ctr_levels, targets, counts, iou_type, w_reg, w_ctr, loss_out, slot_reg, slot_ctr,
                      want_grad=True):
    b = reg_levels[0].shape[0]
    off = _level_off(reg_levels, 4)
    dreg = [torch.empty_like(t) for t in reg_levels] if want_grad else None
    dctr = [torch.empty_like(t) for t in ctr_levels] if want_grad else None
    ws = workspace(lib.sacv_loss_workspace_floats() * 4, targets.device, 'loss')
    check(lib.sacv_fcos_reg_ctr_loss_fwd_bwd(
        len(reg_levels), parr([_chk(t, 'reg').data_ptr() for t in reg_levels]),
        parr([_chk(t, 'ctr').data_ptr() for t in ctr_levels]),
        parr([g.data_ptr() for g in dreg]) if dreg else None, parr([g.data_ptr() for g in dctr]) if dctr else None,
        iarr(off), _ptr(targets), b, off[-1], iou_type, w_reg, w_ctr, _ptr(counts), ws.data_ptr(), _ptr(loss_out),
        slot_reg, slot_ctr, _stream()), 'fcos_reg_ctr_loss')
    return dreg, dctr


def fcos_split_fwd(y8, scale_elem):
    """y8 [B,h,w,8] -> (reg [B,h,w,4] = raw*exp(scale), ctr [B,h,w,1] = sigmoid)."""
    b, h, w, _ = y8.shape
    reg = torch.empty((b, h, w, 4), device=y8.device, dtype=torch.float32)
    ctr = torch.empty((b, h, w, 1), device=y8.device, dtype=torch.float32)
    check(lib.sacv_fcos_split_fwd(_ptr(y8), _ptr(scale_elem), _ptr(reg), _ptr(ctr), b * h * w, _stream()),
          'fcos_split_fwd')
    return reg, ctr


def fcos_split_bwd(dreg, dctr, reg, ctr, scale_elem, dscale, idx, gscale=1.0, accumulate=False):
    b, h, w, _ = reg.shape
    dy8 = torch.empty((b, h, w, 8), device=reg.device, dtype=torch.float32)
    ws = workspace(4096, reg.device, 'split')
    check(lib.sacv_fcos_split_bwd(_ptr(dreg), _ptr(dctr), _ptr(reg), _ptr(ctr), _ptr(scale_elem), gscale, _ptr(dy8),
                                  _ptr(dscale), idx, int(accumulate), b * h * w, ws.data_ptr(), _stream()),
          'fcos_split_bwd')
    return dy8


def _level_off(levels, per_row):
    off = [0]
    for t in levels:
        off.append(off[-1] + t.numel() // (t.shape[0] * per_row))
    return off


def focal_loss(cls_levels, targets, counts, alpha, gamma, weight, loss_out, slot, want_grad=True, tstride=5):
    """cls_levels: list of [B, ..., C] probability tensors.  Returns list of grad tensors (or None)."""
    b = cls_levels[0].shape[0]
    c = cls_levels[0].shape[-1]
    off = _level_off(cls_levels, c)
    grads = [torch.empty_like(t) for t in cls_levels] if want_grad else None
    ws = workspace(lib.sacv_loss_workspace_floats() * 4, targets.device, 'loss')
    check(lib.sacv_focal_loss_fwd_bwd(
        len(cls_levels), parr([_chk(t, 'cls').data_ptr() for t in cls_levels]),
        parr([g.data_ptr() for g in grads]) if grads else None, iarr(off), _ptr(targets), tstride, b, off[-1], c,
        alpha, gamma, weight, _ptr(counts), ws.data_ptr(), _ptr(loss_out), slot, _stream()), 'focal_loss')
    return grads


BOX_SMOOTH_L1, BOX_IOU, BOX_GIOU, BOX_DIOU, BOX_CIOU, BOX_EIOU = 0, 1, 2, 3, 4, 5


def retina_box_loss(reg_levels, targets, anchors, counts, box_type, beta, weight, loss_out, slot, want_grad=True):
    b = reg_levels[0].shape[0]
    off = _level_off(reg_levels, 4)
    grads = [torch.empty_like(t) for t in reg_levels] if want_grad else None
    ws = workspace(lib.sacv_loss_workspace_floats() * 4, targets.device, 'loss')
    check(lib.sacv_retina_box_loss_fwd_bwd(
        len(reg_levels), parr([_chk(t, 'reg').data_ptr() for t in reg_levels]),
        parr([g.data_ptr() for g in grads]) if grads else None, iarr(off), _ptr(targets), _ptr(anchors), b, off[-1],
        box_type, beta, weight, _ptr(counts), ws.data_ptr(), _ptr(loss_out), slot, _stream()), 'retina_box_loss')
    return grads


def decode_topk_nms(kind, cls_levels, reg_levels, ctr_levels, geom, min_score, topn, nms_type, nms_thr, max_obj,
                    soft=None):
    """soft = None: hard NMS (nms_type 0/1/2); soft = (method, sigma, prune_score): batched Soft-NMS instead."""
    b = cls_levels[0].shape[0]
    c = cls_levels[0].shape[-1]
    off = _level_off(reg_levels, 4)
    a = off[-1]
    dev = geom.device
    ws = workspace(lib.sacv_decode_workspace_bytes(b, a, topn), dev, 'decode')
    out_s = torch.empty((b, max_obj), device=dev, dtype=torch.float32)
    out_c = torch.empty((b, max_obj), device=dev, dtype=torch.float32)
    out_b = torch.empty((b, max_obj, 4), device=dev, dtype=torch.float32)
    if soft is not None:
        method, sigma, prune = soft
        check(lib.sacv_decode_topk_soft_nms(
            kind, len(cls_levels), parr([_chk(t, 'cls').data_ptr() for t in cls_levels]),
            parr([_chk(t, 'reg').data_ptr() for t in reg_levels]),
            parr([_chk(t, 'ctr').data_ptr() for t in ctr_levels]) if ctr_levels else None, iarr(off), _ptr(geom), b, a,
            c, min_score, topn, method, nms_thr, sigma, prune, max_obj, ws.data_ptr(), _ptr(out_s), _ptr(out_c),
            _ptr(out_b), _stream()), 'decode_topk_soft_nms')
        return out_s, out_c, out_b
    check(lib.sacv_decode_topk_nms(
        kind, len(cls_levels), parr([_chk(t, 'cls').data_ptr() for t in cls_levels]),
        parr([_chk(t, 'reg').data_ptr() for t in reg_levels]),
        parr([_chk(t, 'ctr').data_ptr() for t in ctr_levels]) if ctr_levels else None, iarr(off), _ptr(geom), b, a, c,
        min_score, topn, nms_type, nms_thr, max_obj, ws.data_ptr(), _ptr(out_s), _ptr(out_c), _ptr(out_b),
        _stream()), 'decode_topk_nms')
    return out_s, out_c, out_b


def adamw_step(params, grads, exp_avg, exp_avg_sq, split, lr, beta1, beta2, eps, wd_lo, wd_hi, step,
               grad_scale=1.0):
    check(lib.sacv_adamw_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(), split,
                              lr, beta1, beta2, eps, wd_lo, wd_hi, step, grad_scale, _stream()), 'adamw_step')


# ------------------------------------------------------------------------------------------ data boundary
def collate_u8(images_u8, resize, canvas=None):
    """images_u8: list of uint8 CUDA tensors [h,w,3] -> logical [B,3,R,R] fp32 tensor in channels_last storage;
    canvas = (Hc, Wc): the rectangular padding-skip canvas [B,3,Hc,Wc] instead."""
    b = len(images_u8)
    dev = images_u8[0].device
    for t in images_u8:
        if t.dtype != torch.uint8 or t.device.type != 'cuda' or not t.is_contiguous() or t.shape[-1] != 3:
            raise ValueError('collate_u8: expected contiguous uint8 CUDA tensors [h,w,3]')
    if canvas is not None:
        hc, wc = canvas
        out = torch.empty((b, hc, wc, 3), device=dev, dtype=torch.float32)
        check(lib.sacv_collate_u8_rect(b, parr([t.data_ptr() for t in images_u8]),
                                       iarr([t.shape[0] for t in images_u8]), iarr([t.shape[1] for t in images_u8]),
                                       hc, wc, _ptr(out), _stream()), 'collate_u8_rect')
        return out.permute(0, 3, 1, 2)
    out = torch.empty((b, resize, resize, 3), device=dev, dtype=torch.float32)
    check(lib.sacv_collate_u8(b, parr([t.data_ptr() for t in images_u8]), iarr([t.shape[0] for t in images_u8]),
                              iarr([t.shape[1] for t in images_u8]), resize, _ptr(out), _stream()), 'collate_u8')
    return out.permute(0, 3, 1, 2)


def eval_postprocess(boxes, scales, sizes):
    """In place on boxes [B,M,4] (fp32 CUDA): / scale, clip to sizes [B,2] = (h,w), xyxy -> xywh."""
    _chk(boxes, 'boxes')
    b, m, _ = boxes.shape
    check(lib.sacv_eval_postprocess(_ptr(boxes), _ptr(_chk(scales, 'scales')), _ptr(_chk(sizes, 'sizes')), b, m,
                                    _stream()), 'eval_postprocess')
    return boxes


# ------------------------------------------------------------------------------------------ classification
def avgpool_fwd(x):
    n, h, w, c = x.shape
    out = torch.empty((n, 1, 1, c), device=x.device, dtype=x.dtype)
    if x.dtype == BF16:
        check(lib.sacv_avgpool_fwd_bf16(_ptr(x), _ptr(out), n, h * w, c, _stream()), 'avgpool_fwd_bf16')
        return out
    check(lib.sacv_avgpool_fwd(_ptr(x), _ptr(out), n, h * w, c, _stream()), 'avgpool_fwd')
    return out


def avgpool_bwd(dy, x_shape):
    n, h, w, c = x_shape
    dx = torch.empty(tuple(x_shape), device=dy.device, dtype=dy.dtype)
    if dy.dtype == BF16:
        check(lib.sacv_avgpool_bwd_bf16(_ptr(dy), _ptr(dx), n, h * w, c, _stream()), 'avgpool_bwd_bf16')
        return dx
    check(lib.sacv_avgpool_bwd(_ptr(dy), _ptr(dx), n, h * w, c, _stream()), 'avgpool_bwd')
    return dx


def softmax_ce(logits, labels, want_grad=True, grad_scale=1.0):
    """logits [N,C] fp32, labels [N] int64 -> (loss f32[1], dlogits or None)."""
    _chk(logits, 'logits')
    n, c = logits.shape
    if labels.dtype != torch.int64 or labels.device != logits.device:
        raise ValueError('softmax_ce: labels must be int64 on the same device')
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dl = torch.empty_like(logits) if want_grad else None
    ws = workspace(((n + 3) // 4 + 64) * 4, logits.device, 'ce')
    check(lib.sacv_softmax_ce_fwd_bwd(_ptr(logits), _ptr(labels.contiguous()), _ptr(dl), _ptr(loss), n, c,
                                      grad_scale, ws.data_ptr(), _stream()), 'softmax_ce')
    return loss, dl


def ema_update(ema_flat, model_flat, decay, gate=None):
    """ema = decay * ema + (1. - decay) * model over two flat fp32 blocks of the same layout, with torch's roundings of
    the two Python-float scalars (tools/utils.py:156)."""
    import numpy as np
    if ema_flat.numel() != model_flat.numel():
        raise ValueError('ema_update: the two blocks differ in size (wrong ema model!)')
    check(lib.sacv_ema_update(_ptr(_chk(ema_flat, 'ema block')), _ptr(_chk(model_flat, 'model block')), ema_flat.numel(),
                              float(np.float32(decay)), float(np.float32(1. - decay)), _ptr(gate), _stream()),
          'ema_update')


def sgd_step(params, grads, momentum_buf, split, lr, momentum, wd_lo, wd_hi, first_step, grad_scale=1.0):
    check(lib.sacv_sgd_step(_ptr(params), _ptr(grads), _ptr(momentum_buf), params.numel(), split, lr, momentum,
                            wd_lo, wd_hi, int(first_step), grad_scale, _stream()), 'sgd_step')


class BnFold:
    """(shift, scale) of the eval-mode BatchNorm layers of a model, recomputed by ONE launch per forward pass
    (sacv_bn_fold_batched) and handed to the producing convolutions as their SACV_CF_AFFINE operand.  `layers`: list of
    (key, gamma, beta, running_mean, running_var); `get(key)` -> the [2][C] tensor."""

    def __init__(self, layers, eps):
        import numpy as np
        self.dev = layers[0][1].device
        self.eps = eps
        total = sum(2 * g.numel() for _, g, _, _, _ in layers)
        self.buf = torch.empty(total, device=self.dev, dtype=torch.float32)
        recs = np.zeros(len(layers), dtype=np.dtype([('g', '<u8'), ('b', '<u8'), ('m', '<u8'), ('v', '<u8'),
                                                     ('o', '<u8'), ('c', '<i4'), ('pad', '<i4')]))
        assert recs.dtype.itemsize == 48
        self.views, self._keep, off = {}, [], 0
        for i, (key, g, b, m, v) in enumerate(layers):
            c = g.numel()
            out = self.buf[off:off + 2 * c].view(2, c)
            off += 2 * c
            recs[i] = (g.data_ptr(), b.data_ptr(), m.data_ptr(), v.data_ptr(), out.data_ptr(), c, 0)
            self.views[key] = out
            self._keep += [g, b, m, v]
        self.ptrs = [t.data_ptr() for t in self._keep]
        self.n = len(layers)
        self.descs = torch.from_numpy(recs.view(np.uint8).copy()).to(self.dev)

    def is_current(self):
        return all(t.data_ptr() == p for t, p in zip(self._keep, self.ptrs))

    def refresh(self):
        check(lib.sacv_bn_fold_batched(self.descs.data_ptr(), self.n, self.eps, _stream()), 'bn_fold_batched')

    def get(self, key):
        return self.views[key]


class DgradPanels:
    """Data-gradient panels ([Cin][KH*KW*Cout]) of a set of conv weights, refreshed by ONE kernel launch per step
    (the per-layer pack launches were 0.4 ms of a 42 ms step).  Buffers and the descriptor table are built once;
    `get(w)` returns the panel of a weight by its storage address."""

    def __init__(self, weights):
        import numpy as np
        self.dev = weights[0].device
        self.panels = {}
        recs = np.zeros(len(weights), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('co', '<i4'), ('ci', '<i4'),
                                                      ('t', '<i4'), ('ohwi', '<i4')]))
        assert recs.dtype.itemsize == 32
        cmap = []
        self._keep = list(weights)
        for i, w in enumerate(weights):
            co, ci, kh, kw = w.shape
            out = torch.empty((ci, kh * kw * co), device=self.dev, dtype=torch.float32)
            self.panels[w.data_ptr()] = out
            recs[i] = (w.data_ptr(), out.data_ptr(), co, ci, kh * kw, weight_layout(w))
            nchunk = kh * kw * ((ci + 31) // 32) * ((co + 31) // 32)  # 32x32 (co, ci) tiles per tap
            cmap.append(np.stack([np.full(nchunk, i, dtype=np.int32), np.arange(nchunk, dtype=np.int32)], axis=1))
        cmap = np.concatenate(cmap)
        self.nchunks = int(cmap.shape[0])
        self.descs = torch.from_numpy(recs.view(np.uint8).copy()).to(self.dev)
        self.cmap = torch.from_numpy(cmap.copy()).to(self.dev)
        self.ptrs = [w.data_ptr() for w in weights]

    def is_current(self):
        return all(w.data_ptr() == p for w, p in zip(self._keep, self.ptrs))

    def refresh(self):
        check(lib.sacv_pack_dgrad_batched(self.descs.data_ptr(), self.cmap.data_ptr(), self.nchunks, _stream()),
              'pack_dgrad_batched')

    def get(self, w):
        return self.panels.get(w.data_ptr())


class HalfPanels:
    """bf16 GEMM panels of the fp32 master weights, refreshed by TWO launches per step: the forward panels are ONE
    elementwise cast of the >=2-D section of the parameter arena (k x k weights are stored channels_last there and 1x1 /
    FC weights are their own panels, so a weight's forward panel is a view at the same offset), the data-gradient
    panels [Cin][taps*Cout] one batched transpose-cast.  The stem (Cin = 3, OIHW) is packed separately."""

    def __init__(self, arena, weights, stem_weight=None):
        import numpy as np
        self.arena = arena
        dev = arena.params.device
        self.base = arena.params.data_ptr()
        self.flat = torch.empty(arena.numel - arena.split, device=dev, dtype=BF16)
        self.stem_weight = stem_weight
        self.stem_panel = None
        self.dpanels = {}
        recs = np.zeros(len(weights), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('co', '<i4'), ('ci', '<i4'),
                                                      ('t', '<i4'), ('ohwi', '<i4')]))
        cmap = []
        self._keep = list(weights)
        for i, w in enumerate(weights):
            w4 = w if w.ndim == 4 else w.view(w.shape[0], w.shape[1], 1, 1)
            co, ci, kh, kw = w4.shape
            out = torch.empty((ci, kh * kw * co), device=dev, dtype=BF16)
            self.dpanels[w.data_ptr()] = out
            recs[i] = (w.data_ptr(), out.data_ptr(), co, ci, kh * kw, weight_layout(w4))
            nchunk = kh * kw * ((ci + 31) // 32) * ((co + 31) // 32)
            cmap.append(np.stack([np.full(nchunk, i, dtype=np.int32), np.arange(nchunk, dtype=np.int32)], axis=1))
        cmap = np.concatenate(cmap)
        self.nchunks = int(cmap.shape[0])
        self.descs = torch.from_numpy(recs.view(np.uint8).copy()).to(dev)
        self.cmap = torch.from_numpy(cmap.copy()).to(dev)
        self.ptrs = [w.data_ptr() for w in weights]

    def is_current(self):
        return self.arena.params.data_ptr() == self.base and all(w.data_ptr() == p for w, p in zip(self._keep, self.ptrs))

    def refresh(self):
        a = self.arena
        cast_bf16(a.params[a.split:], self.flat)
        check(lib.sacv_pack_dgrad_batched_bf16(self.descs.data_ptr(), self.cmap.data_ptr(), self.nchunks, _stream()),
              'pack_dgrad_batched_bf16')
        if self.stem_weight is not None:
            w = self.stem_weight
            kpad = stem_kpad(w.shape[2], w.shape[1], 64)
            self.stem_panel = cast_bf16(pack_weight(w, PACK_STEM, kpad=kpad))

    def fwd(self, w):
        """bf16 forward panel [Cout][taps*Cin] of a master weight that is stored as its own panel."""
        off = (w.data_ptr() - self.base) // 4 - self.arena.split
        n = w.numel()
        if off < 0 or off + n > self.flat.numel():
            raise ValueError('HalfPanels.fwd: not a >=2-D tensor of this arena')
        return self.flat[off:off + n].view(w.shape[0], n // w.shape[0])

    def dgrad(self, w):
        return self.dpanels.get(w.data_ptr())


# ------------------------------------------------------------------------------------------ step gate
STATE_FLOATS = 8  # {steps taken, skip, 1-beta1^t, sqrt(1-beta2^t), clip multiplier, grad norm, 0, 0}


def nonfinite_flag(x, flag, nan_only=False):
    """flag[0] = 1 when x (any float32 CUDA tensor) holds an inf/nan (nan_only: a nan); never lowers the flag."""
    if x.dtype != torch.float32:
        x = x.float()
    # any dense storage order will do (the scan is order-free): no copy of a channels_last batch
    dense = x.is_contiguous() or (x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last))
    if not dense or x.data_ptr() % 16:
        x = x.contiguous().clone()
    check(lib.sacv_nonfinite_flag(_ptr(x), x.numel(), int(nan_only), _ptr(flag), _stream()), 'nonfinite_flag')
    return flag


def loss_skip_flag(values, flag=None):
    """values: f32[k] loss terms -> f32[1] = 1 when their sum is 0 or any is inf/nan (tools/scripts.py:918-930).
    With `flag` given the result is OR-ed into it (input guard / gradient guard share the flag)."""
    acc = flag is not None
    if flag is None:
        flag = torch.empty(1, device=values.device, dtype=torch.float32)
    check(lib.sacv_loss_skip_flag(_ptr(values), values.numel(), _ptr(flag), int(acc), _stream()), 'loss_skip_flag')
    return flag


def step_gate(flag, state, beta1=0.9, beta2=0.999):
    check(lib.sacv_step_gate(_ptr(flag), _ptr(state), beta1, beta2, _stream()), 'step_gate')


def grad_clip_coef(grads, state, grad_scale=1.0, clip_value=0.0, max_norm=0.0):
    """state[5] = ||clamp(grads*grad_scale, +-clip_value)||, state[4] = min(1, max_norm/(norm+1e-6))."""
    ws = workspace(lib.sacv_grad_clip_workspace_bytes(), grads.device, 'clip')
    check(lib.sacv_grad_clip_coef(_ptr(grads), grads.numel(), grad_scale, clip_value, max_norm, _ptr(state),
                                  ws.data_ptr(), _stream()), 'grad_clip_coef')


def adamw_step_gated(params, grads, exp_avg, exp_avg_sq, split, lr, beta1, beta2, eps, wd_lo, wd_hi, state,
                     grad_scale=1.0, clip_value=0.0, lr_hi=None):
    """lr = learning rate of the 1-D group, lr_hi (default: the same) of the >=2-D group."""
    check(lib.sacv_adamw_step_gated(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(), split,
                                    lr, lr if lr_hi is None else lr_hi, beta1, beta2, eps, wd_lo, wd_hi, _ptr(state),
                                    grad_scale, clip_value, _stream()), 'adamw_step_gated')


def sgd_step_gated(params, grads, momentum_buf, split, lr, momentum, wd_lo, wd_hi, state, grad_scale=1.0,
                   clip_value=0.0, lr_hi=None, nesterov=False):
    check(lib.sacv_sgd_step_gated(_ptr(params), _ptr(grads), _ptr(momentum_buf), params.numel(), split, lr,
                                  lr if lr_hi is None else lr_hi, momentum, wd_lo, wd_hi, int(nesterov), _ptr(state),
                                  grad_scale, clip_value, _stream()), 'sgd_step_gated')
