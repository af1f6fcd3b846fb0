"""Oracle (test infrastructure): RetinaNet anchors and FCOS points in NumPy.

Restates simpleAICV/detection/models/anchor.py:5-86 (RetinaAnchors) and :89-130
(FCOSPositions).  All arithmetic is spelled out in explicit float32 so the result
does not depend on NumPy's scalar promotion rules (SURVEY.md finding 4: the
reference's `np.float32 * python_float` rounds in fp32 under NumPy >= 2, which is
the behaviour of the container the goldens were generated in).
"""
import math

import numpy as np

F32 = np.float32

DEFAULT_AREAS = [[32, 32], [64, 64], [128, 128], [256, 256], [512, 512]]
DEFAULT_RATIOS = [0.5, 1, 2]
DEFAULT_SCALES = [2**0, 2**(1.0 / 3.0), 2**(2.0 / 3.0)]
DEFAULT_STRIDES = [8, 16, 32, 64, 128]


def base_anchors(area, scales, ratios):
    """[R*S, 4] fp32 anchors centred on the origin for one level.

    anchor.py:36-57.  Row order is ratio-major, scale-minor.  aspect_w =
    f32(scale) * f32(sqrt(f64(ratio))), aspect_h = f32(scale) *
    f32(sqrt(f64(f32(1)/ratio))), both products rounded once in fp32.
    """
    area = np.asarray(area, dtype=F32)
    out = np.zeros((len(ratios) * len(scales), 4), dtype=F32)
    row = 0
    for r in np.asarray(ratios, dtype=F32):
        sq_w = F32(math.sqrt(float(r)))
        sq_h = F32(math.sqrt(float(F32(1.0) / r)))
        for s in np.asarray(scales, dtype=F32):
            w = area[0] * (s * sq_w)
            h = area[1] * (s * sq_h)
            half_w = w / F32(2.0)
            half_h = h / F32(2.0)
            out[row] = (-half_w, -half_h, half_w, half_h)
            row += 1
    return out


def retina_anchors_per_level(feature_sizes,
                             areas=DEFAULT_AREAS,
                             ratios=DEFAULT_RATIOS,
                             scales=DEFAULT_SCALES,
                             strides=DEFAULT_STRIDES):
    """List of [H, W, A, 4] fp32 arrays; feature_sizes[l] = [W, H].

    anchor.py:18-34, 59-86.  anchor = base (+) shift with ONE fp32 rounding,
    shift = (i + 0.5) * stride (exact in fp32 for the sizes used).
    """
    out = []
    for lvl, (fw, fh) in enumerate(feature_sizes):
        base = base_anchors(areas[lvl], scales, ratios)
        stride = float(strides[lvl])
        sx = ((np.arange(fw, dtype=np.float64) + 0.5) * stride).astype(F32)
        sy = ((np.arange(fh, dtype=np.float64) + 0.5) * stride).astype(F32)
        shift = np.zeros((fh, fw, 1, 4), dtype=F32)
        shift[..., 0, 0] = sx[None, :]
        shift[..., 0, 2] = sx[None, :]
        shift[..., 0, 1] = sy[:, None]
        shift[..., 0, 3] = sy[:, None]
        out.append(np.ascontiguousarray(base[None, None] + shift, dtype=F32))
    return out


def retina_anchors(feature_sizes, **kw):
    """[sum_l H*W*A, 4] fp32; flat index ((y*W + x)*A + a), levels P3..P7."""
    return np.concatenate(
        [a.reshape(-1, 4) for a in retina_anchors_per_level(feature_sizes, **kw)],
        axis=0)


def fcos_positions_per_level(feature_sizes, strides=DEFAULT_STRIDES):
    """List of [H, W, 2] fp32 (x_center, y_center).  anchor.py:94-130."""
    out = []
    for (fw, fh), stride in zip(feature_sizes, strides):
        sx = ((np.arange(fw, dtype=np.float64) + 0.5) * float(stride)).astype(F32)
        sy = ((np.arange(fh, dtype=np.float64) + 0.5) * float(stride)).astype(F32)
        pos = np.zeros((fh, fw, 2), dtype=F32)
        pos[..., 0] = sx[None, :]
        pos[..., 1] = sy[:, None]
        out.append(pos)
    return out


def fcos_positions(feature_sizes, strides=DEFAULT_STRIDES):
    return np.concatenate(
        [p.reshape(-1, 2) for p in fcos_positions_per_level(feature_sizes, strides)],
        axis=0)
