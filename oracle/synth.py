"""Oracle (test infrastructure): portable deterministic synthetic inputs.

A counter-based generator (splitmix64 finaliser over seed/index) in pure NumPy,
so fixtures generated in the build container can be regenerated bit-identically
on the GPU box without shipping big arrays and without depending on torch's RNG.
Shapes follow SURVEY.md 8(d): images U[0,1) [B,3,H,W]; annots [B,100,5] with K
boxes per image, w,h ~ U(40,400)*s, classes U{0..79}, remaining rows -1.
"""
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def uniform(seed, shape, lo=0.0, hi=1.0):
    """fp32 U[lo,hi) array; value i depends only on (seed, i)."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64)
        h = _mix(idx ^ _mix(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(1)))
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, shape, std=1.0):
    """fp32 N(0, std^2) via Box-Muller on two uniform streams (fp64 math, fp32 result)."""
    n = int(np.prod(shape))
    u1 = uniform(seed * 2 + 1, (n,)).astype(np.float64)
    u2 = uniform(seed * 2 + 2, (n,)).astype(np.float64)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
    return (z * std).astype(np.float32).reshape(shape)


def images(seed, batch, height, width):
    return uniform(seed, (batch, 3, height, width))


def uint8_image(seed, height, width):
    """[H,W,3] uint8 RGB image (what cv2.imread + cvtColor hand the dataset pipeline)."""
    return np.floor(uniform(seed, (height, width, 3)).astype(np.float64) * 256.0).astype(np.uint8)


def annotations(seed, batch, height, width, boxes_per_image=8, max_objects=100,
                num_classes=80, size_lo=40.0, size_hi=400.0):
    """[B, max_objects, 5] fp32 (x1,y1,x2,y2,cls); padding rows are -1 (common.py:270-276)."""
    ann = np.full((batch, max_objects, 5), -1, dtype=np.float32)
    k = boxes_per_image
    if k == 0:
        return ann
    size_hi = min(size_hi, 0.9 * min(height, width))
    size_lo = min(size_lo, 0.5 * size_hi)
    r = uniform(seed + 7919, (batch, k, 5)).astype(np.float64)
    w = size_lo + (size_hi - size_lo) * r[..., 0]
    h = size_lo + (size_hi - size_lo) * r[..., 1]
    x1 = r[..., 2] * (width - w)
    y1 = r[..., 3] * (height - h)
    cls = np.floor(r[..., 4] * num_classes)
    ann[:, :k, 0] = x1
    ann[:, :k, 1] = y1
    ann[:, :k, 2] = x1 + w
    ann[:, :k, 3] = y1 + h
    ann[:, :k, 4] = cls
    return ann


def feature_sizes(height, width, levels=5):
    """[[W,H]] per FPN level P3..P7 for an input HxW (conv arithmetic of the backbone/FPN)."""
    def down(v):  # 3x3 (or 7x7 p3 / 1x1) stride-2 with "same" padding
        return (v - 1) // 2 + 1
    h, w = down(down(height)), down(down(width))  # stem + maxpool -> C2
    out = []
    for _ in range(levels):
        h, w = down(h), down(w)
        out.append([w, h])
    return out
