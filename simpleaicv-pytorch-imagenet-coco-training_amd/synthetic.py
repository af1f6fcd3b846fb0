"""Synthetic COCO-shaped batches for bench.py: images U[0,1) [B,3,H,W]; annotations
[B,100,5] = (x1,y1,x2,y2,cls) with K boxes per image, w,h ~ U(40,400), classes U{0..79}, padding rows -1
(the DetectionCollater contract, simpleAICV/detection/common.py:256-286).

A counter-based generator (splitmix64 finaliser over (seed, index)) in NumPy: every rank builds its own
batch from `seed = rank` without any torch RNG state.  (tests/ use the same recipe from oracle/synth.py;
tests/test_abi.py checks that the two agree.)
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def uniform(seed, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64)
        h = _mix(idx ^ _mix(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(1)))
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def images(seed, batch, height, width):
    return uniform(seed, (batch, 3, height, width))


def annotations(seed, batch, height, width, boxes_per_image=8, max_objects=100, num_classes=80, size_lo=40.0,
                size_hi=400.0):
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
    ann[:, :k, 0] = x1
    ann[:, :k, 1] = y1
    ann[:, :k, 2] = x1 + w
    ann[:, :k, 3] = y1 + h
    ann[:, :k, 4] = np.floor(r[..., 4] * num_classes)
    return ann


def pyramid_sizes(height, width, levels=5):
    """(w, h) of the P3..P7 maps: five stride-2 "same" reductions from the image, the first three inside the backbone."""
    half = lambda v: (v + 1) // 2
    w, h = half(half(half(width))), half(half(half(height)))
    out = []
    for _ in range(levels):
        out.append((w, h))
        w, h = half(w), half(h)
    return out


def head_outputs(seed, batch, height, width, fcos=False, num_classes=80, hot_every=29):
    """Head outputs of an evaluation batch for the decode + NMS benchmark: per level cls [B,h,w,(9,)C] with every score
    below the decoder's 0.05 threshold except one class of every `hot_every`-th anchor, whose score is distinct in
    (0.06, 0.99) (6 900 candidates per 800x1333 image: the top-1000 selection and the 1000-box NMS both have real work);
    reg ~ N(0, 0.3) (RetinaNet deltas) or positive distances (FCOS) + centerness in (0.25, 1)."""
    a = 1 if fcos else 9
    cls, reg, ctr = [], [], []
    first = 0
    for lvl, (fw, fh) in enumerate(pyramid_sizes(height, width)):
        n = fw * fh * a
        # background scores: below the threshold also after FCOS's sqrt(score * centerness) (sqrt(0.002) < 0.05)
        c = uniform(seed + 31 * lvl + 1, (batch, n, num_classes), 0.0, 0.002 if fcos else 0.04)
        for b in range(batch):
            idx = np.arange(first, first + n, dtype=np.int64)
            hot = idx[(idx * 5 + b) % hot_every == 0]
            order = (hot * 48271 + b * 69621) % 2147483647  # distinct for distinct anchors (prime modulus)
            c[b, hot - first, (hot * 11 + b * 3) % num_classes] = (0.06 + 0.93 * order / 2147483647.0).astype(np.float32)
        first += n
        cls.append(c.reshape((batch, fh, fw, num_classes) if fcos else (batch, fh, fw, a, num_classes)))
        if fcos:
            reg.append(np.abs(uniform(seed + 31 * lvl + 2, (batch, fh, fw, 4), -1.0, 1.0)) * 2.0 + np.float32(2.0 + 0.7 * lvl))
            ctr.append(uniform(seed + 31 * lvl + 3, (batch, fh, fw, 1), 0.25, 1.0))
        else:
            reg.append(uniform(seed + 31 * lvl + 2, (batch, fh, fw, a, 4), -0.5, 0.5))
    return (cls, reg, ctr) if fcos else (cls, reg)
