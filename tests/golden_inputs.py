"""Input generators shared by the oracle tests and the GPU parity tests.

These are the SAME recipes oracle/make_golden.py fed to the reference (that file
cannot be imported on the GPU box because it pulls in /root/reference), kept in
sync by tests/test_oracle_vs_golden.py which reproduces the reference outputs
from them.
"""
import numpy as np

from oracle import synth

S1 = (800, 1333)
S2 = (1333, 1333)
TINY = (96, 128)
MID = (256, 320)
SHAPES = dict(S1=S1, S2=S2, TINY=TINY, MID=MID)


def synth_preds_retina(seed, B, H, W, num_classes=80, A=9):
    fs = synth.feature_sizes(H, W)
    cls, reg = [], []
    for l, (fw, fh) in enumerate(fs):
        c = synth.uniform(seed + 10 * l + 1, (B, fh, fw, A, num_classes))
        c = (c**3).astype(np.float32)
        flat = c.reshape(-1)
        flat[0::9973] = 0.0
        flat[1::9973] = 1.0
        flat[2::9973] = 5e-5
        flat[3::9973] = 1.0 - 5e-5
        cls.append(c)
        reg.append(synth.normal(seed + 10 * l + 2, (B, fh, fw, A, 4), std=0.3))
    return cls, reg


def synth_preds_fcos(seed, B, H, W, num_classes=80):
    fs = synth.feature_sizes(H, W)
    cls, reg, ctr = [], [], []
    for l, (fw, fh) in enumerate(fs):
        c = (synth.uniform(seed + 10 * l + 1, (B, fh, fw, num_classes))**3).astype(np.float32)
        flat = c.reshape(-1)
        flat[0::9973] = 0.0
        flat[1::9973] = 1.0
        cls.append(c)
        reg.append(synth.normal(seed + 10 * l + 2, (B, fh, fw, 4), std=0.5) + np.float32(3.0 + 0.7 * l))
        t = synth.uniform(seed + 10 * l + 3, (B, fh, fw, 1))
        t.reshape(-1)[0::997] = 0.0
        t.reshape(-1)[1::997] = 1.0
        ctr.append(t)
    return cls, reg, ctr


def synth_preds_decode(seed, B, H, W, fcos=False):
    """Predictions whose above-threshold scores are pairwise DISTINCT by construction (the order of
    equal scores under the reference's np.argsort(-scores) is unspecified, so ties are unpinned)."""
    fs = synth.feature_sizes(H, W)
    A = 1 if fcos else 9
    n_per_img = sum(fw * fh * A for fw, fh in fs)
    n_hot = max(1, n_per_img // 29)
    cls, reg, ctr = [], [], []
    start = 0
    for l, (fw, fh) in enumerate(fs):
        shape = (B, fh, fw, 80) if fcos else (B, fh, fw, A, 80)
        c = (synth.uniform(seed + 10 * l + 1, shape) * 0.04).astype(np.float32)  # all below 0.05
        n_l = fw * fh * A
        flat = c.reshape(B, n_l, 80)
        for b in range(B):
            idx = np.arange(start, start + n_l)
            hot = idx[(idx * 7 + b * 3) % 29 == 0]
            rank = (hot.astype(np.int64) * 7919 + b * 104729) % 1000003  # distinct for distinct anchors
            val = (0.06 + 0.93 * rank.astype(np.float64) / 1000003.0).astype(np.float32)
            klass = (hot * 13 + b * 5) % 80
            flat[b, hot - start, klass] = val
        start += n_l
        cls.append(c)
        if fcos:
            reg.append(synth.normal(seed + 10 * l + 2, (B, fh, fw, 4), std=0.5) + np.float32(3.0 + 0.7 * l))
            ctr.append((0.25 + 0.75 * synth.uniform(seed + 10 * l + 3, (B, fh, fw, 1))).astype(np.float32))
        else:
            reg.append(synth.normal(seed + 10 * l + 2, (B, fh, fw, A, 4), std=0.3))
    return (cls, reg, ctr) if fcos else (cls, reg)


def annots_cases(H, W):
    cases = {}
    for g in (1, 8, 100):
        cases[f'g{g}'] = synth.annotations(100 + g, 2, H, W, boxes_per_image=g)
    mixed = synth.annotations(555, 2, H, W, boxes_per_image=8)
    mixed[1, :, :] = -1
    cases['one_empty'] = mixed
    deg = synth.annotations(777, 2, H, W, boxes_per_image=6)
    deg[0, 0, 2:4] = deg[0, 0, 0:2]
    deg[0, 1, 0:4] = [-50., -40., 30., 35.]
    deg[1, 2, 0:4] = [W - 20., H - 20., W + 60., H + 70.]
    deg[1, 3, 0:4] = deg[1, 4, 0:4]
    cases['degenerate'] = deg
    return cases


def digest(a, nsamp=512):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    step = max(1, a.size // nsamp)
    return dict(sum=float(a.sum()), abssum=float(np.abs(a).sum()), n=int(a.size), step=int(step),
                samples=a[::step][:nsamp].astype(np.float32))


def labels_for(seed, n, num_classes):
    """Seeded class labels of the classification fixtures (same recipe as oracle/make_golden.py::labels_for)."""
    return (synth.uniform(seed, (n,)) * num_classes).astype(np.int64) % num_classes
