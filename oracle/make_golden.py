"""Generate tests/golden/* by running the REFERENCE itself on CPU (build container only).

    python oracle/make_golden.py [section ...]

The reference (/root/reference) is imported through oracle/import_reference.py;
inputs and weights come from the portable generators in oracle/synth.py and
oracle/model.py, so tests can regenerate the same inputs on the GPU box and
compare against the expected outputs stored here.  Only data (inputs' seeds and
expected outputs) is written -- no reference source travels.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import synth, model as omodel  # noqa: E402
from oracle.import_reference import load_reference, load_reference_classification  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
S1 = (800, 1333)
S2 = (1333, 1333)
TINY = (96, 128)
MID = (256, 320)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digest(t, nsamp=512):
    """Size-independent summary of a tensor: sum, abs-sum, strided samples."""
    a = t.detach().cpu().numpy().astype(np.float64).reshape(-1)
    step = max(1, a.size // nsamp)
    return dict(sum=float(a.sum()), abssum=float(np.abs(a).sum()), n=int(a.size),
                step=int(step), samples=a[::step][:nsamp].astype(np.float32))


def synth_preds_retina(seed, B, H, W, num_classes=80, A=9):
    fs = synth.feature_sizes(H, W)
    cls, reg = [], []
    for l, (fw, fh) in enumerate(fs):
        c = synth.uniform(seed + 10 * l + 1, (B, fh, fw, A, num_classes))
        c = (c**3).astype(np.float32)  # skewed towards 0 like real sigmoid outputs
        flat = c.reshape(-1)
        flat[0::9973] = 0.0          # clamp edges (losses.py:196)
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


def annots_cases(H, W):
    """name -> [B,100,5] annotation arrays incl. the edge cases of SURVEY.md 8c."""
    cases = {}
    for g in (1, 8, 100):
        cases[f'g{g}'] = synth.annotations(100 + g, 2, H, W, boxes_per_image=g)
    mixed = synth.annotations(555, 2, H, W, boxes_per_image=8)
    mixed[1, :, :] = -1  # image 1 has no GT (losses.py:341-345)
    cases['one_empty'] = mixed
    deg = synth.annotations(777, 2, H, W, boxes_per_image=6)
    deg[0, 0, 2:4] = deg[0, 0, 0:2]            # zero-area box
    deg[0, 1, 0:4] = [-50., -40., 30., 35.]    # partly outside the image
    deg[1, 2, 0:4] = [W - 20., H - 20., W + 60., H + 70.]
    deg[1, 3, 0:4] = deg[1, 4, 0:4]            # duplicated GT -> IoU ties (first index wins)
    cases['degenerate'] = deg
    return cases


# ----------------------------------------------------------------------------- sections
def sec_contract(ref):
    models, losses, decode = ref
    out = {}
    for name in ('resnet50_retinanet', 'resnet50_fcos'):
        net = models.__dict__[name](num_classes=80)
        sd = net.state_dict()
        out[name] = dict(
            keys=[[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()],
            params=[[k, list(v.shape)] for k, v in net.named_parameters()],
            n_params=int(sum(p.numel() for p in net.parameters())),
            one_dim_params=[k for k, v in net.named_parameters() if v.ndim == 1],
        )
        # the oracle state dict must load strictly into the reference model
        osd = omodel.init_retinanet_state(0, fcos=name.endswith('fcos'))
        missing = net.load_state_dict(osd, strict=True)
        out[name]['oracle_state_loads_strict'] = True
    json.dump(out, open(os.path.join(GOLD, 'contract.json'), 'w'))
    print('contract ok')


def sec_anchors(ref):
    models, losses, decode = ref
    from simpleAICV.detection.models.anchor import RetinaAnchors, FCOSPositions
    ra, fp = RetinaAnchors(), FCOSPositions()
    out = {}
    meta = {}
    for tag, (H, W) in (('S1', S1), ('S2', S2), ('TINY', TINY), ('MID', MID)):
        fs = synth.feature_sizes(H, W)
        lv = ra(fs)
        flat = np.concatenate([a.reshape(-1, 4) for a in lv], 0)
        pos = np.concatenate([p.reshape(-1, 2) for p in fp(fs)], 0)
        meta[tag] = dict(feature_sizes=fs, n_anchors=int(flat.shape[0]), anchors_sha256=sha(flat),
                         n_points=int(pos.shape[0]), points_sha256=sha(pos))
        if tag in ('TINY',):
            out['anchors_' + tag] = flat
            out['points_' + tag] = pos
        if tag == 'S1':
            for l, a in enumerate(lv):
                f = a.reshape(-1, 4)
                out[f'S1_l{l}_first64'] = f[:64]
                out[f'S1_l{l}_last64'] = f[-64:]
    out['base_anchors'] = np.stack([ra.generate_base_anchors(a, ra.scales, ra.ratios) for a in ra.areas], 0)
    # the feature sizes really are what the reference model produces (checked on TINY/MID in e2e)
    np.savez_compressed(os.path.join(GOLD, 'anchors.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'anchors.json'), 'w'))
    print('anchors ok', {k: v['n_anchors'] for k, v in meta.items()})


def sec_assign(ref):
    models, losses, decode = ref
    from simpleAICV.detection.models.anchor import RetinaAnchors, FCOSPositions
    out, meta = {}, {}
    for tag, (H, W) in (('TINY', TINY), ('MID', MID), ('S1', S1)):
        fs = synth.feature_sizes(H, W)
        anc = torch.tensor(np.concatenate([a.reshape(-1, 4) for a in RetinaAnchors()(fs)], 0))
        for cname, ann in annots_cases(H, W).items():
            ann_t = torch.tensor(ann)
            for blt in ('SmoothL1', 'GIoU'):
                crit = losses.__dict__['RetinaLoss'](box_loss_type=blt)
                tg = crit.get_batch_anchors_annotations(anc.unsqueeze(0).repeat(2, 1, 1), ann_t).numpy()
                key = f'retina_{tag}_{cname}_{blt}'
                lab = tg[..., 4].astype(np.int8)
                meta[key] = dict(label_sha256=sha(lab), npos=int((lab > 0).sum()),
                                 nign=int((lab < 0).sum()), nneg=int((lab == 0).sum()))
                if tag == 'TINY':
                    out[key] = tg
                else:
                    posm = lab > 0
                    meta[key]['pos_index_sha256'] = sha(np.nonzero(posm.reshape(-1))[0].astype(np.int64))
                    out[key + '_postargets'] = tg[posm][:, :4]
            # FCOS
            crit = losses.__dict__['FCOSLoss']()
            B = 2
            cls = [torch.zeros(B, fh, fw, 80) for fw, fh in fs]
            reg = [torch.zeros(B, fh, fw, 4) for fw, fh in fs]
            ctr = [torch.zeros(B, fh, fw, 1) for fw, fh in fs]
            pos = [torch.tensor(p).unsqueeze(0).repeat(B, 1, 1, 1) for p in FCOSPositions()(fs)]
            _, _, _, tg = crit.get_batch_position_annotations(cls, reg, ctr, pos, ann_t)
            tg = tg.numpy()
            key = f'fcos_{tag}_{cname}'
            lab = tg[..., 4].astype(np.int8)
            meta[key] = dict(label_sha256=sha(lab), npos=int((lab > 0).sum()))
            if tag == 'TINY':
                out[key] = tg
            else:
                posm = lab > 0
                out[key + '_postargets'] = tg[posm]
    np.savez_compressed(os.path.join(GOLD, 'assign.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'assign.json'), 'w'))
    print('assign ok', len(out), 'arrays')


def _loss_and_grads(crit, preds_np, ann):
    preds = [[torch.tensor(p, requires_grad=True) for p in lv] for lv in preds_np]
    ld = crit(preds, torch.tensor(ann))
    rec = {k: float(v) for k, v in ld.items()}
    total = sum(ld.values())
    grads = None
    if total.requires_grad:
        total.backward()
        grads = [[p.grad if p.grad is not None else torch.zeros_like(p) for p in lv] for lv in preds]
    return rec, grads


def sec_losses(ref):
    models, losses, decode = ref
    out, meta = {}, {}
    for tag, (H, W) in (('TINY', TINY), ('MID', MID), ('S1', S1)):
        cases = annots_cases(H, W)
        cases['none'] = synth.annotations(1, 2, H, W, boxes_per_image=0)  # zero-positive branch
        for cname, ann in cases.items():
            if tag == 'S1' and cname not in ('g8', 'one_empty'):
                continue
            for blt in ('SmoothL1', 'GIoU', 'IoU', 'DIoU', 'CIoU', 'EIoU'):
                if blt not in ('SmoothL1', 'GIoU') and (tag != 'TINY' or cname != 'g8'):
                    continue
                crit = losses.__dict__['RetinaLoss'](box_loss_type=blt)
                cls, reg = synth_preds_retina(11, 2, H, W)
                rec, grads = _loss_and_grads(crit, [cls, reg], ann)
                key = f'retina_{tag}_{cname}_{blt}'
                meta[key] = rec
                meta[key]['has_grad'] = grads is not None
                if grads is not None:
                    for nm, gl in zip(('dcls', 'dreg'), grads):
                        d = digest(torch.cat([g.reshape(-1) for g in gl]))
                        out[f'{key}_{nm}_samples'] = d.pop('samples')
                        meta[key][nm] = d
            crit = losses.__dict__['FCOSLoss']()
            cls, reg, ctr = synth_preds_fcos(13, 2, H, W)
            rec, grads = _loss_and_grads(crit, [cls, reg, ctr], ann)
            key = f'fcos_{tag}_{cname}'
            meta[key] = rec
            meta[key]['has_grad'] = grads is not None
            if grads is not None:
                for nm, gl in zip(('dcls', 'dreg', 'dctr'), grads):
                    d = digest(torch.cat([g.reshape(-1) for g in gl]))
                    out[f'{key}_{nm}_samples'] = d.pop('samples')
                    meta[key][nm] = d
    np.savez_compressed(os.path.join(GOLD, 'losses.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'losses.json'), 'w'), indent=0)
    print('losses ok', len(meta), 'cases')


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


def sec_decode(ref):
    models, losses, decode = ref
    out, meta = {}, {}
    for tag, (H, W) in (('TINY', TINY), ('MID', MID), ('S1', S1)):
        for nms_type in ('python_nms', 'diou_python_nms'):
            if tag == 'S1' and nms_type != 'python_nms':
                continue
            dec = decode.__dict__['RetinaDecoder'](nms_type=nms_type, nms_threshold=0.5)
            cls, reg = synth_preds_decode(21, 2, H, W)
            s, c, b = dec([[torch.tensor(x) for x in cls], [torch.tensor(x) for x in reg]])
            key = f'retina_{tag}_{nms_type}'
            for i in range(s.shape[0]):
                v = s[i][s[i] >= 0]
                assert np.unique(v).size == v.size, 'tied scores in a decode fixture'
            out[key + '_scores'], out[key + '_classes'], out[key + '_boxes'] = s, c, b
            meta[key] = dict(ndet=[int((s[i] >= 0).sum()) for i in range(s.shape[0])])
            dec = decode.__dict__['FCOSDecoder'](nms_type=nms_type, nms_threshold=0.6)
            cls, reg, ctr = synth_preds_decode(23, 2, H, W, fcos=True)
            s, c, b = dec([[torch.tensor(x) for x in cls], [torch.tensor(x) for x in reg],
                           [torch.tensor(x) for x in ctr]])
            key = f'fcos_{tag}_{nms_type}'
            for i in range(s.shape[0]):
                v = s[i][s[i] >= 0]
                assert np.unique(v).size == v.size, 'tied scores in a decode fixture'
            out[key + '_scores'], out[key + '_classes'], out[key + '_boxes'] = s, c, b
            meta[key] = dict(ndet=[int((s[i] >= 0).sum()) for i in range(s.shape[0])])
    np.savez_compressed(os.path.join(GOLD, 'decode.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'decode.json'), 'w'))
    print('decode ok', meta)


def _e2e(ref, name, H, W, fcos, box_loss_type='SmoothL1', ann_seed=3, num_classes=80, resnet='resnet50'):
    models, losses, decode = ref
    torch.manual_seed(0)
    net = models.__dict__[resnet + ('_fcos' if fcos else '_retinanet')](num_classes=num_classes)
    net.load_state_dict(omodel.init_retinanet_state(0, num_classes=num_classes, fcos=fcos,
                                                    backbone=resnet + 'backbone'), strict=True)
    net.train()
    crit = losses.__dict__['FCOSLoss']() if fcos else losses.__dict__['RetinaLoss'](box_loss_type=box_loss_type)
    x = torch.tensor(synth.images(5, 2, H, W))
    ann = torch.tensor(synth.annotations(ann_seed, 2, H, W, boxes_per_image=8))
    if num_classes != 80:
        ann[..., 4] = torch.where(ann[..., 4] >= 0, ann[..., 4] % num_classes, ann[..., 4])
    outs = net(x)
    ld = crit(outs, ann)
    total = sum(ld.values())
    total.backward()
    meta = dict(loss={k: float(v) for k, v in ld.items()},
                feature_sizes=[[o.shape[2], o.shape[1]] for o in outs[0]])
    arrs = {}
    for gi, grp in enumerate(outs):
        for l, o in enumerate(grp):
            d = digest(o, 256)
            arrs[f'{name}_out{gi}_l{l}_samples'] = d.pop('samples')
            meta[f'out{gi}_l{l}'] = d
    gn = {k: float(p.grad.double().norm()) for k, p in net.named_parameters()}
    meta['grad_norm'] = gn
    # a few full small gradients and the BN running stats after one training forward
    head = (('clsregcnt_head.cls_out.bias', 'clsregcnt_head.reg_out.bias', 'scales') if fcos else
            ('cls_head.cls_out.bias', 'reg_head.reg_out.bias'))
    block, layers = omodel.BACKBONES[resnet + 'backbone']
    last = 'conv3' if block == 'Bottleneck' else 'conv2'  # the last conv of a residual block
    for k in ('backbone.conv1.layer.1.weight', 'backbone.layer1.0.conv1.layer.1.bias', 'fpn.P5_1.bias',
              f'backbone.layer4.{layers[3] - 1}.{last}.layer.1.weight', 'backbone.layer1.0.conv1.layer.0.weight',
              'backbone.layer3.0.downsample_conv.layer.1.bias') + head:
        arrs[f'{name}_grad_{k}'] = dict(net.named_parameters())[k].grad.numpy()
    sd = net.state_dict()
    for k in ('backbone.conv1.layer.1.running_mean', 'backbone.conv1.layer.1.running_var',
              f'backbone.layer3.{layers[2] - 1}.{last}.layer.1.running_var'):
        arrs[f'{name}_buf_{k}'] = sd[k].numpy()
    return meta, arrs


def sec_e2e(ref):
    out, meta = {}, {}
    for name, (H, W), fcos, blt in (('retina_TINY', TINY, False, 'SmoothL1'),
                                    ('retina_MID', MID, False, 'SmoothL1'),
                                    ('retina_MID_giou', MID, False, 'GIoU'),
                                    ('fcos_TINY', TINY, True, None), ('fcos_MID', MID, True, None)):
        m, a = _e2e(ref, name, H, W, fcos, blt)
        assert m['feature_sizes'] == synth.feature_sizes(H, W), (m['feature_sizes'], synth.feature_sizes(H, W))
        meta[name] = m
        out.update(a)
        print(name, m['loss'])
    np.savez_compressed(os.path.join(GOLD, 'e2e.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'e2e.json'), 'w'), indent=0)


def sec_e2e_s1(ref):
    out, meta = {}, {}
    for name, fcos in (('retina_S1', False), ('fcos_S1', True)):
        m, a = _e2e(ref, name, S1[0], S1[1], fcos)
        assert m['feature_sizes'] == synth.feature_sizes(*S1)
        meta[name] = m
        out.update(a)
        print(name, m['loss'])
    np.savez_compressed(os.path.join(GOLD, 'e2e_s1.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'e2e_s1.json'), 'w'), indent=0)


def labels_for(seed, n, num_classes):
    return (synth.uniform(seed, (n,)) * num_classes).astype(np.int64) % num_classes


def sec_cls(ref):
    """ResNet50 ImageNet classification row: contract, loss / logits / gradient norms of the
    reference model on seeded inputs, and two reference SGD steps."""
    resnet, closses = load_reference_classification()
    meta, arrs = {}, {}
    net = resnet.resnet50(num_classes=1000)
    sd = net.state_dict()
    meta['contract'] = dict(keys=[[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()],
                            params=[[k, list(v.shape)] for k, v in net.named_parameters()],
                            n_params=int(sum(p.numel() for p in net.parameters())))
    for name, B, R, nc in (('cls_64', 4, 64, 1000), ('cls_224', 8, 224, 1000), ('cls_96_nc40', 6, 96, 40)):
        net = resnet.resnet50(num_classes=nc)
        net.load_state_dict(omodel.init_resnet50_cls_state(0, nc), strict=True)
        net.train()
        crit = closses.CELoss()
        x = torch.tensor(synth.images(11, B, R, R))
        lab = torch.tensor(labels_for(23, B, nc))
        logits = net(x)
        loss = crit(logits, lab)
        loss.backward()
        m = dict(B=B, R=R, num_classes=nc, loss=float(loss.detach()), labels=lab.tolist(),
                 grad_norm={k: float(p.grad.double().norm()) for k, p in net.named_parameters()})
        arrs[f'{name}_logits'] = logits.detach().numpy()
        for k in ('fc.bias', 'conv1.layer.1.weight', 'layer4.2.conv3.layer.1.bias'):
            arrs[f'{name}_grad_{k}'] = dict(net.named_parameters())[k].grad.numpy().copy()
        arrs[f'{name}_buf_conv1.layer.1.running_var'] = net.state_dict()['conv1.layer.1.running_var'].numpy().copy()
        # two optimizer steps of the recipe's SGD (second one exercises the momentum buffer)
        one_d = [p for p in net.parameters() if p.ndim == 1]
        multi = [p for p in net.parameters() if p.ndim > 1]
        opt = torch.optim.SGD([{'params': one_d, 'weight_decay': 0.0}, {'params': multi, 'weight_decay': 1e-4}],
                              lr=0.1, momentum=0.9)
        opt.step()
        opt.zero_grad()
        loss2 = crit(net(x), lab)
        loss2.backward()
        opt.step()
        m['loss_step1'] = float(loss2)
        # yardstick for the step-1 loss: the same step by the oracle in fp64 (lr 0.1 on random-init weights
        # amplifies the fp32 rounding noise of the first backward pass into percents of the next loss)
        sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone())
                for k, v in omodel.init_resnet50_cls_state(0, nc).items()}
        P = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.dtype.is_floating_point
             and 'running_' not in k}
        sd64.update(P)
        omodel.ce_loss(omodel.resnet50_cls_forward(sd64, x.double(), True), lab).backward()
        Pn = {k: v.detach().clone() for k, v in P.items()}
        omodel.sgd_momentum_step(Pn, {k: v.grad for k, v in P.items()}, {}, 0.1, 0.9, 1e-4, first=True)
        sd64.update(Pn)
        with torch.no_grad():
            m['loss_step1_fp64'] = float(omodel.ce_loss(omodel.resnet50_cls_forward(sd64, x.double(), True), lab))
        m['param_norm_after2'] = {k: float(p.detach().double().norm()) for k, p in net.named_parameters()}
        net.eval()
        with torch.no_grad():
            arrs[f'{name}_eval_logits'] = net(x).numpy()
        meta[name] = m
        print(name, m['loss'], m['loss_step1'])
    np.savez_compressed(os.path.join(GOLD, 'cls.npz'), **arrs)
    json.dump(meta, open(os.path.join(GOLD, 'cls.json'), 'w'), indent=0)


def sec_e2e_s2(ref):
    """Full 1333x1333 batch (the collater's square padding): losses, output samples, gradient norms
    and a few full gradients of the reference model, RetinaNet and FCOS."""
    out, meta = {}, {}
    for name, fcos in (('retina_S2', False), ('fcos_S2', True)):
        m, a = _e2e(ref, name, S2[0], S2[1], fcos)
        assert m['feature_sizes'] == synth.feature_sizes(*S2)
        meta[name] = m
        out.update(a)
        print(name, m['loss'])
    np.savez_compressed(os.path.join(GOLD, 'e2e_s2.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'e2e_s2.json'), 'w'), indent=0)


def sec_e2e_other(ref):
    """The square yolo-style inputs of the other detection recipes: VOC 640 x 640 with 20 classes
    (3.detection_training/voc/res50_{retinanet,fcos}_yoloresize640/train_config.py:23-24) and the Objects365 canvas
    1024 x 1024 (objects365/res50_retinanet_yoloresize1024/train_config.py:23-24), the canvas with COCO's 80 classes and
    the 365 classes at 256 x 320."""
    out, meta = {}, {}
    for name, (H, W), fcos, nc in (('retina_V640', (640, 640), False, 20), ('fcos_V640', (640, 640), True, 20),
                                   ('retina_O1024', (1024, 1024), False, 80),
                                   # Objects365's class count (9 x 365 = 3285 class channels: not a multiple of 4) at a
                                   # small canvas -- the 1024 x 1024 recipe itself has 1.1 G class scores per batch
                                   ('retina_C365', (256, 320), False, 365), ('fcos_C365', (256, 320), True, 365)):
        m, a = _e2e(ref, name, H, W, fcos, num_classes=nc)
        assert m['feature_sizes'] == synth.feature_sizes(H, W)
        m['H'], m['W'], m['num_classes'] = H, W, nc
        meta[name] = m
        out.update(a)
        print(name, m['loss'])
    np.savez_compressed(os.path.join(GOLD, 'e2e_other.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'e2e_other.json'), 'w'), indent=0)


def sec_e2e_backbones(ref):
    """The other backbones of the reference's RetinaNet / FCOS factories (retinanet.py:102-130, fcos.py:100-128;
    backbones/resnet.py:127-169): BasicBlock ResNet18 / 34 (C3..C5 = 128 / 256 / 512 channels) and the deeper Bottleneck
    ResNet101 / 152, one training forward + backward at 256 x 320."""
    out, meta = {}, {}
    for name, fcos, resnet in (('retina_R18', False, 'resnet18'), ('fcos_R34', True, 'resnet34'),
                               ('retina_R101', False, 'resnet101'), ('fcos_R152', True, 'resnet152')):
        m, a = _e2e(ref, name, 256, 320, fcos, resnet=resnet)
        assert m['feature_sizes'] == synth.feature_sizes(256, 320)
        m['H'], m['W'], m['num_classes'], m['resnet'] = 256, 320, 80, resnet
        meta[name] = m
        out.update(a)
        print(name, m['loss'])
    np.savez_compressed(os.path.join(GOLD, 'e2e_backbones.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'e2e_backbones.json'), 'w'), indent=0)


class _Cfg:
    pass


def _sched_cfg(name, params, epochs, lr):
    c = _Cfg()
    c.scheduler = (name, params)
    c.epochs = epochs
    c.optimizer = ('AdamW', {'lr': lr})
    return c


class _FakeOpt:
    def __init__(self, lrs):
        self.param_groups = [{'lr': v} for v in lrs]


SCHED_CASES = {
    # 3.detection_training/coco/res50_retinanet_retinaresize800/train_config.py:137-146 (13 epochs)
    'retina_multistep': ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [8, 12]}, 13, 1e-4, [1e-4, 1e-4]),
    # 0.classification_training/imagenet/resnet50/train_config.py (100 epochs, milestones 30/60/90)
    'r50_multistep': ('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [30, 60, 90]}, 100, 0.1, [0.1, 0.1]),
    'warm_multistep': ('MultiStepLR', {'warm_up_epochs': 2, 'gamma': 0.5, 'milestones': [3, 5]}, 6, 2e-4, [2e-4, 5e-5]),
    'cosine_warm': ('CosineLR', {'warm_up_epochs': 1, 'min_lr': 1e-6}, 8, 1e-3, [1e-3, 2.5e-4]),
    'cosine_nomin': ('CosineLR', {'warm_up_epochs': 0}, 5, 0.1, [0.1]),
    'poly': ('PolyLR', {'warm_up_epochs': 1, 'power': 0.9, 'min_lr': 1e-5}, 7, 0.01, [0.01, 0.02]),
}


def load_reference_tools_utils():
    load_reference()
    import importlib
    if ROOT in sys.path:  # the repo has no `tools` package, but keep the reference's first
        pass
    return importlib.import_module('tools.utils')


def sec_scheduler(ref):
    """LR sequences of the reference Scheduler (tools/utils.py:188-272) driven the way train_detection drives it:
    every iteration with epoch = iter_index / iters + (epoch - 1) (tools/scripts.py:1031-1033)."""
    tu = load_reference_tools_utils()
    out = {}
    iters = 7
    for key, (name, params, epochs, lr, group_lrs) in SCHED_CASES.items():
        cfg = _sched_cfg(name, params, epochs, lr)
        opt = _FakeOpt(group_lrs)
        sch = tu.Scheduler(cfg, opt)
        seq = []
        for epoch in range(1, epochs + 1):
            for it in range(1, iters + 1):
                e = it / iters + (epoch - 1)
                sch.step(opt, e)
                seq.append([e, sch.current_lr] + [g['lr'] for g in opt.param_groups])
        out[key] = dict(name=name, params=params, epochs=epochs, lr=lr, group_lrs=group_lrs, iters=iters, seq=seq)
    json.dump(out, open(os.path.join(GOLD, 'scheduler.json'), 'w'))
    print('scheduler ok', {k: len(v['seq']) for k, v in out.items()})


def _module_of(name):
    parts = name.split('.')
    return '.'.join(parts[:2]) if parts[0] == 'backbone' else parts[0]


def _run_reference_loop(ref, tu, H, W, n_iters, accumulation_steps=1, clip_grad_value=0.0, clip_max_norm=0.0,
                        sched=('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [1, 2]}), epochs=3,
                        iters_per_epoch=2, bad_iter=None, dtype=torch.float32):
    """The body of tools/scripts.py::train_detection (:866-1036) for world size 1 on CPU, with the reference's
    model, loss, build_optimizer and Scheduler objects (the function itself needs CUDA and a process group).
    dtype=torch.float64: the same loop with the model, its inputs and the optimizer state in double precision -- the
    yardstick for quantities that sit at the noise floor of fp32 arithmetic (parameter norms after a few AdamW steps:
    elements whose gradient is rounding noise move by +-lr in ANY fp32 implementation, the reference's included)."""
    models, losses, decode = ref
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    net.load_state_dict(omodel.init_retinanet_state(0), strict=True)
    net = net.to(dtype)
    net.train()
    crit = losses.__dict__['RetinaLoss']()
    cfg = _Cfg()
    cfg.optimizer = ('AdamW', {'lr': 1e-4, 'global_weight_decay': False, 'weight_decay': 1e-3,
                               'no_weight_decay_layer_name_list': []})
    cfg.scheduler = sched
    cfg.epochs = epochs
    optimizer, layer_list = tu.build_optimizer(cfg, net)
    scheduler = tu.Scheduler(cfg, optimizer)
    rec = dict(groups=[dict(n=len(g['params']), weight_decay=g['weight_decay'], lr=g['lr'])
                       for g in optimizer.param_groups], iters=[])
    iter_index, epoch, seen = 1, 1, 0
    data_i = 0
    while seen < n_iters:
        images = torch.tensor(synth.images(40 + data_i, 2, H, W)).to(dtype)
        annots = torch.tensor(synth.annotations(60 + data_i, 2, H, W, boxes_per_image=6))
        if bad_iter is not None and data_i == bad_iter:
            annots[0, 0, 0] = float('inf')  # an inf in the annotations: the input guard must skip the batch
        data_i += 1
        seen += 1
        skip = bool(torch.any(torch.isinf(images)) or torch.any(torch.isinf(annots)) or
                    torch.any(torch.isnan(images)) or torch.any(torch.isnan(annots)))
        outs = net(images)
        loss_value = crit(outs, annots)
        loss = sum(loss_value.values())
        bad = any(bool(torch.any(torch.isinf(v)) or torch.any(torch.isnan(v))) for v in loss_value.values())
        if loss == 0. or bad:
            skip = True
        loss = loss / accumulation_steps
        it = dict(loss={k: float(v) / accumulation_steps for k, v in loss_value.items()}, skipped=False,
                  iter_index=iter_index)
        loss.backward()
        if skip:
            optimizer.zero_grad()
            it['skipped'] = True
            rec['iters'].append(it)
            continue
        if iter_index % accumulation_steps == 0:
            if clip_grad_value > 0 or clip_max_norm > 0:
                if clip_grad_value > 0:
                    torch.nn.utils.clip_grad_value_(net.parameters(), clip_grad_value)
                if clip_max_norm > 0:
                    it['grad_norm'] = float(torch.nn.utils.clip_grad_norm_(net.parameters(), clip_max_norm))
            optimizer.step()
            optimizer.zero_grad()
            scheduler.step(optimizer, iter_index / iters_per_epoch + (epoch - 1))
            it['lr_after'] = [g['lr'] for g in optimizer.param_groups]
        if iter_index == iters_per_epoch:
            iter_index, epoch = 0, epoch + 1
        iter_index += 1
        rec['iters'].append(it)
    mods = {}
    for k, p in net.named_parameters():
        mods.setdefault(_module_of(k), 0.0)
        mods[_module_of(k)] += float(p.detach().double().pow(2).sum())
    rec['module_param_norm'] = {k: float(np.sqrt(v)) for k, v in mods.items()}
    init, dmods = omodel.init_retinanet_state(0), {}
    for k, p in net.named_parameters():  # how far the optimizer moved each module (sensitive to lr / clip / skips)
        dmods.setdefault(_module_of(k), 0.0)
        dmods[_module_of(k)] += float((p.detach().double() - init[k].double()).pow(2).sum())
    rec['module_delta_norm'] = {k: float(np.sqrt(v)) for k, v in dmods.items()}
    rec['param_norm'] = {k: float(p.detach().double().norm()) for k, p in net.named_parameters()}
    sd = optimizer.state_dict()
    rec['opt_steps'] = sorted({float(v['step']) for v in sd['state'].values()})
    # optimizer-state digests: parameter index -> (name, norms of the moments)
    names = [n for grp in layer_list for n in grp['name']]
    rec['opt_state_norm'] = {names[i]: [float(v['exp_avg'].double().norm()), float(v['exp_avg_sq'].double().norm())]
                             for i, v in sd['state'].items()}
    return rec, net


def sec_trainer(ref):
    """a-13: the reference's training loop for 3 iterations (model + RetinaLoss + build_optimizer AdamW + Scheduler),
    plain / with gradient accumulation + clipping / with a batch the input guard must skip."""
    tu = load_reference_tools_utils()
    out = {}
    H, W = TINY
    rec, _ = _run_reference_loop(ref, tu, H, W, 3)
    out['plain3'] = rec
    rec, _ = _run_reference_loop(ref, tu, H, W, 4, accumulation_steps=2, clip_grad_value=0.05, clip_max_norm=0.5,
                                 iters_per_epoch=4, epochs=3,
                                 sched=('CosineLR', {'warm_up_epochs': 1, 'min_lr': 1e-6}))
    out['accum2_clip'] = rec
    rec, _ = _run_reference_loop(ref, tu, H, W, 3, bad_iter=1)
    out['skip_inf_annot'] = rec
    H, W = MID
    rec, _ = _run_reference_loop(ref, tu, H, W, 3)
    out['plain3_MID'] = rec
    # fp64 yardsticks of the two plain runs: the same loop in double precision.  module_param_norm_fp64 is what the
    # parameter norms converge to; |fp32 reference - fp64| per module is the reference arithmetic's own noise.
    for case, (H, W) in (('plain3', TINY), ('plain3_MID', MID)):
        rec64, _ = _run_reference_loop(ref, tu, H, W, 3, dtype=torch.float64)
        out[case]['module_param_norm_fp64'] = rec64['module_param_norm']
        out[case]['module_delta_norm_fp64'] = rec64['module_delta_norm']
        out[case]['loss_fp64'] = [it['loss'] for it in rec64['iters']]
        worst = max(abs(out[case]['module_param_norm'][k] - v) / v for k, v in rec64['module_param_norm'].items())
        print(case, 'fp32 reference vs fp64: worst module parameter-norm distance', worst)
    # the clipped total gradient norms of the accumulation run in double precision (96 x 128: BatchNorm statistics over
    # 24 pixels in layer4 -- the gradient is ill-conditioned, the fp32 reference itself is percents away from fp64)
    rec64, _ = _run_reference_loop(ref, tu, TINY[0], TINY[1], 4, accumulation_steps=2, clip_grad_value=0.05,
                                   clip_max_norm=0.5, iters_per_epoch=4, epochs=3,
                                   sched=('CosineLR', {'warm_up_epochs': 1, 'min_lr': 1e-6}), dtype=torch.float64)
    out['accum2_clip']['grad_norm_fp64'] = [it['grad_norm'] for it in rec64['iters'] if 'grad_norm' in it]
    print('accum2_clip grad norms fp32', [it['grad_norm'] for it in out['accum2_clip']['iters'] if 'grad_norm' in it],
          'fp64', out['accum2_clip']['grad_norm_fp64'])
    json.dump(out, open(os.path.join(GOLD, 'trainer.json'), 'w'))
    for k, v in out.items():
        print(k, [(i['loss'], i['skipped']) for i in v['iters']], v['opt_steps'])


def sec_collate(ref):
    """Normalize + DetectionCollater of the reference (simpleAICV/detection/common.py:223-286) on seeded uint8
    images of ragged sizes: full output for a small yolo-style size, sha256 + strided samples at 1333."""
    load_reference()
    import importlib
    common = importlib.import_module('simpleAICV.detection.common')
    out, meta = {}, {}
    for tag, resize, rtype, shapes in (('yolo64', 64, 'yolo_style', [(40, 64), (64, 33), (17, 21)]),
                                       ('retina800', 800, 'retina_style', [(800, 1201), (750, 1333)]),
                                       ('retina96', 96, 'retina_style', [(96, 160), (60, 100), (1, 1)])):
        samples = []
        for i, (h, w) in enumerate(shapes):
            img = synth.uint8_image(900 + 7 * i, h, w)
            na = [3, 0, 100][i % 3]
            ann = synth.annotations(901 + i, 1, h, w, boxes_per_image=na, max_objects=max(na, 1))[0][:na]
            s = dict(image=img.astype(np.float32), annots=ann.astype(np.float32), scale=np.float32(1.0 + 0.25 * i),
                     size=np.array([h, w], dtype=np.float32))
            s = common.Normalize()(s)
            samples.append(s)
        coll = common.DetectionCollater(resize=resize, resize_type=rtype, max_annots_num=100)
        res = coll(samples)
        img = res['image']
        assert img.shape[1] == 3 and not img.is_contiguous()  # logical NCHW, channels_last storage
        a = img.permute(0, 2, 3, 1).contiguous().numpy()      # [B,R,R,3]
        meta[tag] = dict(resize=resize, resize_type=rtype, shapes=shapes, R=int(a.shape[1]), sha=sha(a),
                         annots_sha=sha(res['annots'].numpy()), sum=float(a.astype(np.float64).sum()),
                         strides=list(img.stride()))
        if a.size <= 64 * 64 * 3 * 3:
            out[f'{tag}_image'] = a
        else:
            out[f'{tag}_samples'] = a.reshape(-1)[::9973].copy()
        out[f'{tag}_annots'] = res['annots'].numpy()
        out[f'{tag}_scale'] = res['scale']
        out[f'{tag}_size'] = res['size']
    np.savez_compressed(os.path.join(GOLD, 'collate.npz'), **out)
    json.dump(meta, open(os.path.join(GOLD, 'collate.json'), 'w'))
    print('collate ok', {k: v['sha'][:12] for k, v in meta.items()})


def sec_evalpost(ref):
    """evaluate_coco_detection's post-steps (tools/scripts.py:742-758) applied to the reference decoder's own output.
    tools/scripts.py cannot be imported (pycocotools), so the few array statements are replayed here on the arrays:
    boxes /= scales[:,None,None]; per image: clip x1,y1 at 0 and x2,y2 at (w,h); xyxy -> xywh."""
    models, losses, decode = ref
    out = {}
    for tag, (H, W) in (('TINY', TINY), ('MID', MID)):
        dec = decode.__dict__['RetinaDecoder'](nms_type='python_nms', nms_threshold=0.5)
        cls, reg = synth_preds_decode(21, 2, H, W)
        scores, classes, boxes = dec([[torch.tensor(x) for x in cls], [torch.tensor(x) for x in reg]])
        scales = np.array([1.25, 0.6], dtype=np.float32)
        sizes = np.array([[H * 0.7, W * 0.8], [H / 0.6 - 3, W / 0.6 - 5]], dtype=np.float32)
        out[f'{tag}_boxes_in'] = boxes.copy()
        boxes /= np.expand_dims(np.expand_dims(scales, axis=-1), axis=-1)
        for per_image_boxes, per_image_size in zip(boxes, sizes):
            per_image_boxes[:, 0] = np.maximum(per_image_boxes[:, 0], 0)
            per_image_boxes[:, 1] = np.maximum(per_image_boxes[:, 1], 0)
            per_image_boxes[:, 2] = np.minimum(per_image_boxes[:, 2], per_image_size[1])
            per_image_boxes[:, 3] = np.minimum(per_image_boxes[:, 3], per_image_size[0])
            per_image_boxes[:, 2:] -= per_image_boxes[:, :2]
        out[f'{tag}_boxes_out'] = boxes
        out[f'{tag}_scales'] = scales
        out[f'{tag}_sizes'] = sizes
    np.savez_compressed(os.path.join(GOLD, 'evalpost.npz'), **out)
    print('evalpost ok')



def sec_cifar(ref):
    """BASELINE.json config #1: resnet18cifar on CIFAR-100 shaped batches (32x32, bs 32), the recipe's step
    (0.classification_training/cifar100/resnet18cifar/train_config.py: CELoss, SGD lr 0.1 momentum 0.9 wd 5e-4
    on ndim>1, total batch 128): contract, logits / loss / gradient norms, two reference SGD steps; plus resnet34cifar's contract."""
    load_reference()
    import importlib
    rc = importlib.import_module('simpleAICV.classification.backbones.resnetforcifar')
    closses = importlib.import_module('simpleAICV.classification.losses')
    meta, arrs = {}, {}
    for arch, layers in (('resnet18cifar', (2, 2, 2, 2)), ('resnet34cifar', (3, 4, 6, 3))):
        net = rc.__dict__[arch](num_classes=100)
        meta[arch + '_contract'] = dict(
            keys=[[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in net.state_dict().items()],
            params=[[k, list(v.shape)] for k, v in net.named_parameters()],
            n_params=int(sum(p.numel() for p in net.parameters())))
    for name, B, R, nc in (('r18c_32', 32, 32, 100), ('r18c_8', 4, 8, 10)):
        net = rc.resnet18cifar(num_classes=nc)
        net.load_state_dict(omodel.init_resnet_cls_state(0, nc), strict=True)
        net.train()
        crit = closses.CELoss()
        x = torch.tensor(synth.images(31, B, R, R))
        lab = torch.tensor(labels_for(37, B, nc))
        logits = net(x)
        loss = crit(logits, lab)
        loss.backward()
        m = dict(B=B, R=R, num_classes=nc, loss=float(loss.detach()), labels=lab.tolist(),
                 grad_norm={k: float(p.grad.double().norm()) for k, p in net.named_parameters()})
        arrs[f'{name}_logits'] = logits.detach().numpy()
        for k in ('fc.bias', 'conv1.layer.1.weight', 'conv1.layer.0.weight', 'layer4.1.conv2.layer.1.bias',
                  'layer2.0.downsample_conv.layer.0.weight'):
            arrs[f'{name}_grad_{k}'] = dict(net.named_parameters())[k].grad.numpy().copy()
        arrs[f'{name}_buf_conv1.layer.1.running_var'] = net.state_dict()['conv1.layer.1.running_var'].numpy().copy()
        one_d = [p for p in net.parameters() if p.ndim == 1]
        multi = [p for p in net.parameters() if p.ndim > 1]
        opt = torch.optim.SGD([{'params': one_d, 'weight_decay': 0.0}, {'params': multi, 'weight_decay': 5e-4}],
                              lr=0.01, momentum=0.9)  # a linear-regime rate: lr 0.1 on random init is chaotic (cls)
        opt.step()
        opt.zero_grad()
        loss2 = crit(net(x), lab)
        loss2.backward()
        opt.step()
        m['loss_step1'] = float(loss2)
        m['param_norm_after2'] = {k: float(p.detach().double().norm()) for k, p in net.named_parameters()}
        net.eval()
        with torch.no_grad():
            arrs[f'{name}_eval_logits'] = net(x).numpy()
        meta[name] = m
        print(name, m['loss'], m['loss_step1'])
    np.savez_compressed(os.path.join(GOLD, 'cifar.npz'), **arrs)
    json.dump(meta, open(os.path.join(GOLD, 'cifar.json'), 'w'), indent=0)



def sec_cls_bf16(ref):
    """SURVEY.md 8f-1 as written (autocast): the reference ResNet50 + CELoss under torch.autocast('cpu', bfloat16) on the
    inputs of section cls (fp32 results of the same inputs: cls.json) -- logits, loss, gradient norms, a few full
    gradients.  The reference recipe autocasts to fp16 on CUDA with a GradScaler; BASELINE.json asks for bf16, whose
    CPU autocast is what can run here (parity of the fp16-CUDA path itself is unpinned)."""
    resnet, closses = load_reference_classification()
    meta, arrs = {}, {}
    for name, B, R, nc in (('cls_64', 4, 64, 1000), ('cls_224', 8, 224, 1000), ('cls_64_b32', 32, 64, 1000)):
        net = resnet.resnet50(num_classes=nc)
        net.load_state_dict(omodel.init_resnet50_cls_state(0, nc), strict=True)
        net.train()
        crit = closses.CELoss()
        x = torch.tensor(synth.images(11, B, R, R))
        lab = torch.tensor(labels_for(23, B, nc))
        with torch.autocast('cpu', dtype=torch.bfloat16):
            logits = net(x)
            loss = crit(logits, lab)
        loss.backward()
        m = dict(B=B, R=R, num_classes=nc, loss=float(loss.detach()), logits_dtype=str(logits.dtype),
                 grad_norm={k: float(p.grad.double().norm()) for k, p in net.named_parameters()})
        arrs[f'{name}_logits'] = logits.detach().float().numpy()
        for k in ('fc.bias', 'conv1.layer.1.weight', 'layer4.2.conv3.layer.1.bias', 'layer1.0.conv1.layer.0.weight'):
            arrs[f'{name}_grad_{k}'] = dict(net.named_parameters())[k].grad.numpy().copy()
        # fp32 run of the same inputs as the yardstick for what bf16 costs
        net32 = resnet.resnet50(num_classes=nc)
        net32.load_state_dict(omodel.init_resnet50_cls_state(0, nc), strict=True)
        net32.train()
        l32 = net32(x)
        loss32 = crit(l32, lab)
        loss32.backward()
        m['loss_fp32'] = float(loss32.detach())
        m['grad_norm_fp32'] = {k: float(p.grad.double().norm()) for k, p in net32.named_parameters()}
        arrs[f'{name}_logits_fp32'] = l32.detach().numpy()
        meta[name] = m
        print(name, m['loss'], m['loss_fp32'], logits.dtype)
    np.savez_compressed(os.path.join(GOLD, 'cls_bf16.npz'), **arrs)
    json.dump(meta, open(os.path.join(GOLD, 'cls_bf16.json'), 'w'), indent=0)



MAE_CFGS = {
    # name: (cfg of oracle.mae, batch)
    'tiny': (dict(patch=16, image=64, E=128, Le=2, He=4, D=64, Ld=1, Hd=4), 4),
    'base_d2': (dict(patch=16, image=224, E=768, Le=2, He=12, D=512, Ld=1, Hd=16), 4),
    'base': (dict(patch=16, image=224, E=768, Le=12, He=12, D=512, Ld=8, Hd=16), 2),
    # the widths of vit_large_patch16_224_mae_pretrain_model (vit_mae.py: 1024 x 24 blocks x 16 heads), two blocks deep
    'large_d2': (dict(patch=16, image=224, E=1024, Le=2, He=16, D=512, Ld=1, Hd=16), 4),
}


def sec_mae(ref):
    """the reference VITMAEPretrainModel + MSELoss + collater labels on seeded inputs, fp32 and under
    torch.autocast('cpu', bfloat16) (the recipe's use_amp).  The masking noise is the reference's own torch.rand draw
    after torch.manual_seed(s) (recorded: the first RNG use of forward is random_masking, dropout is 0)."""
    load_reference()
    import importlib
    mm = importlib.import_module('simpleAICV.masked_image_modeling.models.vit_mae')
    ml = importlib.import_module('simpleAICV.masked_image_modeling.losses')
    mc = importlib.import_module('simpleAICV.masked_image_modeling.common')
    from oracle import mae as omae
    meta, arrs = {}, {}
    for name, (cfg, B) in MAE_CFGS.items():
        def build():
            net = mm.VITMAEPretrainModel(patch_size=cfg['patch'], image_size=cfg['image'], mask_ratio=0.75,
                                         encoder_embedding_planes=cfg['E'], encoder_block_nums=cfg['Le'],
                                         encoder_head_nums=cfg['He'], decoder_embedding_planes=cfg['D'],
                                         decoder_block_nums=cfg['Ld'], decoder_head_nums=cfg['Hd'])
            net.load_state_dict(omae.init_mae_state(0, cfg), strict=True)
            return net.train()
        net = build()
        if name == 'tiny':
            meta['contract_tiny'] = dict(
                keys=[[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in net.state_dict().items()],
                params=[[k, list(v.shape), bool(v.requires_grad)] for k, v in net.named_parameters()])
        if name == 'base':
            full = mm.vit_base_patch16_224_mae_pretrain_model()
            meta['contract_base'] = dict(
                keys=[[k, list(v.shape)] for k, v in full.state_dict().items()],
                n_params=int(sum(p.numel() for p in full.parameters())),
                pos_embed_sha=sha(full.encoder.pos_embed.detach().numpy()),
                dec_pos_embed_sha=sha(full.decoder.pos_embed.detach().numpy()))
        S = cfg['image']
        hwc = np.transpose(synth.images(71, B, S, S), (0, 2, 3, 1)) * 4.0 - 2.0   # ~ mean/std-normalised pixels
        batch = mc.MAESelfSupervisedPretrainCollater(image_size=S, patch_size=cfg['patch'], norm_label=True)(
            [dict(image=hwc[i]) for i in range(B)])
        images, labels = batch['image'], batch['label']
        crit = ml.MSELoss()
        n = (S // cfg['patch'])**2
        torch.manual_seed(1234 + len(name))
        noise = torch.rand(B, n)
        m = dict(cfg=cfg, B=B, noise_seed=1234 + len(name))
        arrs[f'{name}_noise'] = noise.numpy()
        arrs[f'{name}_labels_samples'] = labels.numpy().reshape(-1)[::997].copy()
        m['labels_sha'] = sha(labels.numpy())
        for tag, ctx in (('fp32', None), ('bf16', torch.autocast('cpu', dtype=torch.bfloat16))):
            net = build()
            torch.manual_seed(1234 + len(name))
            if ctx is None:
                out, mask = net(images)
                loss = crit(out, labels, mask)
            else:
                with ctx:
                    out, mask = net(images)
                    loss = crit(out, labels, mask)
            loss.backward()
            m[f'loss_{tag}'] = float(loss.detach())
            m[f'grad_norm_{tag}'] = {k: float(p.grad.double().norm()) for k, p in net.named_parameters()
                                     if p.grad is not None}
            arrs[f'{name}_pred_{tag}'] = out.detach().float().numpy().reshape(-1)[::211].copy()
            if tag == 'fp32':
                arrs[f'{name}_mask'] = mask.numpy()
                # restore ids and mask from the reference encoder itself (same seed => same noise draw)
                probe = build()  # (construction consumes RNG: seed AFTER it)
                torch.manual_seed(1234 + len(name))
                with torch.no_grad():
                    _, mask_e, restore_e = probe.encoder(images)
                assert torch.equal(mask_e, mask)
                arrs[f'{name}_restore'] = restore_e.numpy()
                for k in ('encoder.cls_token', 'decoder.mask_token', 'encoder.norm.weight', 'decoder.fc.bias',
                          'encoder.blocks.0.attn.qkv.bias', 'encoder.patch_embed.proj.bias'):
                    arrs[f'{name}_grad_{k}'] = dict(net.named_parameters())[k].grad.numpy().copy()
        meta[name] = m
        print(name, m['loss_fp32'], m['loss_bf16'])
    np.savez_compressed(os.path.join(GOLD, 'mae.npz'), **arrs)
    json.dump(meta, open(os.path.join(GOLD, 'mae.json'), 'w'), indent=0)



SECTIONS = dict(cls=sec_cls, contract=sec_contract, anchors=sec_anchors, assign=sec_assign, losses=sec_losses,
                decode=sec_decode, e2e=sec_e2e, e2e_s1=sec_e2e_s1, e2e_s2=sec_e2e_s2, scheduler=sec_scheduler,
                trainer=sec_trainer, collate=sec_collate, e2e_other=sec_e2e_other, e2e_backbones=sec_e2e_backbones, evalpost=sec_evalpost, cifar=sec_cifar, cls_bf16=sec_cls_bf16, mae=sec_mae)

if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    ref = load_reference()
    for s in (sys.argv[1:] or list(SECTIONS)):
        SECTIONS[s](ref)
