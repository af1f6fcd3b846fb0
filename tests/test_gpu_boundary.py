"""SURVEY.md 8f-2 / 8f-3 on the GPU: the padding-skip collater canvas, the pinned-memory prefetcher, the evaluation
pipeline (forward -> decode/NMS -> box post-steps -> result records) and its sharding over two ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import decode as odecode, model as omodel, synth


def _samples(seed, shapes):
    rng = np.random.RandomState(seed)
    out = []
    for h, w, n in shapes:
        out.append({'image': rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8),
                    'annots': (rng.rand(n, 5) * 30).astype(np.float32), 'scale': np.float32(0.5 + rng.rand()),
                    'size': (h * 2, w * 2)})
    return out


def test_padding_skip_canvas_is_the_crop_of_the_reference_batch():
    """crop_padding=True: the batch equals the reference collater's R x R batch (oracle/decode.py::collate, pinned by
    tests/golden/collate.*) cropped to the canvas -- bit for bit -- and everything cropped away was zero padding."""
    from sacv_amd.detection import common
    samples = _samples(0, ((37, 51, 3), (64, 40, 0), (5, 66, 7)))
    ref_img, ref_ann = odecode.collate(samples, resize=40, max_annots_num=10)  # R = 67
    col = common.DetectionCollater(resize=40, max_annots_num=10, crop_padding=True, pad_multiple=8)
    out = col(samples)
    img = out['image']
    assert tuple(img.shape) == (3, 3, 64, 67) and img.is_contiguous(memory_format=torch.channels_last)
    got = img.permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(got, ref_img[:, :64, :67]) and not ref_img[:, 64:, :].any()
    assert np.array_equal(out['annots'].numpy(), ref_ann)
    # pinned staging gives the same bytes
    col_p = common.DetectionCollater(resize=40, max_annots_num=10, pinned=True)
    for _ in range(5):  # more batches than ring slots: buffers are reused behind their events
        assert np.array_equal(col_p(samples)['image'].permute(0, 2, 3, 1).cpu().numpy(), ref_img)


def test_device_prefetcher_hands_out_the_same_batches():
    from sacv_amd.detection import common
    loader = [_samples(10 + i, ((30 + i, 41, 2), (25, 50 - i, 1))) for i in range(5)]
    col = common.DetectionCollater(resize=40, max_annots_num=4, pinned=True)
    plain = common.DetectionCollater(resize=40, max_annots_num=4)
    n = 0
    for batch, samples in zip(common.DevicePrefetcher(loader, col), loader):
        ref = plain(samples)
        assert batch['image'].is_cuda and batch['annots'].is_cuda
        # consume on the caller's stream (what a training step does), then compare
        y = batch['image'] * 2.0
        assert torch.equal(y, ref['image'] * 2.0) and torch.equal(batch['annots'].cpu(), ref['annots'])
        assert np.array_equal(batch['scale'], ref['scale'])
        n += 1
    assert n == 5 and len(common.DevicePrefetcher(loader, col)) == 5
    assert list(common.DevicePrefetcher([], col)) == []


H, W, BS, NIMG = 96, 128, 2, 5
IMAGE_IDS = [500 + i for i in range(NIMG)]
CAT = {c: c + 1 for c in range(80)}


def _eval_setup():
    from sacv_amd.detection import models, decode
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    sd = omodel.init_retinanet_state(0)
    sd['cls_head.cls_out.bias'] = sd['cls_head.cls_out.bias'] + 2.5  # random init scores 0.01: lift some over 0.05
    net.load_state_dict(sd)
    net = net.cuda().eval()
    dec = decode.__dict__['RetinaDecoder'](min_score_threshold=0.05)
    batches = []
    for i in range(-(-NIMG // BS)):
        b = min(BS, NIMG - i * BS)
        batches.append({'image': torch.tensor(synth.images(300 + i, b, H, W)),
                        'annots': torch.tensor(synth.annotations(400 + i, b, H, W, boxes_per_image=4)),
                        'scale': np.array([0.5 + 0.25 * j for j in range(b)], dtype=np.float32),
                        'size': np.array([[H * 2 - 7, W * 2 - 3]] * b, dtype=np.float32)})
    return net, dec, batches


def test_evaluation_pipeline_matches_the_reference_statements():
    """One rank: records == the reference's per-image statements (scripts.py:742-789) applied by NumPy to this
    decoder's own [B,100] host outputs (the decoder itself is pinned elsewhere)."""
    from sacv_amd import evaluate as ev
    from sacv_amd.detection import losses
    net, dec, batches = _eval_setup()
    crit = losses.__dict__['RetinaLoss']()
    out = ev.evaluate_coco_detection(batches, ev.detection_predict_fn(net, dec, crit), IMAGE_IDS, CAT, BS)
    want, k, loss_sum = [], 0, 0.0
    for b in batches:
        with torch.no_grad():
            outs = net(b['image'].cuda())
            loss_sum += float(sum(crit(outs, b['annots'].cuda()).values())) * len(b['scale'])
        s, c, bx = dec(outs)
        bx = odecode.eval_postprocess(bx, b['scale'], b['size'])
        for i in range(len(b['scale'])):
            for j in range(s.shape[1]):
                if int(c[i, j]) == -1:
                    break
                want.append({'image_id': IMAGE_IDS[k], 'category_id': CAT[int(c[i, j])], 'score': float(s[i, j]),
                             'bbox': bx[i, j].tolist()})
            k += 1
    assert len(want) >= 20 and out['results'] == want and out['image_ids'] == IMAGE_IDS
    assert abs(out['test_loss'] - loss_sum / NIMG) <= 1e-6 * abs(loss_sum / NIMG)
    assert all(r['bbox'][2] >= 0 and r['bbox'][0] >= 0 for r in want)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sacv_amd import evaluate as ev
        net, dec, batches = _eval_setup()
        out = ev.evaluate_coco_detection(batches, ev.detection_predict_fn(net, dec), IMAGE_IDS, CAT, BS,
                                         process_group=dist.group.WORLD)
        ret[rank] = (out['results'], out['image_ids'])
    finally:
        dist.destroy_process_group()


def test_sharded_evaluation_two_ranks_shared_gpu():
    from sacv_amd import evaluate as ev
    net, dec, batches = _eval_setup()
    single = ev.evaluate_coco_detection(batches, ev.detection_predict_fn(net, dec), IMAGE_IDS, CAT, BS)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        assert ret[r][0] == single['results'] and ret[r][1] == single['image_ids']


@pytest.mark.parametrize('kind', ['retina', 'fcos'])
def test_single_image_detect_like_the_gradio_demo(kind):
    """gradio_demo/3.gradio_detect_single_image.py::predict (SURVEY.md 8f-3's smoke test) on this path: ONE image, padded
    to a multiple of 32 only (736 x 992: no multiple of the coarsest stride), model.eval(), decoder(outputs), then
    boxes /= scale, keep classes > -1, clip to the original size.  The network outputs are compared with the CPU oracle
    model (eval mode), the decoder with the oracle decoder on the SAME device outputs (bit-exact), the post-steps with
    their NumPy statements.  Random init gives scores around the 0.01 prior: the threshold is set just below it so that
    thousands of candidates reach the top-1000 selection and the NMS."""
    from sacv_amd.detection import models, decode
    fcos = kind == 'fcos'
    H, W, scale, orig = 736, 992, 1.55, (470, 640)  # resized 728 x 992 image in a 736 x 992 canvas
    img = np.zeros((1, 3, H, W), dtype=np.float32)
    img[:, :, :728, :] = synth.images(9, 1, 728, W)
    sd = omodel.init_retinanet_state(0, fcos=fcos)
    net = models.__dict__['resnet50_fcos' if fcos else 'resnet50_retinanet']()
    net.load_state_dict(sd)
    net = net.cuda().eval()
    x = torch.tensor(img)
    with torch.no_grad():
        outs = net(x.cuda())
        ref = (omodel.fcos_forward if fcos else omodel.retinanet_forward)({k: v.clone() for k, v in sd.items()}, x,
                                                                          training=False)
    for grp, rgrp in zip(outs, ref):
        for o, r in zip(grp, rgrp):
            assert tuple(o.shape) == tuple(r.shape)
            assert float((o.cpu() - r).abs().max()) <= 2e-4 * max(1.0, float(r.abs().max()))
    thr = 0.0099 if not fcos else 0.05
    name = 'FCOSDecoder' if fcos else 'RetinaDecoder'
    dec = decode.__dict__[name](min_score_threshold=thr)
    scores, classes, boxes = dec(outs)
    host = [[o.cpu().numpy() for o in grp] for grp in outs]
    rs, rc, rb = (odecode.fcos_decode if fcos else odecode.retina_decode)(*host, min_score_threshold=thr)
    assert np.array_equal(scores, rs) and np.array_equal(classes, rc) and np.array_equal(boxes, rb)
    assert (classes[0] > -1).sum() >= 20, (classes[0] > -1).sum()
    # the demo's post-steps, in its order
    boxes /= scale
    s, c, b = scores.squeeze(0), classes.squeeze(0), boxes.squeeze(0)
    s, b, c = s[c > -1], b[c > -1], c[c > -1]
    b[:, 0] = np.maximum(b[:, 0], 0)
    b[:, 1] = np.maximum(b[:, 1], 0)
    b[:, 2] = np.minimum(b[:, 2], orig[1])
    b[:, 3] = np.minimum(b[:, 3], orig[0])
    assert len(s) == len(b) == len(c) and np.all(b[:, 2] <= orig[1]) and np.all(b[:, 3] <= orig[0]) and np.all(b >= 0)
    assert np.all(np.diff(s) <= 0)  # kept boxes come out in descending score
