"""SURVEY.md 8f-3, host logic on CPU: the sharded evaluation loop (sacv_amd.evaluate) produces, for any number of
ranks, exactly the result list the reference's single loop builds (tools/scripts.py:693-838) -- checked with a stub
`predict` (the device pipeline has its own -m gpu test) at world 1 and over gloo at world 2 -- and the padding-skip
canvas arithmetic of the collater (8f-2)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_IMAGES, BS, M = 23, 4, 6  # 6 batches, the last one ragged (3 images)
IMAGE_IDS = [1000 + 7 * i for i in range(N_IMAGES)]
CAT = {c: 10 + 3 * c for c in range(80)}


def make_batches():
    rng = np.random.RandomState(5)
    batches = []
    for i in range(-(-N_IMAGES // BS)):
        b = min(BS, N_IMAGES - i * BS)
        scores = np.sort(rng.rand(b, M).astype(np.float32), axis=1)[:, ::-1].copy()
        classes = rng.randint(0, 80, size=(b, M)).astype(np.float32)
        boxes = (rng.rand(b, M, 4) * 100).astype(np.float32)
        for r in range(b):  # ragged tails padded like the decoders pad: score = class = -1, box = 0
            k = rng.randint(0, M + 1)
            scores[r, k:], classes[r, k:], boxes[r, k:] = -1, -1, 0
        batches.append({'scores': scores, 'classes': classes, 'boxes': boxes, 'loss': float(rng.rand())})
    return batches


def stub_predict(batch):
    return batch['scores'], batch['classes'], batch['boxes'], batch['loss']


def reference_loop(batches):
    """The reference's single loop (scripts.py:760-789, :731-733) written out plainly."""
    results, seen, loss_sum, n = [], [], 0.0, 0
    ids = list(range(N_IMAGES))
    for i, b in enumerate(batches):
        per_batch_ids = ids[i * BS:(i + 1) * BS]
        loss_sum += b['loss'] * len(per_batch_ids)
        n += len(per_batch_ids)
        for s_row, c_row, b_row, index in zip(b['scores'], b['classes'], b['boxes'], per_batch_ids):
            for s, c, box in zip(s_row, c_row, b_row):
                if int(c) == -1:
                    break
                results.append({'image_id': IMAGE_IDS[index], 'category_id': CAT[int(c)], 'score': float(s),
                                'bbox': box.tolist()})
            seen.append(IMAGE_IDS[index])
    return results, seen, loss_sum / n


def test_single_rank_equals_the_reference_loop():
    from sacv_amd import evaluate as ev
    batches = make_batches()
    out = ev.evaluate_coco_detection(batches, stub_predict, IMAGE_IDS, CAT, BS)
    res, seen, loss = reference_loop(batches)
    assert out['results'] == res and out['image_ids'] == seen and abs(out['test_loss'] - loss) < 1e-12
    assert len(res) > 20 and seen == IMAGE_IDS
    # an iterable loader (no __getitem__) is walked, foreign batches skipped
    out2 = ev.evaluate_coco_detection(iter(batches), stub_predict, IMAGE_IDS, CAT, BS)
    assert out2['results'] == res
    # the caller's scorer (the reference's COCOeval lines) receives the merged list
    out3 = ev.evaluate_coco_detection(batches, stub_predict, IMAGE_IDS, CAT, BS, scorer=lambda r, i: (len(r), len(i)))
    assert out3['stats'] == (len(res), N_IMAGES)
    assert ev.shard_batches(7, 1, 3) == [1, 4] and ev.shard_batches(2, 3, 4) == []
    import pytest
    with pytest.raises(RuntimeError, match='missing or duplicated'):
        ev.merge_shards([[(0, [], [], 0.0, 1)], [(2, [], [], 0.0, 1)]])


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sacv_amd import evaluate as ev
        batches = make_batches()
        calls = []

        def predict(batch):
            calls.append(batch['loss'])
            return stub_predict(batch)

        out = ev.evaluate_coco_detection(batches, predict, IMAGE_IDS, CAT, BS, process_group=dist.group.WORLD)
        ret[rank] = (out['results'], out['image_ids'], out['test_loss'], len(calls))
    finally:
        dist.destroy_process_group()


def test_sharded_over_two_ranks_equals_the_single_loop():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    res, seen, loss = reference_loop(make_batches())
    for r in range(world):
        got = ret[r]
        assert got[0] == res and got[1] == seen and abs(got[2] - loss) < 1e-12
        assert got[3] == 3  # 6 batches: each rank ran only its own three (the reference runs all six on every rank)


def test_padding_skip_canvas():
    from sacv_amd.detection import common
    col = common.DetectionCollater(resize=800, device='cpu', crop_padding=True)
    assert col.resize == 1333
    assert col.canvas([(800, 1333), (800, 1201)]) == (896, 1333)   # width capped at the reference's R
    assert col.canvas([(800, 1199), (750, 1000)]) == (896, 1280)
    assert col.canvas([(1333, 800)]) == (1333, 896)
    assert common.DetectionCollater(resize=640, resize_type='yolo_style', device='cpu', crop_padding=True,
                                    pad_multiple=32).canvas([(480, 640), (600, 400)]) == (608, 640)
