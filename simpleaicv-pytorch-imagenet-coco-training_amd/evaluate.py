"""Detection evaluation loop with the images SHARDED over the ranks.

The reference evaluates the whole validation set on EVERY rank (tools/train_detection_model.py:85-90 builds the test
loader without a sampler; tools/scripts.py:693-838 evaluate_coco_detection) -- eight GPUs do eight times the same work.
Here rank r takes the batches r, r + world, r + 2 world, ... of the same loader order, runs forward (eval mode) ->
loss -> decoder -> box post-steps on the device, and the per-batch result records are gathered and merged in batch
order, so every rank ends with EXACTLY the list the reference's single loop builds (same entries, same order):
`results` = [{'image_id', 'category_id', 'score', 'bbox'}], `image_ids`, and the mean loss weighted by batch size
(`AverageMeter`, scripts.py:731-733).  No data-path collective: one object gather at the end.

What is NOT rebuilt: COCOeval itself (pycocotools is the reference's third-party scorer, scripts.py:819-828, absent in
this image; mAP is out of scope in SURVEY.md 8c).  `evaluate_coco_detection` returns the list COCO.loadRes() takes, and
calls a `scorer(results, image_ids)` the caller may pass (the reference's three COCOeval lines).

The per-image statements restate scripts.py:742-789: boxes /= scale; clip to the ORIGINAL image size; xyxy -> xywh
(on the device: sacv_eval_postprocess); stop at the first class == -1; category id through coco_label_to_cat_id.
"""
import collections

import numpy as np
import torch
import torch.distributed as dist

__all__ = ['shard_batches', 'merge_shards', 'records_of_batch', 'detection_predict_fn', 'evaluate_coco_detection']


def shard_batches(n_batches, rank, world):
    """Batch indices of rank `rank`: round-robin, so consecutive (similarly sized) batches spread over the ranks."""
    return list(range(rank, n_batches, world))


def records_of_batch(scores, classes, boxes_xywh, indices, image_ids, label_to_cat):
    """scripts.py:760-789 for one batch: scores / classes [B,M], boxes_xywh [B,M,4] host arrays (already rescaled,
    clipped and converted), `indices` the dataset indices of the B images.  -> (results, seen_image_ids)."""
    results, seen = [], []
    for s_row, c_row, b_row, index in zip(scores, classes, boxes_xywh, indices):
        for s, c, b in zip(s_row, c_row, b_row):
            c = int(c)
            if c == -1:
                break
            results.append({'image_id': image_ids[index], 'category_id': label_to_cat[c], 'score': float(s),
                            'bbox': b.tolist()})
        seen.append(image_ids[index])
    return results, seen


def merge_shards(shards):
    """shards: per rank, a list of (batch_index, results, image_ids, loss_sum, n_images).  Merged in batch order =
    the order of the reference's single loop, whatever the number of ranks."""
    rows = sorted((row for shard in shards for row in shard), key=lambda r: r[0])
    idx = [r[0] for r in rows]
    if idx != list(range(len(idx))):
        raise RuntimeError(f'sharded evaluation: batches missing or duplicated after the gather ({idx[:8]}...)')
    results = [rec for r in rows for rec in r[1]]
    image_ids = [i for r in rows for i in r[2]]
    n = sum(r[4] for r in rows)
    loss = sum(r[3] for r in rows) / n if n else 0.0
    return results, image_ids, loss


def detection_predict_fn(model, decoder, criterion=None):
    """The device side of one evaluation batch: forward in eval mode, (loss), decode + NMS, box post-steps; only the
    [B,M] result arrays cross PCIe.  Returns f(batch) -> (scores, classes, boxes_xywh, loss_or_None)."""
    from .detection.common import eval_postprocess

    def predict(batch):
        if getattr(model, 'training', False):
            raise RuntimeError('evaluation needs model.eval(): in train mode BatchNorm would use batch statistics '
                               '(tools/scripts.py calls model.eval() before its evaluation loop)')
        images = batch['image']
        if not images.is_cuda:
            images = images.cuda(non_blocking=True)
        with torch.no_grad():
            outs = model(images)
            loss = None
            if criterion is not None:
                annots = batch['annots']
                annots = annots.cuda(non_blocking=True) if not annots.is_cuda else annots
                loss = float(sum(criterion(outs, annots).values()))
            s, c, b = decoder.device_call(outs)
            b = eval_postprocess(b, batch['scale'], batch['size'])
        return s.cpu().numpy(), c.cpu().numpy(), b.cpu().numpy(), loss

    return predict


def evaluate_coco_detection(test_loader, predict, image_ids, coco_label_to_cat_id, batch_size, process_group=None,
                            scorer=None):
    """test_loader: an indexable or iterable of batches in DATASET ORDER (the reference's un-sampled test loader);
    predict: detection_predict_fn(...) or any f(batch) -> (scores, classes, boxes_xywh, loss_or_None);
    image_ids / coco_label_to_cat_id: the dataset's tables (datasets/cocodataset.py); batch_size: images per batch
    (scripts.py:705, used to map batch i to dataset indices [i*bs, (i+1)*bs))."""
    world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(process_group) if world > 1 else 0
    n_images = len(image_ids)
    n_batches = -(-n_images // batch_size)
    mine = set(shard_batches(n_batches, rank, world))
    shard = []
    if hasattr(test_loader, '__getitem__'):
        stream = ((i, test_loader[i]) for i in sorted(mine))
    else:
        # a plain iterable (e.g. a DataLoader) still LOADS every batch on every rank -- only the inference of the
        # non-owned ones is skipped; pass an indexable loader (or one built on a sampler over shard_batches) to shard the
        # host work too
        stream = ((i, b) for i, b in enumerate(test_loader) if i in mine)
    for i, batch in stream:
        indices = list(range(i * batch_size, min((i + 1) * batch_size, n_images)))
        scores, classes, boxes, loss = predict(batch)
        res, seen = records_of_batch(scores, classes, boxes, indices, image_ids, coco_label_to_cat_id)
        shard.append((i, res, seen, (loss or 0.0) * len(indices), len(indices)))
    if world > 1:
        shards = [None] * world
        dist.all_gather_object(shards, shard, group=process_group)
    else:
        shards = [shard]
    results, seen_ids, loss = merge_shards(shards)
    out = collections.OrderedDict(test_loss=loss, results=results, image_ids=seen_ids)
    if scorer is not None and len(results) > 0:
        out['stats'] = scorer(results, seen_ids)
    return out
