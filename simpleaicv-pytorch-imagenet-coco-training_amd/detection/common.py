"""Device-side mirrors of the two host steps that sit either side of the training/eval step
(simpleAICV/detection/common.py:223-286 Normalize + DetectionCollater; tools/scripts.py:742-758 the box
post-steps of evaluate_coco_detection).  Same constructor arguments and output contract as the reference;
the pixel work runs as HIP kernels and only uint8 pixels cross PCIe."""
import numpy as np
import torch

from .. import ops

__all__ = ['DetectionCollater', 'DevicePrefetcher', 'eval_postprocess']


class _PinnedRing:
    """Page-locked staging buffers for the uint8 pixels: `slots` generations of buffers, each guarded by the event of
    the copy that last read it, so an asynchronous H2D copy is never overtaken by the host refilling its source."""

    def __init__(self, slots=3):
        self.slots = [dict(bufs=[], event=None) for _ in range(slots)]
        self.turn = 0

    def stage(self, arrays, device):
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        if slot['event'] is not None:
            slot['event'].synchronize()  # the copies issued from this slot `slots` batches ago
        out = []
        for i, a in enumerate(arrays):
            n = a.size
            if i >= len(slot['bufs']) or slot['bufs'][i].numel() < n:
                buf = torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory()
                if i >= len(slot['bufs']):
                    slot['bufs'].append(buf)
                else:
                    slot['bufs'][i] = buf
            view = slot['bufs'][i][:n].view(a.shape)
            view.numpy()[...] = a  # host memcpy into page-locked memory
            out.append(view.to(device, non_blocking=True))
        slot['event'] = torch.cuda.Event()
        slot['event'].record()
        return out


class DetectionCollater:
    """collate_fn replacement.  Each sample: {'image': uint8 HWC array (already resized, NOT yet /255),
    'annots': [n,5] float32, 'scale': float, 'size': (h, w)}.  Returns {'image': fp32 CUDA tensor
    [B,3,R,R] (channels_last strides, zero padded, / 255), 'annots': [B,max_annots_num,5] padded with -1,
    'scale': float32 [B], 'size': float32 [B,2]} -- the reference's output with Normalize folded in.

    Two opt-in extras of the device-side data boundary, both off by default = the reference's output:
      crop_padding=True  the canvas is the largest image of the batch rounded up to `pad_multiple` (the network's
                         coarsest stride, 128) instead of the R x R square: an 800 x 1333 batch is convolved as
                         896 x 1408 (1.26 M pixels) instead of 1333 x 1333 (1.78 M).  It CHANGES the BatchNorm batch
                         statistics (fewer zero pixels) and the anchor grid, so it is a recipe change, not a parity
                         mode; annotations are unaffected (images stay at the top-left corner).
      pinned=True        pixels go through page-locked staging buffers so the H2D copies are asynchronous (used by
                         DevicePrefetcher to overlap them with the previous step)."""

    def __init__(self, resize=800, resize_type='retina_style', max_annots_num=100, device='cuda', crop_padding=False,
                 pad_multiple=128, pinned=False):
        assert resize_type in ['retina_style', 'yolo_style']
        self.resize = resize
        if resize_type == 'retina_style':
            self.resize = int(round(self.resize * 1333. / 800))
        self.max_annots_num = max_annots_num
        self.device = device
        self.crop_padding, self.pad_multiple = bool(crop_padding), int(pad_multiple)
        self.ring = _PinnedRing() if pinned else None

    def canvas(self, shapes):
        """(Hc, Wc) of the padding-skip canvas for images of the given (h, w): the batch maximum rounded up to
        pad_multiple, never beyond the reference's R."""
        m = self.pad_multiple
        hc = min(self.resize, -(-max(h for h, _ in shapes) // m) * m)
        wc = min(self.resize, -(-max(w for _, w in shapes) // m) * m)
        return hc, wc

    def __call__(self, data):
        arrays = []
        for s in data:
            im = np.ascontiguousarray(s['image'])
            if im.dtype != np.uint8:
                raise TypeError('sacv_amd DetectionCollater takes uint8 images (Normalize runs on the device)')
            arrays.append(im)
        if self.ring is not None:
            imgs = self.ring.stage(arrays, self.device)
        else:
            imgs = [torch.from_numpy(im).to(self.device, non_blocking=True) for im in arrays]
        canvas = self.canvas([im.shape[:2] for im in arrays]) if self.crop_padding else None
        image = ops.collate_u8(imgs, self.resize, canvas=canvas)
        annots = np.full((len(data), self.max_annots_num, 5), -1, dtype=np.float32)
        for i, s in enumerate(data):
            a = np.asarray(s['annots'], dtype=np.float32)
            if a.shape[0] > 0:
                annots[i, :a.shape[0], :] = a
        return {
            'image': image,
            'annots': torch.from_numpy(annots),
            'scale': np.array([s['scale'] for s in data], dtype=np.float32),
            'size': np.array([s['size'] for s in data], dtype=np.float32),
        }


class DevicePrefetcher:
    """Overlaps the data boundary of batch i+1 with the training step of batch i (the reference's
    loop does `images.cuda()` synchronously at the top of every iteration, tools/scripts.py:872-876).

    `loader` yields lists of samples (a DataLoader with collate_fn=lambda x: x) or anything `collate` accepts;
    `collate` is a DetectionCollater(pinned=True).  Collation -- pinned staging, asynchronous H2D of the uint8 pixels,
    the collate kernel, the annotations' H2D -- is issued on a side HIP stream one batch ahead; iteration hands out
    batches whose tensors are device-resident and makes the CALLER's current stream wait on the side stream's event,
    so no host synchronisation happens here."""

    def __init__(self, loader, collate, device='cuda'):
        self.loader, self.collate, self.device = loader, collate, device
        self.stream = torch.cuda.Stream(device=device)

    def _issue(self, samples):
        with torch.cuda.stream(self.stream):
            batch = self.collate(samples)
            batch['annots'] = batch['annots'].pin_memory().to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return batch, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._issue(next(it))
        except StopIteration:
            return
        while nxt is not None:
            batch, ev = nxt
            try:
                nxt = self._issue(next(it))  # batch i+1 is in flight while the caller works on batch i
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for v in batch.values():  # the caching allocator must not hand these blocks to the side stream's next
                if isinstance(v, torch.Tensor) and v.is_cuda:  # batch while the caller's stream still reads them
                    v.record_stream(cur)
            yield batch

    def __len__(self):
        return len(self.loader)


def eval_postprocess(boxes, scales, sizes):
    """boxes: fp32 CUDA [B,M,4] from a decoder's device_call; scales [B], sizes [B,2] array-likes."""
    dev = boxes.device
    sc = torch.as_tensor(np.asarray(scales, dtype=np.float32)).to(dev)
    sz = torch.as_tensor(np.asarray(sizes, dtype=np.float32)).to(dev)
    return ops.eval_postprocess(boxes, sc, sz)
