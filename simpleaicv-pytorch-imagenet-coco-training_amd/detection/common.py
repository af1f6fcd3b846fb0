"""Device-side mirrors of the two host steps that sit either side of the training/eval step
(simpleAICV/detection/common.py:223-286 Normalize + DetectionCollater; tools/scripts.py:742-758 the box
post-steps of evaluate_coco_detection).  Same constructor arguments and output contract as the reference;
the pixel work runs as HIP kernels and only uint8 pixels cross PCIe."""
import numpy as np
import torch

from .. import ops

__all__ = ['DetectionCollater', 'eval_postprocess']


class DetectionCollater:
    """collate_fn replacement.  Each sample: {'image': uint8 HWC array (already resized, NOT yet /255),
    'annots': [n,5] float32, 'scale': float, 'size': (h, w)}.  Returns {'image': fp32 CUDA tensor
    [B,3,R,R] (channels_last strides, zero padded, / 255), 'annots': [B,max_annots_num,5] padded with -1,
    'scale': float32 [B], 'size': float32 [B,2]} -- the reference's output with Normalize folded in."""

    def __init__(self, resize=800, resize_type='retina_style', max_annots_num=100, device='cuda'):
        assert resize_type in ['retina_style', 'yolo_style']
        self.resize = resize
        if resize_type == 'retina_style':
            self.resize = int(round(self.resize * 1333. / 800))
        self.max_annots_num = max_annots_num
        self.device = device

    def __call__(self, data):
        imgs = []
        for s in data:
            im = np.ascontiguousarray(s['image'])
            if im.dtype != np.uint8:
                raise TypeError('sacv_amd DetectionCollater takes uint8 images (Normalize runs on the device)')
            imgs.append(torch.from_numpy(im).to(self.device, non_blocking=True))
        image = ops.collate_u8(imgs, self.resize)
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


def eval_postprocess(boxes, scales, sizes):
    """boxes: fp32 CUDA [B,M,4] from a decoder's device_call; scales [B], sizes [B,2] array-likes."""
    dev = boxes.device
    sc = torch.as_tensor(np.asarray(scales, dtype=np.float32)).to(dev)
    sz = torch.as_tensor(np.asarray(sizes, dtype=np.float32)).to(dev)
    return ops.eval_postprocess(boxes, sc, sz)
