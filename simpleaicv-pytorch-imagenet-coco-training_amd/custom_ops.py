"""torch.library registration of the detector's HIP path (SURVEY.md 7 step 1 / BASELINE north_star: "exposed to Python
through a thin C-ABI/FFI layer as PyTorch-ROCm custom ops").

Two operators carry the whole hot path across the dispatcher:

    sacv::detector_forward(Tensor images, Tensor[] params, int handle, bool training, bool keep) -> Tensor[]
    sacv::detector_backward(Tensor[] grads, int handle) -> Tensor[]

`handle` names a live model (its executor, flat arenas and saved activations); the operators are opaque to the tracer,
have FakeTensor implementations (output geometry from the convolution arithmetic, no GPU needed) and an autograd
formula, so `torch.compile(model)` (config.use_compile, tools/train_detection_model.py:176-178) captures a detector as
ONE graph node instead of falling back to eager at the model boundary.  The eager path (HipDetector.forward without
torch.compile, the trainers' fused step) does not go through the dispatcher: a step enqueues ~700 kernels from one
autograd node and from ctypes calls that cost 2-3 us each.

The BatchNorm running statistics a training forward pass updates belong to the model behind `handle` (like the saved
activations): they are not operands, because torch.library attaches autograd formulas to functional schemas only.
Nothing inside a captured graph reads them; state_dict() / checkpointing happen outside.

Per-kernel operators (conv2d_gemm, bn_train_bwd, ...) are deliberately NOT registered: every launch of the executor
would pay the dispatcher's ~20 us, three times the step's whole host budget.
"""
import itertools
import weakref
from typing import List

import torch

_models = weakref.WeakValueDictionary()
_ids = itertools.count(1)


def register_model(model):
    """-> handle of `model` for the two operators (kept as model._co_handle)."""
    h = getattr(model, '_co_handle', None)
    if h is None or _models.get(h) is not model:
        h = next(_ids)
        _models[h] = model
        model._co_handle = h
    return h


def _model(handle):
    m = _models.get(handle)
    if m is None:
        raise RuntimeError(f'sacv::detector_*: model handle {handle} is not alive')
    return m


def level_sizes(h, w):
    """(H, W) of P3..P7 for an input of h x w: the stem (7x7/s2/p3), the 3x3/s2/p1 max pool, three stride-2 stages and
    the two stride-2 3x3 convolutions P6 / P7 (detection/models/backbones/resnet.py:96-113, fpn.py:60-87)."""
    o = lambda s, k, st, p: (s + 2 * p - k) // st + 1
    h, w = o(h, 7, 2, 3), o(w, 7, 2, 3)
    h, w = o(h, 3, 2, 1), o(w, 3, 2, 1)
    out = []
    for _ in range(3):  # C3, C4, C5
        h, w = o(h, 3, 2, 1), o(w, 3, 2, 1)
        out.append((h, w))
    for _ in range(2):  # P6, P7
        h, w = o(h, 3, 2, 1), o(w, 3, 2, 1)
        out.append((h, w))
    return out


def output_shapes(model, batch, h, w):
    """Shapes of the flattened forward outputs (group-major, five levels per group)."""
    lv = level_sizes(h, w)
    if getattr(model, '_fcos', False):
        tails = [(model.num_classes,), (4,), (1,)]
    else:
        tails = [(model.num_anchors, model.num_classes), (model.num_anchors, 4)]
    return [(batch, lh, lw) + t for t in tails for lh, lw in lv]


@torch.library.custom_op('sacv::detector_forward', mutates_args=())
def detector_forward(images: torch.Tensor, params: List[torch.Tensor], handle: int, training: bool,
                     keep: bool) -> List[torch.Tensor]:
    model = _model(handle)
    if images.device.type != 'cuda':
        raise RuntimeError('sacv::detector_forward runs on MI355X only (no CPU fallback)')
    groups = model._engine.forward(images, training=training, keep=keep)
    outs = [t for grp in groups for t in grp]
    if keep:
        model._last_outputs = tuple(outs)
    return outs


@detector_forward.register_fake
def _(images, params, handle, training, keep):
    model = _model(handle)
    b, _, h, w = images.shape
    return [images.new_empty(s, dtype=torch.float32) for s in output_shapes(model, b, h, w)]


@torch.library.custom_op('sacv::detector_backward', mutates_args=())
def detector_backward(grads: List[torch.Tensor], handle: int) -> List[torch.Tensor]:
    model = _model(handle)
    n = 5
    gs = [g.contiguous() for g in grads]
    model._engine.backward(*[gs[k:k + n] for k in range(0, len(gs), n)])
    arena = model._arena
    # fresh tensors: an operator's outputs must not alias state that outlives the call (the eager path hands out views
    # of the gradient arena instead; this path is the traced one)
    # (contiguous, as the FakeTensor implementation promises: k x k weights are stored channels_last in the arena)
    return [arena.grad_views[name].clone(memory_format=torch.contiguous_format) for name, _ in model._param_list]


@detector_backward.register_fake
def _(grads, handle):
    model = _model(handle)
    # (shapes only: the parameters themselves are real tensors and must not meet a FakeTensorMode)
    return [grads[0].new_empty(tuple(p.shape), dtype=torch.float32) for _, p in model._param_list]


def _setup_context(ctx, inputs, output):
    ctx.handle = inputs[2]


def _backward(ctx, grads):
    gparams = torch.ops.sacv.detector_backward(list(grads), ctx.handle)
    return None, gparams, None, None, None


detector_forward.register_autograd(_backward, setup_context=_setup_context)
