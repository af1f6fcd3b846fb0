"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm).

One process per GPU.  The gradient arena is a single flat fp32 buffer whose >=2-D section is
laid out in module order (backbone ... fpn, cls_head, reg_head), i.e. in REVERSE of the order
the backward pass finishes them.  The exchanger cuts it into a few large contiguous stages
(heads | fpn + layer4 | layer3 | rest) and launches one asynchronous all-reduce per stage as soon as
the backward pass has produced that stage, so the transfer over xGMI runs under the remaining
backward convolutions; the optimizer waits for all of them.  Stages are zero-copy slices, so
there is no bucket packing.  Replaces the DDP wrapper of tools/utils.py:178-180 (mean of the
gradients over ranks; the 1/world factor is folded into the fused AdamW kernel) and DDP's
per-forward broadcast of the BatchNorm buffers from rank 0.

The class is device-agnostic (works on CPU tensors with the gloo backend) so that the
world_size>1 logic is covered by CPU tests.
"""
import torch
import torch.distributed as dist


def stage_ranges(arena_order, offsets, numel, split):
    """Contiguous [begin, end) element ranges of the gradient arena per backward stage.

    arena_order: [(name, tensor)] in arena order (1-D tensors first, then >=2-D in module order).
    Returns dict stage -> (begin, end):  'heads' (cls_head/reg_head weights; the classifier's fc),
    'neck' (fpn + backbone layer4 weights), 'mid' (backbone layer3 weights, 28 MB), 'rest' (everything else
    including the whole 1-D section: ~6 MB, the only transfer left exposed at the end of the backward pass)."""
    multi = [(n, offsets[n]) for n, t in arena_order if t.ndim != 1]

    def first(prefixes):
        for n, o in multi:
            if n.startswith(prefixes):
                return o
        return numel

    heads_begin = first(('cls_head.', 'reg_head.', 'clsregcnt_head.', 'fc.', 'decoder.'))
    neck_begin = min(first(('backbone.layer4.', 'layer4.')), first(('fpn.',)))
    neck_begin = min(neck_begin, heads_begin)
    mid_begin = min(first(('backbone.layer3.', 'layer3.')), neck_begin)
    return {'heads': (heads_begin, numel), 'neck': (neck_begin, heads_begin), 'mid': (mid_begin, neck_begin),
            'rest': (0, mid_begin)}


class RcclComm:
    """Communicator of the C ABI (`sacv_comm_*`, include/sacv_hip.h): RCCL called directly from libsacv_hip.so, the
    bucket all-reduces on a communicator-owned side HIP stream that is event-chained to the launch stream.  The
    rendezvous id travels over the already initialised torch.distributed group (any backend)."""

    def __init__(self, group=None):
        import ctypes
        from ._lib import lib, check
        self._lib, self._check = lib, check
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        nbytes = lib.sacv_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(nbytes)
        if self.rank == 0:
            check(lib.sacv_comm_unique_id(buf), 'comm_unique_id')
        box = [bytes(buf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                       group=group)
        handle = ctypes.c_void_p()
        check(lib.sacv_comm_init(ctypes.create_string_buffer(box[0], nbytes), self.rank, self.world,
                                 ctypes.byref(handle)), 'comm_init')
        self.handle = handle

    def all_reduce(self, t):
        self._check(self._lib.sacv_allreduce_bucket(self.handle, t.data_ptr(), t.numel(),
                                                    torch.cuda.current_stream().cuda_stream), 'allreduce_bucket')

    def wait(self):
        self._check(self._lib.sacv_comm_wait(self.handle, torch.cuda.current_stream().cuda_stream), 'comm_wait')

    def broadcast(self, t, root=0):
        self._check(self._lib.sacv_comm_broadcast(self.handle, t.data_ptr(), t.numel(), root,
                                                  torch.cuda.current_stream().cuda_stream), 'comm_broadcast')

    def close(self):
        if self.handle:
            self._lib.sacv_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GradExchanger:
    """transport 'torch': torch.distributed all-reduces (backend "nccl" = RCCL on ROCm, or gloo on CPU tensors);
    transport 'rccl': the C ABI's sacv_allreduce_bucket (RcclComm).  Same stage protocol either way."""

    def __init__(self, grads_flat, ranges, group=None, comm=None):
        self.g = grads_flat
        self.ranges = ranges
        self.group = group
        self.comm = comm
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if comm is not None:
            self.world = comm.world
        self.pending = []
        self.enabled = True  # False inside a gradient-accumulation window (DDP.no_sync)
        # One-GPU measurement aid (bench.py --emulate-rccl): dict(workgroups, lds_bytes, gbps) -- at world
        # size 1 every stage launches a stand-in for its all-reduce's kernels (sacv_comm_emulate_load: that many resident
        # workgroups for the time an 8-rank ring all-reduce of the stage would take at `gbps` GB/s) on a stream of its
        # own, chained exactly like the real collective
        self.emulate = None
        self._emu_stream, self._emu_scratch, self._emu_busy = None, None, False

    def _emulate_stage(self, stage):
        from ._lib import lib, check
        b, e = self.ranges[stage]
        if e <= b:
            return
        em = self.emulate
        us = 2.0 * 7 / 8 * (e - b) * 4 / (em.get('gbps', 200.0) * 1e3)
        if self._emu_stream is None:
            self._emu_stream = torch.cuda.Stream()
            self._emu_scratch = torch.zeros(4 << 20, device=self.g.device, dtype=torch.float32)
        cur = torch.cuda.current_stream()
        self._emu_stream.wait_stream(cur)
        check(lib.sacv_comm_emulate_load(int(em['workgroups']), int(em.get('lds_bytes', 16384)), float(us),
                                         self._emu_scratch.data_ptr(), self._emu_scratch.numel() * 4,
                                         self._emu_stream.cuda_stream), 'comm_emulate_load')
        self._emu_busy = True

    def stage_done(self, stage):
        """Called by the backward pass when every gradient of `stage` has been enqueued."""
        if self.world == 1 and self.emulate and self.enabled:
            return self._emulate_stage(stage)
        if self.world == 1 or not self.enabled:
            return
        b, e = self.ranges[stage]
        if e <= b:
            return
        if self.comm is not None:
            self.comm.all_reduce(self.g[b:e])
            self.pending.append(None)
        else:
            self.pending.append(dist.all_reduce(self.g[b:e], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True))

    def wait(self):
        """Make the current stream wait for all outstanding all-reduces (sums; divide by world)."""
        if self._emu_busy:
            torch.cuda.current_stream().wait_stream(self._emu_stream)
            self._emu_busy = False
        if self.comm is not None:
            if self.pending:
                self.comm.wait()
        else:
            for w in self.pending:
                w.wait()
        self.pending = []


def broadcast_buffers(flat_buffers, group=None, src=0, comm=None):
    """DDP's broadcast_buffers=True: rank 0's BatchNorm running statistics overwrite every rank's
    before each forward (one flat 208 KB broadcast instead of 106 tensors)."""
    if comm is not None:
        if comm.world > 1 and flat_buffers.numel() > 0:
            comm.broadcast(flat_buffers, src)
        return
    if dist.is_initialized() and dist.get_world_size(group) > 1 and flat_buffers.numel() > 0:
        dist.broadcast(flat_buffers, src=src, group=group)


def vote_and_reduce_scalars(values, group=None, comm=None):
    """tools/scripts.py:963-966 (skip-batch vote) and :1017-1029 (loss terms averaged over ranks)
    folded into ONE small all-reduce: `values` is a float tensor [k]; returns its sum over ranks.  With the C ABI's
    communicator (`comm`) it travels there, so that a process never has collectives of two communicators in flight."""
    if comm is not None:
        if comm.world > 1:
            values = values.contiguous()
            comm.all_reduce(values)
            comm.wait()
        return values
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values
