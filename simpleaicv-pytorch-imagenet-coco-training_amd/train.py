"""Training-step driver: the build's restatement of tools/scripts.py::train_detection (:852-1050) and
::train_classification (:108-267) for one iteration, with the optimizer of tools/utils.py:275-582 (two parameter
groups: 1-D tensors without weight decay, >=2-D tensors with it) as ONE fused kernel over the flat parameter arena.

Per iteration, in the reference's order:
  input guard (inf/nan in images or targets, :878-884) -> forward (rank-0 BN buffers broadcast first, as DDP does)
  -> loss dict -> zero/inf/nan loss guard (:918-930) -> backward of loss / accumulation_steps, with the gradient
  all-reduce of each finished stage overlapped under the remaining backward kernels -- only on the last micro-batch
  of an accumulation window (DDP.no_sync on the others, :886-949) -> optional nan-gradient guard (:951-961) -> skip
  vote over the ranks (:963-974) -> clip_grad_value_ / clip_grad_norm_ (:978-990) -> optimizer (mean over ranks
  folded in) -> loss terms.
The guards, the vote and the optimizer gate run on the device (`nonfinite_flag` / `loss_skip_flag` -> one-float
all-reduce -> `step_gate` -> gated optimizer kernel that leaves parameters, moments and the step count untouched when
any rank votes to skip), so with accumulation_steps == 1 an iteration needs NO host synchronisation; the loss averaging
(:1017-1029) travels in one small all-reduce in `finish_step`.  With accumulation_steps > 1 the vote is read back
once per micro-batch, because the reference's control flow depends on it (a skipped micro-batch drops the
accumulated gradients and does not advance the iteration index, :968-974).

Learning rates live in `param_groups` exactly like a torch optimizer's, so the reference's `Scheduler.step(optimizer,
epoch)` (tools/utils.py:199-272; mirrored in sacv_amd.utils.Scheduler) drives a trainer unchanged;
`state_dict()` / `load_state_dict()` use torch.optim's layout, so a reference checkpoint's `optimizer_state_dict`
(tools/train_detection_model.py:141-142, 252-267) resumes here and vice versa.
"""
import torch
import torch.distributed as dist

from . import ops
from .parallel import GradExchanger, RcclComm, stage_ranges, broadcast_buffers, vote_and_reduce_scalars


class _StepDriver:
    """What both trainers share: parameter groups, guards, accumulation, exchange, clipping, checkpoint state."""

    STAGES = ('heads', 'neck', 'mid', 'rest')

    def _init_driver(self, model, lr, weight_decay, accumulation_steps, clip_grad_value, clip_max_norm,
                     skip_inf_nan_grad, process_group, transport):
        assert accumulation_steps >= 1, 'illegal accumulation_steps!'
        self.model = model
        self.accumulation_steps = int(accumulation_steps)
        self.clip_grad_value = float(clip_grad_value or 0.0)
        self.clip_max_norm = float(clip_max_norm or 0.0)
        self.skip_inf_nan_grad = bool(skip_inf_nan_grad)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        arena = model.arena
        self._arena = arena  # the trainer's buffers are sized for THIS arena (see _check_arena)
        names_1d = [n for n, p in arena.order if p.ndim == 1]
        names_nd = [n for n, p in arena.order if p.ndim != 1]
        # build_optimizer's two groups (tools/utils.py:480-582); Scheduler.step rewrites their 'lr'
        self.param_groups = [
            {'lr': lr, 'weight_decay': 0.0, 'initial_lr': lr, 'param_names': names_1d},
            {'lr': lr, 'weight_decay': weight_decay, 'initial_lr': lr, 'param_names': names_nd},
        ]
        dev = arena.params.device
        # device-side step state, layout in csrc/optim.hip
        self.state = torch.zeros(ops.STATE_FLOATS, device=dev, dtype=torch.float32)
        self.flag = torch.zeros(1, device=dev, dtype=torch.float32)
        self.accum = torch.zeros_like(arena.grads) if self.accumulation_steps > 1 else None
        self.micro = 0       # micro-batches accumulated in the current window
        self.ema_model = None  # a utils.EmaModel: updated after every optimizer step (tools/scripts.py:1013-1015)
        self.iter_index = 1  # the reference's iter_index (advances only on iterations that are not skipped)
        if hasattr(transport, 'all_reduce') and hasattr(transport, 'wait') and hasattr(transport, 'broadcast'):
            self.comm = transport if self.world > 1 else None  # a caller-built communicator with RcclComm's surface
        elif transport in ('torch', 'rccl'):
            self.comm = RcclComm(process_group) if (transport == 'rccl' and self.world > 1) else None
        else:
            raise ValueError("transport must be 'torch' (torch.distributed), 'rccl' (sacv_allreduce_bucket) or a "
                             "communicator object with RcclComm's all_reduce / wait / broadcast")
        self.exchanger = GradExchanger(arena.grads, stage_ranges(arena.order, arena.offsets, arena.numel, arena.split),
                                       process_group, comm=self.comm)
        if self.world > 1:
            # the collective's kernels share the chip with the backward pass: the persistent convolution grids plan for 224
            # of the 256 CUs there (free on one GPU, and it takes the stand-in collective's 0.3 ms away)
            engine = getattr(model, 'engine', None)
            if engine is not None and getattr(engine, 'BWD_SHARE', 0) is None:
                engine.BWD_SHARE = (3, 2, 3, 2, 32)
        if self.world > 1:  # DDP init_sync: every rank starts from rank 0's parameters and buffers
            if self.comm is not None:
                self.comm.broadcast(arena.params, 0)
            else:
                dist.broadcast(arena.params, src=0, group=process_group)
            broadcast_buffers(arena.buffers, process_group, comm=self.comm)

    def close(self):
        """Release the C ABI's communicator (transport 'rccl').  Call it BEFORE dist.destroy_process_group(): an
        ncclCommDestroy issued from __del__ at interpreter shutdown can run after the process group (and the HIP
        runtime's streams) are gone."""
        comm, self.comm = self.comm, None
        if comm is not None:
            self.exchanger.wait()
            self.exchanger.comm = None
            if hasattr(comm, 'close'):
                comm.close()

    # -- torch.optim-style surface ---------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """The backward pass overwrites the gradient arena; only an open accumulation window holds state."""
        self.micro = 0

    def start_epoch(self):
        """The reference restarts `iter_index` at 1 in every epoch (tools/scripts.py:866) and leaves gradients of an
        unfinished accumulation window in place; so does this."""
        self.iter_index = 1

    def _check_arena(self):
        if self.model.arena is not self._arena:
            raise RuntimeError('the model re-allocated its parameter arena after this trainer was built (.to(), '
                               '.cuda(), deepcopy or parameter assignment): the trainer still holds the old '
                               'gradient buffer, optimizer state and exchange stages -- build a new trainer')

    # -- pieces of one iteration ------------------------------------------------------------------------------
    def _begin(self, *inputs):
        """Input guard (tools/scripts.py:878-884) + DDP's per-forward buffer broadcast."""
        self._check_arena()
        if self.world > 1:
            broadcast_buffers(self._arena.buffers, self.group, comm=self.comm)
        ops.fill(self.flag, 0.0)
        for t in inputs:
            if t.is_floating_point():
                ops.nonfinite_flag(t, self.flag)
        self.boundary = (self.iter_index % self.accumulation_steps == 0)
        self.exchanger.enabled = self.boundary  # DDP.no_sync on the other micro-batches
        self._vote = None

    def _vote_begin(self):
        """All inputs of the skip vote are known once the loss flag is set (unless the gradients are guarded too):
        start the one-float all-reduce now so that it completes under the backward pass."""
        if self.world > 1 and not self.skip_inf_nan_grad:
            if self.comm is not None:
                # transport 'rccl': the vote travels on the SAME communicator and side stream as the gradient stages
                # (strictly ordered with them).  A second communicator -- torch's -- with collectives in flight at the
                # same time would be two RCCL kernels whose launch order may differ between ranks: a hang hazard.
                self.comm.all_reduce(self.flag)
                self._vote = 'comm'
            else:
                self._vote = dist.all_reduce(self.flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _stage_cb(self):
        """The engines' per-stage callback, or None when a stage has nothing to do (one rank, no accumulation window):
        the engine then skips the stage's event chaining altogether."""
        return self._on_stage if (self.world > 1 or self.accum is not None or self.exchanger.emulate) else None

    def _on_stage(self, stage):
        """Backward finished the gradients of `stage`: fold in the accumulation window, start the exchange.  Called with
        the engine's weight-gradient stream current (DetectorEngine._stage): what is enqueued here runs behind the
        stage's last weight gradient without holding up the launch stream."""
        if self.accum is not None and self.boundary and self.micro > 0:
            b, e = self.exchanger.ranges[stage]
            if e > b:
                ops.eltwise(ops.EW_ADD, self._arena.grads[b:e], self.accum[b:e], out=self._arena.grads[b:e])
        self.exchanger.stage_done(stage)

    def _after_backward(self):
        """-> True when this iteration is known (host-side) to be skipped; only accumulation_steps > 1 reads the vote
        back, accumulation_steps == 1 keeps it on the device."""
        arena = self._arena
        if self.skip_inf_nan_grad:
            ops.nonfinite_flag(arena.grads, self.flag, nan_only=True)  # the reference tests isnan twice (:957-958)
        if self.comm is not None and self.world > 1:
            if self._vote is None:
                self.comm.all_reduce(self.flag)
            self.comm.wait()  # in order behind the stages: the optimizer needs those too
            self._vote = None
        elif self._vote is not None:
            self._vote.wait()
            self._vote = None
        elif self.world > 1:
            dist.all_reduce(self.flag, op=dist.ReduceOp.SUM, group=self.group)
        if self.accumulation_steps == 1:
            return False
        if float(self.flag) > 0:  # optimizer.zero_grad(); continue  (:968-974)
            self.exchanger.wait()
            self.micro = 0
            ops.step_gate(self.flag, self.state)  # records the skip for finish_step
            return True
        if not self.boundary:
            if self.micro == 0:
                self.accum.copy_(arena.grads)
            else:
                ops.eltwise(ops.EW_ADD, self.accum, arena.grads, out=self.accum)
            self.micro += 1
        return False

    def _gate_and_clip(self, beta1=0.9, beta2=0.999):
        ops.step_gate(self.flag, self.state, beta1, beta2)
        self.exchanger.wait()
        if self.clip_grad_value > 0 or self.clip_max_norm > 0:
            ops.grad_clip_coef(self._arena.grads, self.state, 1.0 / self.world, self.clip_grad_value,
                               self.clip_max_norm)

    # -- checkpoint state (torch.optim layout) ------------------------------------------------------------------
    def _moment_names(self):
        raise NotImplementedError

    def _hyper(self):
        raise NotImplementedError

    def state_dict(self):
        """torch.optim-compatible: {'state': {index: {...per-parameter tensors}}, 'param_groups': [...]} with the
        parameters numbered group by group in named_parameters() order (what build_optimizer hands torch.optim)."""
        arena = self._arena
        step = float(self.state[0])
        state, groups, idx = {}, [], 0
        for gi, grp in enumerate(self.param_groups):
            ids = []
            for n in grp['param_names']:
                o, p = arena.offsets[n], arena.module_params[n]
                entry = {k: arena._view(buf[o:o + p.numel()], p.shape).clone(memory_format=torch.contiguous_format)
                         for k, buf in self._moment_names().items()}
                entry.update(self._per_param_extra(step))
                if step > 0:
                    state[idx] = entry
                ids.append(idx)
                idx += 1
            g = {'lr': grp['lr'], 'weight_decay': grp['weight_decay'], 'initial_lr': grp['initial_lr'], 'params': ids}
            g.update(self._hyper())
            groups.append(g)
        return {'state': state, 'param_groups': groups,
                'sacv': {'iter_index': self.iter_index, 'micro': self.micro, 'steps': step}}

    def load_state_dict(self, sd):
        """Accepts this class's state_dict() and a torch.optim state_dict of the reference optimizer (its two groups
        may come in either order: they are recognised by their size)."""
        arena = self._arena
        groups = sd['param_groups']
        if len(groups) != 2:
            raise ValueError(f'expected the two parameter groups of build_optimizer, got {len(groups)}')
        mine = {len(g['param_names']): g for g in self.param_groups}
        if len(mine) != 2 or sorted(len(g['params']) for g in groups) != sorted(mine):
            raise ValueError('parameter groups do not match this model (sizes '
                             f"{[len(g['params']) for g in groups]} vs {[len(g['param_names']) for g in self.param_groups]})")
        steps, loaded = set(), False
        for g in groups:
            tgt = mine[len(g['params'])]
            tgt['lr'] = g['lr']
            tgt['weight_decay'] = g['weight_decay']
            tgt['initial_lr'] = g.get('initial_lr', tgt['initial_lr'])
            for pid, n in zip(g['params'], tgt['param_names']):
                st = sd['state'].get(pid)
                if st is None:
                    continue
                o, p = arena.offsets[n], arena.module_params[n]
                for k, buf in self._moment_names().items():
                    if k in st and st[k] is not None:
                        arena._view(buf[o:o + p.numel()], p.shape).copy_(st[k].to(buf.device, torch.float32))
                        loaded = True
                if 'step' in st:
                    steps.add(float(st['step']))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): not a checkpoint of this optimizer')
        # torch.optim.SGD keeps no per-parameter step count: a checkpoint of it that carries momentum buffers has taken
        # at least one step (the count only decides whether state_dict() lists the buffers; SGD has no bias correction)
        step = steps.pop() if steps else float(sd.get('sacv', {}).get('steps', 1.0 if loaded else 0.0))
        self.state.zero_()
        self.state[0] = step
        extra = sd.get('sacv', {})
        self.iter_index = int(extra.get('iter_index', 1))
        self.micro = 0

    def _per_param_extra(self, step):
        return {}


class DetectionTrainer(_StepDriver):

    def __init__(self, model, criterion, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3,
                 accumulation_steps=1, process_group=None, clip_grad_value=0.0, clip_max_norm=0.0,
                 skip_inf_nan_grad=False, transport='torch'):
        self.criterion = criterion
        self.betas, self.eps = betas, eps
        self._init_driver(model, lr, weight_decay, accumulation_steps, clip_grad_value, clip_max_norm,
                          skip_inf_nan_grad, process_group, transport)
        self.exp_avg = torch.zeros_like(self._arena.params)
        self.exp_avg_sq = torch.zeros_like(self._arena.params)

    def _moment_names(self):
        return {'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}

    def _per_param_extra(self, step):
        return {'step': torch.tensor(step, dtype=torch.float32)}

    def _hyper(self):
        return {'betas': tuple(self.betas), 'eps': self.eps, 'amsgrad': False, 'maximize': False, 'foreach': None,
                'capturable': False, 'differentiable': False, 'fused': None, 'decoupled_weight_decay': True}

    def step(self, images, annots):
        """One iteration on a device-resident batch.  Returns the loss terms (divided by accumulation_steps, like
        the reference's `loss_value`) as 0-d device tensors (no host synchronisation when accumulation_steps == 1)."""
        model, crit = self.model, self.criterion
        self._begin(images, annots)
        engine, arena = model.engine, self._arena
        outs = engine.forward(images, training=True, keep=True)
        fused = crit.fused(list(outs), annots, want_grad=True)
        values, grads = fused[0], fused[2:]
        ops.loss_skip_flag(values, self.flag)
        self._vote_begin()
        inv = 1.0 / self.accumulation_steps
        engine.backward(*grads, grad_scale=inv, on_stage=self._stage_cb())
        names = ('cls_loss', 'reg_loss', 'center_ness_loss')[:values.numel()]
        if inv != 1.0:
            values = ops.eltwise(ops.EW_SCALE, values, scalar=inv)
        out = {k: values[i] for i, k in enumerate(names)}
        if self._after_backward():
            return out
        if self.boundary:
            self._gate_and_clip(self.betas[0], self.betas[1])
            g1, g2 = self.param_groups
            ops.adamw_step_gated(arena.params, arena.grads, self.exp_avg, self.exp_avg_sq, arena.split, g1['lr'],
                                 self.betas[0], self.betas[1], self.eps, g1['weight_decay'], g2['weight_decay'],
                                 self.state, grad_scale=1.0 / self.world, clip_value=self.clip_grad_value,
                                 lr_hi=g2['lr'])
            if self.ema_model is not None:
                self.ema_model.update(self.model, gate=self.state)
            self.micro = 0
        self.iter_index += 1
        return out

    def finish_step(self, loss_dict):
        """Loss terms averaged over ranks as python floats (the caller's logging values): one tiny
        all-reduce and ONE host synchronisation."""
        keys = list(loss_dict)
        v = torch.cat([torch.stack([loss_dict[k] for k in keys]).float(), self.state[0:2]])
        v = vote_and_reduce_scalars(v, self.group, comm=self.comm).tolist()
        out = {k: x / self.world for k, x in zip(keys, v)}
        out['loss'] = sum(out.values())
        out['optimizer_steps'] = int(round(v[-2] / self.world))
        out['skipped'] = v[-1] > 0
        return out


class MAETrainer(DetectionTrainer):
    """One iteration of tools/scripts.py::train_mae_self_supervised_learning (:1748-1900) with the recipe's optimizer
    (2.masked_image_modeling_training/imagenet/mae_vit_base_patch16_224/train_config.py: AdamW lr 6e-4, betas
    (0.9, 0.95), weight decay 5e-2 on ndim>1 tensors, CosineLR with 40 warm-up epochs, use_amp).  The labels of the
    reference come from its collater (per-patch normalised pixels, masked_image_modeling/common.py:8-44); when the
    caller passes none they are computed from the images on the device."""

    def __init__(self, model, criterion, lr=6e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=5e-2, norm_label=True,
                 **kw):
        super().__init__(model, criterion, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self.norm_label = norm_label

    def step(self, images, labels=None, noise=None):
        from . import vit_ops as V
        model = self.model
        self._begin(*([images] if labels is None else [images, labels]))
        engine, arena = model.engine, self._arena
        images = images.float()
        if labels is None:
            labels = V.patch_labels(images, model.patch_size, norm=self.norm_label)
        pred, mask = engine.forward(images, noise=noise, keep=True)
        inv = 1.0 / self.accumulation_steps
        loss, dpred = self.criterion.fused(pred, labels.contiguous().float(), mask, grad_scale=inv)
        ops.loss_skip_flag(loss, self.flag)
        self._vote_begin()
        engine.backward(dpred, on_stage=self._stage_cb())
        if inv != 1.0:
            loss = ops.eltwise(ops.EW_SCALE, loss, scalar=inv)
        out = {'loss': loss[0]}
        if self._after_backward():
            return out
        if self.boundary:
            self._gate_and_clip(self.betas[0], self.betas[1])
            g1, g2 = self.param_groups
            ops.adamw_step_gated(arena.params, arena.grads, self.exp_avg, self.exp_avg_sq, arena.split, g1['lr'],
                                 self.betas[0], self.betas[1], self.eps, g1['weight_decay'], g2['weight_decay'],
                                 self.state, grad_scale=1.0 / self.world, clip_value=self.clip_grad_value,
                                 lr_hi=g2['lr'])
            if self.ema_model is not None:
                self.ema_model.update(self.model, gate=self.state)
            self.micro = 0
        self.iter_index += 1
        return out

    def finish_step(self, loss_dict):
        v = torch.cat([loss_dict['loss'].reshape(1).float(), self.state[0:2]])
        v = vote_and_reduce_scalars(v, self.group, comm=self.comm).tolist()
        return {'loss': v[0] / self.world, 'optimizer_steps': int(round(v[1] / self.world)), 'skipped': v[2] > 0}


class ClassificationTrainer(_StepDriver):
    """One ImageNet training iteration of tools/scripts.py::train_classification with the ResNet50 recipe's
    optimizer (0.classification_training/imagenet/resnet50/train_config.py:68-79: SGD lr 0.1, momentum 0.9,
    weight decay 1e-4 on ndim>1 tensors only) as one fused kernel over the flat arena."""

    def __init__(self, model, criterion, lr=0.1, momentum=0.9, weight_decay=1e-4, accumulation_steps=1,
                 process_group=None, nesterov=False, clip_grad_value=0.0, clip_max_norm=0.0, skip_inf_nan_grad=False,
                 transport='torch', use_amp=None):
        if use_amp is not None:  # config.use_amp: bf16 matrix-core path (no GradScaler: bf16 has fp32's exponent)
            model.set_amp(use_amp)
        self.criterion = criterion
        self.momentum, self.nesterov = momentum, nesterov
        self._init_driver(model, lr, weight_decay, accumulation_steps, clip_grad_value, clip_max_norm,
                          skip_inf_nan_grad, process_group, transport)
        self.momentum_buf = torch.zeros_like(self._arena.params)

    def _moment_names(self):
        return {'momentum_buffer': self.momentum_buf}

    def _hyper(self):
        return {'momentum': self.momentum, 'dampening': 0, 'nesterov': self.nesterov, 'maximize': False,
                'foreach': None, 'differentiable': False, 'fused': None}

    def step(self, images, labels):
        model = self.model
        self._begin(images)
        engine, arena = model.engine, self._arena
        logits = engine.forward(images, training=True, keep=True)
        inv = 1.0 / self.accumulation_steps
        loss, dlogits = self.criterion.fused(logits, labels, grad_scale=inv)
        ops.loss_skip_flag(loss, self.flag)
        self._vote_begin()
        engine.backward(dlogits, on_stage=self._stage_cb())
        if inv != 1.0:
            loss = ops.eltwise(ops.EW_SCALE, loss, scalar=inv)
        out = {'loss': loss[0]}
        if self._after_backward():
            return out
        if self.boundary:
            self._gate_and_clip()
            g1, g2 = self.param_groups
            ops.sgd_step_gated(arena.params, arena.grads, self.momentum_buf, arena.split, g1['lr'], self.momentum,
                               g1['weight_decay'], g2['weight_decay'], self.state, grad_scale=1.0 / self.world,
                               clip_value=self.clip_grad_value, lr_hi=g2['lr'], nesterov=self.nesterov)
            if self.ema_model is not None:
                self.ema_model.update(self.model, gate=self.state)
            self.micro = 0
        self.iter_index += 1
        return out

    def finish_step(self, loss_dict):
        v = torch.cat([loss_dict['loss'].reshape(1).float(), self.state[0:2]])
        v = vote_and_reduce_scalars(v, self.group, comm=self.comm).tolist()
        return {'loss': v[0] / self.world, 'optimizer_steps': int(round(v[1] / self.world)), 'skipped': v[2] > 0}
