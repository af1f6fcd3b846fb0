"""Training-step driver: the build's restatement of tools/scripts.py::train_detection (:852-1050)
for one step, with the optimizer of tools/utils.py:480-582 (AdamW, lr 1e-4, wd 1e-3 on ndim>1
tensors, 0 on 1-D tensors) as ONE fused kernel over the flat parameter arena.

Per step, in the reference's order: H2D-resident batch -> forward (rank-0 BN buffers broadcast
first, as DDP does) -> loss dict -> backward with the gradient all-reduce of each finished stage
overlapped under the remaining backward kernels -> AdamW (mean over ranks folded in) -> loss
terms.  The reference's zero / inf / nan loss guard and skip vote (:918-930, :963-974) run on the device:
`loss_skip_flag` -> one-float all-reduce -> `step_gate` -> the gated optimizer kernel, which leaves parameters,
moments and the step count untouched when any rank votes to skip; the loss averaging (:1017-1029) travels in one
small all-reduce in `finish_step`, so a step costs one host synchronisation instead of the reference's ten.
"""
import torch
import torch.distributed as dist

from . import ops
from .parallel import GradExchanger, stage_ranges, broadcast_buffers, vote_and_reduce_scalars


class DetectionTrainer:

    def __init__(self, model, criterion, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3,
                 accumulation_steps=1, process_group=None):
        self.model, self.criterion = model, criterion
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        if accumulation_steps != 1:
            # engine.backward overwrites the gradient arena; accumulating micro-batches is not implemented
            raise NotImplementedError('accumulation_steps > 1 (the reference recipes on this path use 1)')
        self.accumulation_steps = accumulation_steps
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        arena = model.arena
        # device-side optimizer state {steps taken, skip flag of the last step, bias corrections}: a batch whose
        # loss is 0 / inf / nan on any rank is skipped by all ranks (tools/scripts.py:918-974) without a host sync
        self.state = torch.zeros(4, device=arena.params.device, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(arena.params)
        self.exp_avg_sq = torch.zeros_like(arena.params)
        self.exchanger = GradExchanger(arena.grads, stage_ranges(arena.order, arena.offsets, arena.numel, arena.split),
                                       process_group)
        if self.world > 1:  # DDP init_sync: every rank starts from rank 0's parameters and buffers
            dist.broadcast(arena.params, src=0, group=process_group)
            broadcast_buffers(arena.buffers, process_group)

    def step(self, images, annots):
        """One optimizer step on a device-resident batch.  Returns the loss terms as 0-d device tensors
        (no host synchronisation here)."""
        model, crit = self.model, self.criterion
        engine, arena = model.engine, model.arena
        if self.world > 1:
            broadcast_buffers(arena.buffers, self.group)
        outs = engine.forward(images, training=True, keep=True)
        fused = crit.fused(list(outs), annots, want_grad=True)
        values, counts, grads = fused[0], fused[1], fused[2:]
        flag = ops.loss_skip_flag(values)
        vote = dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        engine.backward(*grads, grad_scale=1.0 / self.accumulation_steps, on_stage=self.exchanger.stage_done)
        if vote is not None:
            vote.wait()
        ops.step_gate(flag, self.state, self.betas[0], self.betas[1])
        self.exchanger.wait()
        ops.adamw_step_gated(arena.params, arena.grads, self.exp_avg, self.exp_avg_sq, arena.split, self.lr,
                             self.betas[0], self.betas[1], self.eps, 0.0, self.weight_decay, self.state,
                             grad_scale=1.0 / self.world)
        names = ('cls_loss', 'reg_loss', 'center_ness_loss')[:values.numel()]
        return {k: values[i] for i, k in enumerate(names)}

    def finish_step(self, loss_dict):
        """Loss terms averaged over ranks as python floats (the caller's logging values): one tiny
        all-reduce and ONE host synchronisation."""
        keys = list(loss_dict)
        v = torch.cat([torch.stack([loss_dict[k] for k in keys]).float(), self.state[0:2]])
        v = vote_and_reduce_scalars(v, self.group).tolist()
        out = {k: x / self.world for k, x in zip(keys, v)}
        out['loss'] = sum(out.values())
        out['optimizer_steps'] = int(round(v[-2] / self.world))
        out['skipped'] = v[-1] > 0
        return out


class ClassificationTrainer:
    """One ImageNet training step of tools/scripts.py::train_classification with the ResNet50 recipe's
    optimizer (0.classification_training/imagenet/resnet50/train_config.py:68-79: SGD lr 0.1, momentum 0.9,
    weight decay 1e-4 on ndim>1 tensors only) as one fused kernel over the flat arena."""

    def __init__(self, model, criterion, lr=0.1, momentum=0.9, weight_decay=1e-4, accumulation_steps=1,
                 process_group=None):
        self.model, self.criterion = model, criterion
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        if accumulation_steps != 1:
            # engine.backward overwrites the gradient arena; accumulating micro-batches is not implemented
            raise NotImplementedError('accumulation_steps > 1 (the reference recipes on this path use 1)')
        self.accumulation_steps = accumulation_steps
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        arena = model.arena
        self.state = torch.zeros(4, device=arena.params.device, dtype=torch.float32)  # see DetectionTrainer
        self.momentum_buf = torch.zeros_like(arena.params)
        self.exchanger = GradExchanger(arena.grads, stage_ranges(arena.order, arena.offsets, arena.numel, arena.split),
                                       process_group)
        if self.world > 1:
            dist.broadcast(arena.params, src=0, group=process_group)
            broadcast_buffers(arena.buffers, process_group)

    def step(self, images, labels):
        model = self.model
        engine, arena = model.engine, model.arena
        if self.world > 1:
            broadcast_buffers(arena.buffers, self.group)
        logits = engine.forward(images, training=True, keep=True)
        loss, dlogits = self.criterion.fused(logits, labels, grad_scale=1.0 / self.accumulation_steps)
        flag = ops.loss_skip_flag(loss)
        vote = dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        engine.backward(dlogits, on_stage=self.exchanger.stage_done)
        if vote is not None:
            vote.wait()
        ops.step_gate(flag, self.state)
        self.exchanger.wait()
        ops.sgd_step_gated(arena.params, arena.grads, self.momentum_buf, arena.split, self.lr, self.momentum, 0.0,
                           self.weight_decay, self.state, grad_scale=1.0 / self.world)
        return {'loss': loss[0]}

    def finish_step(self, loss_dict):
        v = torch.cat([loss_dict['loss'].reshape(1).float(), self.state[0:2]])
        v = vote_and_reduce_scalars(v, self.group).tolist()
        return {'loss': v[0] / self.world, 'optimizer_steps': int(round(v[1] / self.world)), 'skipped': v[2] > 0}
