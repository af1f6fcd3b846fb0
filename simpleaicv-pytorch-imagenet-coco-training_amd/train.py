"""Training-step driver: the build's restatement of tools/scripts.py::train_detection (:852-1050)
for one step, with the optimizer of tools/utils.py:480-582 (AdamW, lr 1e-4, wd 1e-3 on ndim>1
tensors, 0 on 1-D tensors) as ONE fused kernel over the flat parameter arena.

Per step, in the reference's order: H2D-resident batch -> inf/nan input guard -> forward -> loss
dict -> sum -> inf/nan/zero-loss guard -> backward (gradient all-reduce, mean over ranks) ->
cross-rank skip vote -> AdamW -> loss terms averaged over ranks.  The guards are evaluated on
the device and voted on together with the loss terms in ONE small all-reduce, so a step costs
one host synchronisation instead of the reference's ten.
"""
import torch
import torch.distributed as dist

from . import ops


class DetectionTrainer:

    def __init__(self, model, criterion, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3,
                 accumulation_steps=1, process_group=None, bucket_mb=32):
        self.model, self.criterion = model, criterion
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.accumulation_steps = accumulation_steps
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        arena = model.arena
        self.exp_avg = torch.zeros_like(arena.params)
        self.exp_avg_sq = torch.zeros_like(arena.params)
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self.comm_stream = torch.cuda.Stream() if self.world > 1 else None

    def _allreduce_grads(self):
        """Mean of the flat gradient arena over ranks: RCCL all-reduce of zero-copy arena slices."""
        arena = self.model.arena
        g = arena.grads
        n = g.numel()
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            for s in range(0, n, self.bucket_elems):
                dist.all_reduce(g[s:min(n, s + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group)
        torch.cuda.current_stream().wait_stream(self.comm_stream)

    def step(self, images, annots):
        model, crit = self.model, self.criterion
        engine = model.engine
        arena = model.arena
        cls, reg = engine.forward(images, training=True, keep=True)
        values, counts, dcls, dreg = crit.fused([cls, reg], annots, want_grad=True)
        engine.backward(dcls, dreg, grad_scale=1.0 / self.accumulation_steps)
        if self.world > 1:
            self._allreduce_grads()
        self.step_count += 1
        ops.adamw_step(arena.params, arena.grads, self.exp_avg, self.exp_avg_sq, arena.split, self.lr,
                       self.betas[0], self.betas[1], self.eps, 0.0, self.weight_decay, self.step_count,
                       grad_scale=1.0 / self.world)
        return {'cls_loss': values[0], 'reg_loss': values[1]}
