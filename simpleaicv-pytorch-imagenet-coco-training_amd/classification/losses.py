"""CELoss with the reference's contract (simpleAICV/classification/losses.py:14-26: nn.CrossEntropyLoss,
reduction='mean'), computed by one fused softmax / cross-entropy / gradient kernel."""
import torch
import torch.nn as nn

from .. import ops

__all__ = ['CELoss']


class _CEFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pred, label):
        loss, dl = ops.softmax_ce(pred.contiguous(), label, want_grad=True)
        ctx.dl = dl
        return loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        return ctx.dl * grad_out, None


class CELoss(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, pred, label):
        if pred.device.type != 'cuda':
            raise RuntimeError('sacv_amd CELoss runs on MI355X only (no CPU fallback)')
        if torch.is_grad_enabled() and pred.requires_grad:
            return _CEFunction.apply(pred.float(), label)
        return ops.softmax_ce(pred.float().contiguous(), label, want_grad=False)[0][0]

    def fused(self, pred, label, grad_scale=1.0):
        """(loss f32[1], dloss/dpred * grad_scale) in one pass -- the trainer's path (no autograd)."""
        return ops.softmax_ce(pred, label, want_grad=True, grad_scale=grad_scale)
