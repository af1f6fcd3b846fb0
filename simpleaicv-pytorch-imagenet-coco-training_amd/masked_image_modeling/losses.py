"""MSELoss with the reference's contract (simpleAICV/masked_image_modeling/losses.py:11-23: mean squared error per
patch, averaged over the REMOVED patches: (loss * mask).sum() / (mask.sum() + 1e-4)), computed by one fused kernel that
also yields d loss / d pred."""
import torch
import torch.nn as nn

from .. import vit_ops as V

__all__ = ['MSELoss']


class _MSEFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pred, rows, label, mask):
        loss, dpred = V.mse_loss(rows, label.contiguous().float(), mask.contiguous().float(), want_grad=True)
        ctx.dpred = dpred
        ctx.shape = pred.shape
        return loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        b, n, k = ctx.shape
        return (ctx.dpred.float().view(b, n + 1, k)[:, 1:, :] * grad_out).to(torch.bfloat16), None, None, None


class MSELoss(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, pred, label, mask):
        if pred.device.type != 'cuda':
            raise RuntimeError('sacv_amd MSELoss runs on MI355X only (no CPU fallback)')
        rows = getattr(pred, '_sacv_rows', None)
        if rows is None:
            raise RuntimeError('sacv_amd MSELoss takes the prediction tensor returned by the sacv_amd MAE model')
        if torch.is_grad_enabled() and pred.requires_grad:
            return _MSEFunction.apply(pred, rows, label, mask)
        return V.mse_loss(rows, label.contiguous().float(), mask.contiguous().float(), want_grad=False)[0][0]

    def fused(self, pred_rows, label, mask, grad_scale=1.0):
        """(loss f32[1], d loss / d pred_rows * grad_scale bf16) -- the trainer's path; pred_rows includes the cls row."""
        return V.mse_loss(pred_rows, label, mask, want_grad=True, grad_scale=grad_scale)
