"""Autograd glue for training.

Forward values ALWAYS come from the HIP kernels.  Backward:
  * PIT-MSE -> masks: hand-written kernel (csship_pit_mse_bwd);
  * masks -> parameters: STOP-GAP.  The encoder backward kernels are not written yet, so the gradient is
    obtained by re-evaluating a differentiable PyTorch-ROCm restatement of the model ("twin", GPU torch ops,
    fp32) and back-propagating grad_masks through it (SURVEY.md 7, step 5 allows exactly this stop-gap).
    The twin never produces a forward result that is returned to the caller.
"""
import torch

from . import ops


class PitMse(torch.autograd.Function):
    """loss = mean_b min_perm MSE(mask_k * f_orig, target_{perm})  with the hand-written backward."""

    @staticmethod
    def forward(ctx, masks, f_orig, n_masks, *targets):
        loss, perm, _, _ = ops.pit_mse(masks, f_orig, list(targets))
        ctx.save_for_backward(masks, f_orig, perm, *targets)
        ctx.n_masks = n_masks
        ctx.mark_non_differentiable(perm)
        return loss, perm

    @staticmethod
    def backward(ctx, grad_loss, _grad_perm):
        masks, f_orig, perm, *targets = ctx.saved_tensors
        g = ops.pit_mse_bwd(masks, f_orig, targets, perm, ctx.n_masks, 1.0)
        return (g * grad_loss,) + (None,) * (2 + len(targets))


class MasksWithTwinBackward(torch.autograd.Function):
    """masks = model.forward_masks(f) on the HIP path; d masks / d params through the model's torch twin."""

    @staticmethod
    def forward(ctx, model, f, *params):
        ctx.model = model
        ctx.save_for_backward(f)
        ctx.n_params = len(params)
        return model._forward_masks_hip(f)

    @staticmethod
    def backward(ctx, grad_masks):
        (f,) = ctx.saved_tensors
        model = ctx.model
        params = [p for p in model.parameters()]
        with torch.enable_grad():
            out = model._twin_forward_masks(f.detach())
            need = [p for p in params if p.requires_grad]
            grads = torch.autograd.grad(out, need, grad_masks.contiguous(), allow_unused=True)
        it = iter(grads)
        full = tuple(next(it) if p.requires_grad else None for p in params)
        return (None, None) + full


def masks_for_training(model, f):
    params = tuple(model.parameters())
    return MasksWithTwinBackward.apply(model, f, *params)
