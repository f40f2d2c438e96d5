"""Permutation-invariant loss -- drop-in for css.objectives.pit.pit_loss (pit.py:5-62).

Only the case the reference ever uses is built: ``loss_fn = mse_loss`` over [K, T, F] tensors
with the speaker axis first.  The K! candidate losses are derived from one K x K pairwise
squared-error matrix computed by a single HIP kernel (css_amd/csrc/pit.hip).
"""
import torch

from .. import ops


def pit_loss(estimate, target, axis=0, loss_fn=torch.nn.functional.mse_loss, return_permutation=False, **kwargs):
    sources = estimate.size()[axis]
    assert sources < 30, (
        f"Are you sure? sources={sources}, estimate.shape={estimate.shape}, target.shape={target.shape}")  # pit.py:34-37
    assert estimate.size() == target.size(), f"{estimate.size()} != {target.size()}"  # pit.py:47
    if loss_fn is not torch.nn.functional.mse_loss or kwargs:
        raise NotImplementedError("css_amd.pit_loss implements loss_fn=mse_loss only (the reference's only use)")
    if estimate.dim() != 3:
        raise NotImplementedError("css_amd.pit_loss expects [K, T, F] (or any 3-D tensor with a speaker axis)")
    axis = axis % 3
    if axis != 0:
        estimate, target = estimate.movedim(axis, 0), target.movedim(axis, 0)
    K, T, F = estimate.shape
    if K > 4:
        raise NotImplementedError("css_amd.pit_loss is built for at most 4 sources")
    est = estimate.contiguous()
    tgt = target.contiguous()
    # estimate as "masks" with f_orig = 1: reuse the fused kernel on a [1, T, K*F] view
    masks = est.permute(1, 0, 2).reshape(1, T, K * F).contiguous()
    ones = torch.ones(1, T, F, device=est.device, dtype=torch.float32)
    loss, perm, _, _ = ops.pit_mse(masks, ones, [tgt[k].unsqueeze(0) for k in range(K)])
    if return_permutation:
        return loss, tuple(int(v) for v in perm[0].tolist())
    return loss
