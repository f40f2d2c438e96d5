"""PIT-MSE objective -- drop-in for css.objectives.mse.MeanSquaredError (mse.py:6-37).

``forward(model, sample, device)`` moves the batch to the device, runs the model and returns
the batch mean of the per-sample minimum-over-permutations MSE.  With a css_amd model the
mask application (conformer.py:103) and the loss are ONE fused kernel and ``y_pred`` is never
materialised.  The reference objective crashes on the ``(y_pred, masks)`` tuple its own models
return; like the oracle, this one takes ``model(x)[0]``.
"""
import torch

from .. import ops


def model_device(model, device):
    """``device`` unless it is the reference's default "cpu" and the model's parameters are on a GPU."""
    if str(device) == "cpu":
        p = next(iter(model.parameters()), None)
        if p is not None and p.is_cuda:
            return p.device
    return device


class MeanSquaredError(torch.nn.Module):
    @staticmethod
    def add_args(parser):
        pass

    @classmethod
    def build_objective(cls, conf):
        return MeanSquaredError()

    @classmethod
    def add_state_dict(cls, s1, s2, fraction, iteration=None):
        return s1

    def __init__(self):
        super().__init__()
        self.last_permutation = None

    def forward(self, model, sample, device="cpu"):
        """Same signature and default as mse.py:22.  There is no CPU path here: the reference's default ``"cpu"`` can only
        mean "where the model is", so with a model that lives on a GPU the batch follows the model (a caller that relies
        on the default, as train_one_epoch's own default does, gets the reference's behaviour, not a device mismatch)."""
        device = model_device(model, device)
        xs = sample["mix"].to(device)
        y1, y2 = sample["source1"].to(device), sample["source2"].to(device)
        if hasattr(model, "forward_masks"):  # css_amd model: fused mask-apply + PIT
            if xs.ndim == 4:
                xs = xs.squeeze(0)
            B, T, _ = xs.shape
            F = model.num_bins
            assert model.num_spk == 2, "MeanSquaredError hard-codes two sources (mse.py:27-28)"
            assert tuple(y1.shape) == (B, T, F), f"{(B, 2, T, F)} != {tuple(torch.stack((y1, y2), 1).shape)}"
            wants_grad = torch.is_grad_enabled() and (model.training or getattr(model, "eval_gradients", False)) and \
                any(p.requires_grad for p in model.parameters())
            if not wants_grad and hasattr(model, "forward_loss"):   # loss only: head + mask application + PIT in one kernel
                loss, perm, _ = model.forward_loss(xs, [y1, y2])
                self.last_permutation = perm
                return loss
            masks = model.forward_masks(xs)
            if masks.requires_grad:  # training: hand-written PIT backward w.r.t. the masks
                from ..autograd import PitMse
                loss, perm = PitMse.apply(masks, xs, model.num_spk + model.num_noise, y1, y2)
            else:
                loss, perm, _, _ = ops.pit_mse(masks, xs, [y1, y2])
        else:  # any other nn.Module returning (y_pred, masks) or y_pred
            out = model(xs)
            y_pred = out[0] if isinstance(out, (tuple, list)) else out
            B, N, T, F = y_pred.shape
            y_true_shape = (B, 2, T, F)
            assert tuple(y_pred.shape) == y_true_shape, f"{tuple(y_pred.shape)} != {y_true_shape}"
            ones = torch.ones(B, T, F, device=y_pred.device, dtype=torch.float32)
            flat = y_pred.permute(0, 2, 1, 3).reshape(B, T, N * F).contiguous()
            loss, perm, _, _ = ops.pit_mse(flat, ones, [y1, y2])
        self.last_permutation = perm
        return loss
