"""Epoch loops -- same positional signatures as css/trainer/iteration_types.py:17,67.

``validate`` (forward only) runs entirely on the HIP path.  ``train_one_epoch`` keeps the
reference's step order (forward, backward, clip, step, zero_grad, scheduler).  Forward and the
PIT backward are HIP kernels; the Conformer (d_model 256) backward is explicit HIP kernels too, other models use
the torch-autograd stop-gap of css_amd/autograd.py
(PyTorch-ROCm restatement used only to back-propagate) until the backward kernels exist.
In a multi-GPU job (torch.distributed initialised) gradients are averaged across ranks over RCCL
before clipping (css_amd/parallel.py: with FusedAdam, buckets of ~4 encoder blocks of the flat
gradient buffer are all-reduced while the remaining blocks are back-propagated), replacing the
reference's per-epoch checkpoint averaging (scripts/python/combine_models.py:55-82).
"""
import logging

import torch

from .. import ops, parallel


def train_one_epoch(args, generator, model, objective, optim, lr_sched, device="cpu"):
    total_loss = 0.0
    reducer = None
    if hasattr(optim, "flat_grad") and parallel.is_distributed():
        reducer = parallel.OverlappedGradReducer(model, optim, blocks_per_bucket=getattr(args, "blocks_per_bucket", 4))
    for i in range(1, args.batches_per_epoch + 1):
        b = next(generator)
        log = (f"Iter: {i} of {args.batches_per_epoch} LR:{lr_sched.curr_lr:0.5e} bsize: {b['mix'].size(0)} "
               f"window (# frames): {b['mix'].size(1)} ovl: {b['ovl']:0.4f} ")
        loss = objective(model, b, device=device)
        if not loss.requires_grad:
            raise RuntimeError("css_amd: loss has no grad_fn (model parameters frozen or grad disabled)")
        loss_val = loss.data.item()
        ops.lstm_dist_check()  # the item() above synchronised: a timed-out BLSTM hand-off invalidates this loss
        log += f"Loss: {loss_val:0.5f} "
        total_loss += loss_val
        if reducer is not None:
            reducer.begin()   # finished blocks are all-reduced while the rest of the backward pass runs
        loss.backward()
        del b
        if hasattr(optim, "flat_grad"):  # css_amd.optim.FusedAdam: the flat gradient buffer is the all-reduce bucket
            if reducer is not None:
                reducer.finish()
            grad_norm = optim.step(max_norm=args.grad_thresh)  # clip + Adam in one pass, norm stays on the device
        else:
            parallel.allreduce_gradients(model.parameters())
            grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), args.grad_thresh)
            optim.step()
        log += f"Grad_norm: {grad_norm.data.item():0.5f}"
        ops.lstm_dist_check()  # ... and the backward's hand-offs before the update is logged as good
        logging.info(log)
        optim.zero_grad()
        lr_sched.step(1.0)
    return total_loss / args.batches_per_epoch


def validate(generator, model, objective, device="cpu", num_batches=100):
    """100 dev batches, mean loss (iteration_types.py:67-77); the mean stays a tensor like the reference's."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        avg_loss = 0.0
        for _ in range(num_batches):
            b = next(generator)
            avg_loss = avg_loss + objective(model, b, device=device)
        avg_loss = avg_loss / num_batches
    ops.lstm_dist_check()
    if was_training:
        model.train()
    return avg_loss
