"""Epoch loops -- same positional signatures as css/trainer/iteration_types.py:17,67.

``validate`` (forward only) runs entirely on the HIP path.  ``train_one_epoch`` keeps the
reference's step order (forward, backward, clip, step, zero_grad, scheduler) and its per-iteration log line; forward,
PIT and the whole backward are HIP kernels (css_amd/autograd.py).  With ``css_amd.optim.FusedAdam`` a batch
shape that repeats is replayed as a captured HIP graph (``GraphedTrainStep``: 14.2 ms against 19.8 ms launch by launch for
Conformer-base at 32 x 400; ``args.graph = False`` switches it off), bit-identical to the eager step.
In a multi-GPU job (torch.distributed initialised) gradients are averaged across ranks over RCCL
before clipping (css_amd/parallel.py: with FusedAdam, buckets of ~4 encoder blocks of the flat
gradient buffer are all-reduced while the remaining blocks are back-propagated), replacing the
reference's per-epoch checkpoint averaging (scripts/python/combine_models.py:55-82).
"""
import logging
from collections import OrderedDict

import torch

from .. import ops, parallel
from ..objectives.mse import model_device


def _graphed_step(args, model, objective, optim, b, device):
    """The captured step for this batch shape, or None: no FusedAdam, switched off, SyncBatchNorm, capture failed - or a
    shape met for the first time.  The reference draws the window length per batch (separation.py:94-95,175), so the
    shape changes from batch to batch: a shape is captured the SECOND time it is met (``args.graph_after``), at most
    ``args.graph_cache`` (default 4) captured steps are kept, least recently used first out (each owns a private
    memory pool holding a whole step's activations), everything else runs the eager step.  Ranks of a distributed job
    decide independently: both forms issue the same collectives (parallel.bucket_plan)."""
    if not hasattr(optim, "flat_grad") or not getattr(args, "graph", True) or str(device) == "cpu" or \
            getattr(model, "sync_batchnorm", False) or not model.training:
        return None
    state = optim.__dict__.setdefault("_graphed_steps", {"cache": OrderedDict(), "seen": {}})
    cache, seen = state["cache"], state["seen"]
    key = (id(model), id(objective), float(args.grad_thresh)) + tuple((k, tuple(b[k].shape)) for k in ("mix", "source1", "source2") if k in b)
    if key in cache:
        cache.move_to_end(key)
        return cache[key]
    seen[key] = seen.get(key, 0) + 1
    if len(seen) > 4096:   # bounded bookkeeping
        seen.clear()
    if seen[key] < int(getattr(args, "graph_after", 2)):
        return None
    from .graphed import GraphedTrainStep
    try:
        step = GraphedTrainStep(model, objective, optim, b, max_norm=args.grad_thresh, device=device,
                                buckets=getattr(args, "buckets", 4), blocks_per_bucket=getattr(args, "blocks_per_bucket", 4))
    except (RuntimeError, TypeError, NotImplementedError) as e:   # e.g. a model whose step syncs with the host
        logging.warning("css_amd: training step not captured (%s); launching it kernel by kernel", e)
        step = None
    cache[key] = step
    while len(cache) > max(1, int(getattr(args, "graph_cache", 4))):
        old_key, old = cache.popitem(last=False)
        seen.pop(old_key, None)   # an evicted shape starts over: it must repeat again before it is re-captured
        if old is not None:
            old.release()
    return step


def train_one_epoch(args, generator, model, objective, optim, lr_sched, device="cpu"):
    device = model_device(model, device)   # the reference's default "cpu" = "where the model is" (objectives/mse.py)
    total_loss = 0.0
    reducer = None
    for i in range(1, args.batches_per_epoch + 1):
        b = next(generator)
        log = (f"Iter: {i} of {args.batches_per_epoch} LR:{lr_sched.curr_lr:0.5e} bsize: {b['mix'].size(0)} "
               f"window (# frames): {b['mix'].size(1)} ovl: {b['ovl']:0.4f} ")
        step = _graphed_step(args, model, objective, optim, b, device)
        if step is not None:   # zero_grad, forward, backward, (all-reduce,) clip + Adam: one replay
            loss, grad_norm = step(b)
            loss_val = loss.item()
            ops.lstm_dist_check()
            total_loss += loss_val
            logging.info(log + f"Loss: {loss_val:0.5f} Grad_norm: {grad_norm.item():0.5f}")
            lr_sched.step(1.0)
            continue
        if reducer is None and hasattr(optim, "flat_grad") and parallel.is_distributed():
            reducer = parallel.OverlappedGradReducer(model, optim, blocks_per_bucket=getattr(args, "blocks_per_bucket", 4))
        if hasattr(optim, "flat_grad"):
            # a replayed step leaves its (averaged) gradients in the flat buffer and the explicit backward ACCUMULATES
            # into the .grad views: an eager step that follows a replay starts from a zeroed buffer, like the replay does
            optim.zero_grad()
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
    device = model_device(model, device)
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
