"""One-process-per-GPU data parallelism over torch.distributed (backend "nccl" = RCCL on ROCm).

The reference has no collective at all: "multi-node" training launches independent jobs and
averages their checkpoints on disk once per epoch (scripts/bash/train_parallel.sh:56-87,
scripts/python/combine_models.py:55-82).  Here every step sums ONE flat gradient bucket across
ranks -- a single large all-reduce is what the point-to-point xGMI links want (21.6 M params =
86 MB fp32 for Conformer-base) -- and long-recording inference shards window indices across
ranks with no collective on the data path.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(n_items, rank, world_size):
    """Contiguous [lo, hi) slice of n_items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_gradients(params, average=True):
    """Sum (or average) .grad of all params across ranks through one flat bucket.  No-op on 1 rank."""
    if not is_distributed():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allreduce_flat(flat, average=True):
    """The same on a gradient buffer that is already flat (css_amd.optim.FusedAdam.flat_grad): no copies."""
    if not is_distributed():
        return
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()


def average_parameters(models_state_dicts):
    """combine_models.py:55-82 equivalent: elementwise mean of state dicts (scalar buffers keep dtype)."""
    n = len(models_state_dicts)
    out = {}
    for k, v in models_state_dicts[0].items():
        acc = torch.zeros_like(v, dtype=torch.float64)
        for sd in models_state_dicts:
            acc += sd[k].double() / n
        out[k] = acc.to(v.dtype)
    return out


def gather_windows(local, n_total, rank, world_size):
    """All-gather per-window results (dim 0) sharded with shard_range back to every rank."""
    if not is_distributed():
        return local
    counts = [shard_range(n_total, r, world_size) for r in range(world_size)]
    mx = max(hi - lo for lo, hi in counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(counts)], 0)
