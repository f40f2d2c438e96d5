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


class GradBucket:
    """ONE flat gradient buffer for a parameter list: every ``p.grad`` becomes a view of it (as css_amd.optim.FusedAdam
    does for its own flat buffers), so the all-reduce needs no gather / scatter copies.  For optimisers other than
    FusedAdam (which already owns such a buffer: pass its ``flat_grad`` to ``allreduce_flat``)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket got no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("GradBucket needs one device and one dtype")
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.flat[o:o + p.numel()].view(p.shape)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view

    def zero_(self):
        self.flat.zero_()

    def allreduce(self, average=True, buckets=1):
        allreduce_flat(self.flat, average=average, buckets=buckets)


_BUCKETS = {}


def allreduce_gradients(params, average=True):
    """Sum (or average) .grad of all params across ranks.  No-op on 1 rank.  The first call re-homes the gradients as
    views of one flat buffer (GradBucket); every later call is ONE collective on that buffer.  An optimiser whose
    ``zero_grad()`` drops the gradients (torch's default ``set_to_none=True``) makes autograd allocate fresh tensors
    every step: those are copied INTO the existing flat buffer and re-pointed at it (one copy per step, no 86 MB
    reallocation) - call ``zero_grad(set_to_none=False)`` to avoid even that."""
    if not is_distributed():
        return
    params = [p for p in params if p.requires_grad]
    key = tuple(id(p) for p in params)
    b = _BUCKETS.get(key)
    if b is None:
        mine = set(key)   # identities: `q in params` would compare tensors element-wise
        for k in [k for k, v in _BUCKETS.items() if any(q.grad is None and id(q) not in mine for q in v.params)]:
            del _BUCKETS[k]   # buckets of models that are gone
        b = _BUCKETS[key] = GradBucket(params)
    else:
        with torch.no_grad():
            for p, o in zip(b.params, _offsets(b.params)):
                view = b.flat[o:o + p.numel()].view(p.shape)
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view
    b.allreduce(average=average)


def _offsets(params):
    n = 0
    for p in params:
        yield n
        n += (p.numel() + 3) // 4 * 4


def allreduce_flat(flat, average=True, buckets=1, async_handles=None):
    """The same on a gradient buffer that is already flat (css_amd.optim.FusedAdam.flat_grad): no copies.

    ``buckets`` > 1 issues that many asynchronous all-reduces on contiguous chunks back to back and waits for all of
    them: the chunks pipeline through the ring (xGMI is point to point: a ring step is bound by one link, so several
    medium messages in flight keep all links busy while the first ones drain).  With ``async_handles`` (a list) the
    work handles are appended instead of waited for - the caller overlaps them with whatever it runs next and calls
    ``wait_all(handles, flat, average)``."""
    if not is_distributed():
        return
    world = dist.get_world_size()
    avg_op = average and dist.get_backend() == "nccl"   # RCCL averages in the collective; gloo has no AVG
    op = dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM
    n = flat.numel()
    if flat.is_cuda and dist.get_backend() == "gloo":
        # rehearsal backend (no RCCL): staged through host memory, synchronous
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host / world if average else host)
        return
    buckets = max(1, min(int(buckets), n))
    step = ((n + buckets - 1) // buckets + 3) // 4 * 4
    handles = [dist.all_reduce(flat[lo:min(n, lo + step)], op=op, async_op=True) for lo in range(0, n, step)]
    if async_handles is not None:
        async_handles.extend(handles)
        return
    for h in handles:
        h.wait()
    if average and not avg_op:
        flat /= world


def allreduce_sum_(t):
    """In-place SUM all-reduce of a small tensor inside a step (SyncBatchNorm statistics)."""
    if not is_distributed():
        return t
    if t.is_cuda and dist.get_backend() == "gloo":  # rehearsal backend: staged through host memory
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def convert_sync_batchnorm(model):
    """css_amd's form of nn.SyncBatchNorm.convert_sync_batchnorm (reference train.py:70): the BatchNorm modules stay
    what they are (parameter containers); the training forward / backward of the conv module all-reduce the batch
    statistics across ranks (ops.convmod(sync=True)).  Needs the eager step: collectives are not captured into the
    GraphedTrainStep graphs."""
    model.sync_batchnorm = True
    return model


def allreduce_segment(flat, lo, hi, handles, average=True):
    """Asynchronous all-reduce of flat[lo:hi] (a bucket whose gradients are final while the backward pass is still
    producing the rest).  Averaging for backends without AVG happens in ``wait_all``."""
    if not is_distributed() or hi <= lo:
        return
    if flat.is_cuda and dist.get_backend() == "gloo":  # rehearsal backend: staged through host memory, synchronous
        host = flat[lo:hi].cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat[lo:hi].copy_(host)
        return
    avg_op = average and dist.get_backend() == "nccl"
    handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM, async_op=True))


def wait_all(handles, flat=None, average=True):
    for h in handles:
        h.wait()
    handles.clear()
    if flat is not None and average and is_distributed() and dist.get_backend() != "nccl":
        flat /= dist.get_world_size()


class OverlappedGradReducer:
    """Bucketed gradient all-reduce overlapped with the backward pass (replaces per-epoch checkpoint averaging,
    scripts/python/combine_models.py:55-82).

    css_amd.optim.FusedAdam lays the parameters out in ``model.parameters()`` order in ONE flat gradient buffer, and the
    explicit backward (css_amd/autograd.py ConformerTrain) finishes the mask head first, then encoder blocks L-1 .. 0,
    then the embedding / position table.  Each finished module reports here; every ``blocks_per_bucket`` encoder blocks
    the stretch of the flat buffer they cover is all-reduced asynchronously (RCCL runs it on its own stream behind an
    event on the compute stream), while the remaining blocks are still being back-propagated.  ``finish()`` reduces
    what is left (embedding, position table) and waits for everything.  One process per GPU; xGMI is point to point,
    so a handful of ~20 MB messages in flight suits it better than one 86 MB message or hundreds of small ones."""

    def __init__(self, model, optim, blocks_per_bucket=4, average=True, emit=None):
        self.model, self.optim, self.every, self.average = model, optim, max(1, int(blocks_per_bucket)), average
        self.span = {}  # id(parameter) -> (lo, hi) in the flat buffer
        for p, o in zip(optim.param_groups[0]["params"], optim._offs):
            self.span[id(p)] = (o, o + (p.numel() + 3) // 4 * 4)
        self.handles, self.pending, self.low = [], [], optim.flat_grad.numel()
        # emit(lo, hi): what happens when a stretch is final.  Default: start its all-reduce.  bucket_plan() records
        # the stretches instead, the graphed step closes one captured sub-graph per stretch (trainer/graphed.py).
        self.emit = emit if emit is not None else \
            (lambda lo, hi: allreduce_segment(self.optim.flat_grad, lo, hi, self.handles, self.average))

    def _span_of(self, module):
        spans = [self.span[id(p)] for p in module.parameters() if id(p) in self.span]
        return (min(lo for lo, _ in spans), max(hi for _, hi in spans)) if spans else None

    def begin(self, force=False):
        self.handles, self.pending, self.low = [], [], self.optim.flat_grad.numel()
        self.model._grad_bucket_hook = self._on_module if (force or is_distributed()) else None

    def _flush(self):
        if not self.pending:
            return
        lo, hi = min(sp[0] for sp in self.pending), max(sp[1] for sp in self.pending)
        self.pending = []
        hi = min(hi, self.low)          # never send a stretch twice
        if lo < hi:
            self.emit(lo, hi)
            self.low = lo

    def _on_module(self, module):
        sp = self._span_of(module)
        if sp is None:
            return
        self.pending.append(sp)
        is_block = module is not getattr(self.model, "linear", None)
        if not is_block or len(self.pending) >= self.every:
            self._flush()

    def flush_rest(self):
        """The stretches no module has reported (embedding, position table: the front of the buffer)."""
        self.model._grad_bucket_hook = None
        self._flush()
        if self.low > 0:
            self.emit(0, self.low)
            self.low = 0

    def finish(self):
        """All-reduce whatever has not been sent (the front of the buffer) and wait for every bucket."""
        self.model._grad_bucket_hook = None
        if not is_distributed():
            return
        self.flush_rest()
        wait_all(self.handles, self.optim.flat_grad, self.average)


def bucket_plan(model, optim, blocks_per_bucket=4):
    """The all-reduce schedule of one training step: the (lo, hi) stretches of FusedAdam's flat gradient buffer in the
    order the explicit backward finishes them (mask head, encoder blocks L-1 .. 0 in groups of ``blocks_per_bucket``,
    then the front of the buffer).  It is a function of the model and the optimiser layout ONLY, and both forms of the
    step - eager (OverlappedGradReducer) and captured (GraphedTrainStep) - issue exactly these collectives in this
    order, so ranks may take different forms in the same step (a rank that meets a new batch shape runs eager while
    the others replay a graph) without their collectives pairing up wrongly."""
    plan = []
    r = OverlappedGradReducer(model, optim, blocks_per_bucket, emit=lambda lo, hi: plan.append((lo, hi)))
    r.begin(force=True)
    enc = getattr(model, "conformer", None)
    if enc is not None and hasattr(enc, "encoders"):   # the Conformer backward reports head and blocks (autograd.py)
        r._on_module(model.linear)
        for lay in reversed(list(enc.encoders)):
            r._on_module(lay)
    r.flush_rest()                                      # (other models: one stretch, the whole buffer)
    return plan


def allreduce_plan(flat, plan, average=True, async_handles=None):
    """Issue the stretches of ``bucket_plan`` back to back (asynchronous on RCCL) and wait, unless ``async_handles``."""
    if not is_distributed():
        return
    handles = [] if async_handles is None else async_handles
    for lo, hi in plan:
        allreduce_segment(flat, lo, hi, handles, average)
    if async_handles is None:
        wait_all(handles, flat, average)


def average_parameters(models_state_dicts):
    """combine_models.py:55-82, statement by statement (pinned by tests/golden/combine.npz, which is the output of the
    reference script itself on three job checkpoints):
      * tensors with at least one dimension: zeroed (:57-59), then ``add_(p, alpha=1/n)`` job by job in the tensor's
        own dtype (:73-76) - the f32 summation order of the reference;
      * 0-dim tensors (BatchNorm ``num_batches_tracked``): NOT accumulated - every job REPLACES the entry with
        ``(p * fraction).type(dtype)`` (:77-78), so the result is the LAST job's counter times 1/n, truncated
        (three jobs at 500, 500, 501 steps -> 167).  Reproduced as is: the counter only feeds BatchNorm's
        cumulative-average mode (momentum=None), which the reference models do not use."""
    n = len(models_state_dicts)
    fraction = 1.0 / n
    out = {}
    for k, v in models_state_dicts[0].items():
        if v.dim() == 0:
            out[k] = (models_state_dicts[-1][k] * fraction).type(v.dtype)
            continue
        acc = torch.zeros_like(v)
        for sd in models_state_dicts:
            acc.add_(sd[k].to(acc.device), alpha=fraction)
        out[k] = acc
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
