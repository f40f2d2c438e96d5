"""One-process-per-GPU data parallelism over torch.distributed (backend "nccl" = RCCL on ROCm).

The reference has no collective at all: "multi-node" training launches independent jobs and
averages their checkpoints on disk once per epoch (scripts/bash/train_parallel.sh:56-87,
scripts/python/combine_models.py:55-82).  Here every step sums ONE flat gradient bucket across
ranks -- a single large all-reduce is what the point-to-point xGMI links want (21.6 M params =
86 MB fp32 for Conformer-base) -- and long-recording inference shards window indices across
ranks with no collective on the data path.
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    """True when gradients have to be exchanged: an initialised process group with more than one rank.
    CSS_AMD_DRY_DIST=1 (a rehearsal switch, tests/test_gpu_dist.py) takes the distributed code path with ONE rank as
    well: every collective of the step is then really issued through the backend (RCCL on a one-GPU box: AVG
    all-reduces of the real bucket schedule between sub-graph replays) and is an identity."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CSS_AMD_DRY_DIST") == "1"


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
    if flat.is_cuda and use_p2p() and lo % 4 == 0:   # hand-written direct all-reduce over peer-mapped buffers (off by default)
        handles.append(p2p_reducer(flat).allreduce_async(lo, hi, average))   # on its own stream; the mean is taken inside
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
    if flat is not None and average and is_distributed() and dist.get_backend() != "nccl" and not (flat.is_cuda and use_p2p()):
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


class P2PAllReduce:
    """Direct (one-shot) all-reduce of FusedAdam's flat gradient buffer over peer-mapped memory - the hand-written variant
    SURVEY 5.8 keeps beside RCCL's collective (replaces scripts/python/combine_models.py:55-82).  OFF by default:
    ``CSS_AMD_P2P=1`` (or ``args.p2p``) makes allreduce_segment / the trainers use it; what RCCL picks for an 86 MB
    bucket on an 8-GPU xGMI node and which of the two is faster is an A/B for the first run on such a node.

    A ring all-reduce moves 2 (W-1)/W of the buffer over ONE link per GPU; here every rank reduces ONE 1/W slice, reading
    it from all W - 1 peers at once over W - 1 different links, and then fetches the other W - 1 reduced slices the same
    way: 2/W of the buffer per link.  Mechanics, all of it stream-ordered (no in-kernel flags, no cross-device polling):
      * set-up (once): every rank exports its flat buffer (hipIpc memory handle via torch's CUDA-tensor reductions) and
        three interprocess events; handles travel through ``dist.all_gather_object``; a small POSIX shared-memory block
        holds one sequence number per (phase, rank);
      * per call, three phases.  In each, a rank records its event on its stream, publishes the call's sequence number
        and - for every peer - waits on the HOST until the peer has published it too (the peer has then ISSUED its
        record: an interprocess event must be recorded before it is waited for) and makes its stream wait for the
        peer's event.  (1) "ready": gradients are final -> csship_p2p_reduce sums this rank's slice over all buffers
        in rank order, in place; (2) "reduced": -> the other slices are copied from their owners; (3) "done": nobody
        still reads this rank's buffer -> the call returns, the stream may overwrite the gradients.
    Sums run in rank order 0 .. W-1 on ONE rank per slice, so every rank ends up with the same bits, whatever the timing.
    Visibility across devices rests on the event semantics of the runtime (a recorded interprocess event releases to
    system scope, a stream that waited for one acquires): nothing here reads another device's memory except behind such
    a wait.  Covered on one GPU by two rank processes that share the device (tests/test_gpu_dist.py)."""

    PHASES = 3

    def __init__(self, flat, timeout_s=60.0):
        import mmap
        import numpy as np
        from torch.multiprocessing.reductions import reduce_tensor
        assert dist.is_initialized() and flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        if flat.data_ptr() % 16:
            raise ValueError("P2PAllReduce: the flat buffer must be 16-byte aligned")
        self.flat, self.rank, self.world = flat, dist.get_rank(), dist.get_world_size()
        if self.world > 8:
            raise ValueError("P2PAllReduce: one node, at most 8 ranks")
        self.timeout_s, self.seq = float(timeout_s), 0
        dev = flat.device
        # interprocess events: created by a first record, then exported
        self.events = [torch.cuda.Event(enable_timing=False, interprocess=True) for _ in range(self.PHASES)]
        for e in self.events:
            e.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        mine = {"tensor": reduce_tensor(flat), "events": [e.ipc_handle() for e in self.events], "n": flat.numel()}
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine)
        if any(o["n"] != flat.numel() for o in everyone):
            raise ValueError("P2PAllReduce: ranks hold gradient buffers of different sizes")
        self.peers, self.peer_events = [], []
        for j, o in enumerate(everyone):
            if j == self.rank:
                self.peers.append(flat)
                self.peer_events.append(self.events)
            else:
                fn, args = o["tensor"]
                self.peers.append(fn(*args))          # this process's mapping of rank j's buffer
                self.peer_events.append([torch.cuda.Event.from_ipc_handle(dev, h) for h in o["events"]])
        import ctypes as C
        self._ptrs = (C.c_void_p * self.world)(*[t.data_ptr() for t in self.peers])
        # sequence numbers [phase][rank] in shared memory (rank 0 names the file)
        name = [None]
        if self.rank == 0:
            name[0] = f"/dev/shm/css_amd_p2p_{os.getpid()}_{id(self) & 0xffffff:x}"
            with open(name[0], "wb") as fh:
                fh.write(b"\0" * (8 * self.PHASES * self.world))
        dist.broadcast_object_list(name, src=0)
        self._shm_name = name[0]
        self._fh = open(self._shm_name, "r+b")
        self._mm = mmap.mmap(self._fh.fileno(), 8 * self.PHASES * self.world)
        self._seqs = np.frombuffer(self._mm, dtype=np.int64).reshape(self.PHASES, self.world)
        dist.barrier()
        if self.rank == 0:
            os.unlink(self._shm_name)   # every rank has it mapped: the name can go

    def _phase(self, k, stream):
        """record my event of phase k, publish, and make `stream` wait for every peer's event of this call"""
        import time
        self.events[k].record(stream)
        self._seqs[k, self.rank] = self.seq
        deadline = time.monotonic() + self.timeout_s
        for j in range(self.world):
            if j == self.rank:
                continue
            while self._seqs[k, j] < self.seq:
                if time.monotonic() > deadline:
                    raise RuntimeError(f"css_amd.P2PAllReduce: rank {j} did not reach phase {k} of call {self.seq} within "
                                       f"{self.timeout_s:.0f} s (ranks must issue the same all-reduces in the same order)")
            stream.wait_event(self.peer_events[j][k])

    def allreduce_async(self, lo=0, hi=None, average=True):
        """The same on this reducer's own stream, ordered behind what the current stream has been given so far (the
        backward pass goes on beside it); returns a handle whose wait() orders the current stream behind the result."""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.flat.device)
        self._side.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self._side):
            self.allreduce(lo, hi, average)
        return _StreamHandle(self._side)

    def slices(self, lo, hi):
        """[(lo_j, hi_j)]: the part of [lo, hi) rank j reduces (multiples of 4 floats; the last ranks may get nothing)"""
        n = hi - lo
        per = ((n + self.world - 1) // self.world + 3) // 4 * 4
        return [(min(hi, lo + j * per), min(hi, lo + (j + 1) * per)) for j in range(self.world)]

    def allreduce(self, lo=0, hi=None, average=True):
        """In-place all-reduce (mean or sum) of flat[lo:hi] on the current stream; lo a multiple of 4."""
        from ._lib import check, lib
        hi = self.flat.numel() if hi is None else hi
        assert 0 <= lo <= hi <= self.flat.numel() and lo % 4 == 0
        stream = torch.cuda.current_stream(self.flat.device)
        self.seq += 1
        sl = self.slices(lo, hi)
        self._phase(0, stream)                                   # everybody's gradients are final
        a, b = sl[self.rank]
        check(lib().csship_p2p_reduce(self._ptrs, self.world, a, b, (1.0 / self.world) if average else 1.0,
                                      self.flat.data_ptr(), stream.cuda_stream))
        self._phase(1, stream)                                   # every slice is reduced at its owner
        for j, (a, b) in enumerate(sl):
            if j != self.rank and b > a:
                self.flat[a:b].copy_(self.peers[j][a:b], non_blocking=True)
        self._phase(2, stream)                                   # nobody reads my buffer any more


class _StreamHandle:
    """what allreduce_async returns: wait() makes the CURRENT stream wait for the communication stream (no host block)"""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)


_P2P = {}


def p2p_reducer(flat):
    """The P2PAllReduce of this flat buffer (built on first use: a collective set-up - every rank must get here)."""
    key = (flat.data_ptr(), flat.numel())
    r = _P2P.get(key)
    if r is None:
        r = _P2P[key] = P2PAllReduce(flat)
    return r


def use_p2p():
    return os.environ.get("CSS_AMD_P2P") == "1"


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
