"""Whole training step as ONE HIP graph.

A css_amd training step is ~2 300 kernel launches (forward operators, PIT, every gradient GEMM, optimiser) issued
from Python; at 23 ms of GPU work the host cannot keep up.  ``GraphedTrainStep`` captures
zero_grad -> objective(model, batch) -> backward -> clip + Adam once and replays it per step.  In a multi-GPU job
(torch.distributed initialised, one process per GPU) the forward + backward is captured as a CHAIN OF SUB-GRAPHS cut
where a stretch of FusedAdam's flat gradient buffer becomes final (the mask head, then every ``blocks_per_bucket``
encoder blocks: the backward's ``_grad_bucket_hook`` ends one capture and begins the next), and the replay issues that
stretch's all-reduce (RCCL over xGMI, asynchronous, on RCCL's own stream behind an event) BETWEEN two sub-graph
replays: the all-reduce of blocks 15..12 travels while blocks 11..8 are back-propagated, as in the eager step's
OverlappedGradReducer.  No collective is captured, so nothing depends on the runtime's support for
capturing RCCL; the last graph is clip + Adam, replayed after every handle has been waited for.  The schedule of
collectives is ``parallel.bucket_plan`` - the one the eager step issues (same stretches, same order) - so a rank that
runs a step eagerly (a batch shape it has not captured) stays paired with ranks that replay.  The warm-up steps before
the capture issue NO collective.  Replaces the reference's per-epoch
checkpoint averaging (scripts/python/combine_models.py:55-82).  Common to both forms:
inputs are copied into static buffers, the learning rate / Adam bias corrections live in device memory
(``FusedAdam.advance_hyper``), BatchNorm bookkeeping and the PIT permutation search are device-side already.
One captured step serves ONE batch shape; the reference draws the window length per batch (separation.py:94-95,175:
2 to 5 s), so train_one_epoch keeps a small LRU of captured shapes and runs shapes it has seen only once eagerly.
"""
import torch

from .. import parallel
from ..optim import FusedAdam


class GraphedTrainStep:
    def __init__(self, model, objective, optim, example_batch, max_norm, device="cuda", warmup=2, buckets=4,
                 blocks_per_bucket=4):
        if not isinstance(optim, FusedAdam):
            raise TypeError("GraphedTrainStep needs css_amd.optim.FusedAdam (static parameter / gradient buffers)")
        self.distributed = parallel.is_distributed()
        self.buckets = buckets   # (kept for callers; the collective schedule is bucket_plan's)
        self.plan = parallel.bucket_plan(model, optim, blocks_per_bucket) if self.distributed else []
        self.model, self.objective, self.optim, self.max_norm = model, objective, optim, max_norm
        self.keys = [k for k in ("mix", "source1", "source2") if k in example_batch]
        self.static = {k: example_batch[k].to(device).clone() for k in self.keys}
        # warm-up steps outside the capture (allocator pools, lazy initialisation) must not train: snapshot every
        # piece of state they touch and put it back
        if hasattr(model, "conformer"):   # buffers that the training forward re-homes must be in place before the snapshot
            from ..autograd import bn_counters
            bn_counters(model)
        snap = [t.clone() for t in (optim.flat_param, optim.exp_avg, optim.exp_avg_sq)]
        bufs = [(b, b.clone()) for b in model.buffers()]
        step0 = optim._step
        def restore():
            torch.cuda.synchronize()
            for t, s_ in zip((optim.flat_param, optim.exp_avg, optim.exp_avg_sq), snap):
                t.copy_(s_)
            for b, s_ in bufs:
                b.copy_(s_)
            optim._step = step0

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager(collective=False)   # ranks capture at different times: no collective in here
        except Exception:   # a step that cannot run: leave model and optimiser as they were, eager-capable
            restore()
            raise
        torch.cuda.current_stream().wait_stream(side)
        restore()
        self.graph = torch.cuda.CUDAGraph()
        # buffers outside the graph's pool that the captured backward shares with the eager steps (autograd._skew_buffer:
        # zero outside what the kernels write, zeroed once by the warm-up steps above): alive as long as this object
        self._zero_once = []
        model._skew_captured = self._zero_once
        try:
            if not self.distributed:
                # thread_local: other threads of this process may touch the runtime during the capture
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.loss, self.grad_norm = self._eager()
            else:
                self._capture_distributed(blocks_per_bucket)
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt, pool=self.graphs[0].pool(), capture_error_mode="thread_local"):
                    self.grad_norm = self.optim.step(max_norm=self.max_norm)
        except Exception:   # the capture does not execute anything; put back what the warm-up touched
            restore()
            raise
        finally:
            model._skew_captured = None

    def _capture_distributed(self, blocks_per_bucket):
        """zero_grad + forward + PIT + backward as sub-graphs cut at the bucket boundaries of parallel.bucket_plan.  The
        cut happens inside the backward pass (autograd's device thread), so the captures run in "relaxed" mode: a capture
        may be ended by another thread than the one that began it."""
        import gc
        self.graphs, self.cuts = [torch.cuda.CUDAGraph()], []
        cutting = [True]

        def on_final(lo, hi):
            self.cuts.append((lo, hi))
            if cutting[0]:   # stretches reported while the backward is still running end a sub-graph
                self.graphs[-1].capture_end()
                g = torch.cuda.CUDAGraph()
                g.capture_begin(pool=self.graphs[0].pool(), capture_error_mode="relaxed")
                self.graphs.append(g)
        reducer = parallel.OverlappedGradReducer(self.model, self.optim, blocks_per_bucket, emit=on_final)
        torch.cuda.synchronize()
        gc.collect()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            self.graphs[0].capture_begin(capture_error_mode="relaxed")
            try:
                reducer.begin(force=True)
                self.loss = self._forward_backward()
                cutting[0] = False
                reducer.flush_rest()      # the front of the buffer (embedding, position table): after the last kernel
            finally:
                self.model._grad_bucket_hook = None
                self.graphs[-1].capture_end()
        torch.cuda.current_stream().wait_stream(stream)
        self.n_cut = len(self.graphs) - 1     # all-reduces issued between sub-graph replays; the rest after the last one
        if self.cuts != self.plan:
            raise RuntimeError(f"css_amd: captured bucket boundaries {self.cuts} differ from parallel.bucket_plan {self.plan}")
        self.graph = self.graphs[0]

    def _forward_backward(self):
        self.optim.zero_grad()
        loss = self.objective(self.model, self.static, device=self.static[self.keys[0]].device)
        loss.backward()
        return loss.detach()

    def _eager(self, collective=True):
        loss = self._forward_backward()
        if self.distributed and collective:
            parallel.allreduce_plan(self.optim.flat_grad, self.plan)
        return loss, self.optim.step(max_norm=self.max_norm)

    def release(self):
        """Drop the captured graphs and their private memory pool (LRU eviction in train_one_epoch)."""
        self.graph = None
        self.graph_opt = None
        self.graphs = []
        self.static = {}
        self._zero_once = []

    def __call__(self, batch):
        """One optimiser step on ``batch``; returns (loss, grad_norm) as 0-dim device tensors of THIS step (they are
        overwritten by the next call)."""
        for k in self.keys:
            self.static[k].copy_(batch[k], non_blocking=True)
        self.optim.advance_hyper()
        if not self.distributed:
            self.graph.replay()
            return self.loss, self.grad_norm
        handles = []
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < self.n_cut:   # this sub-graph finished a stretch: its all-reduce travels beside the next sub-graph
                parallel.allreduce_segment(self.optim.flat_grad, *self.cuts[i], handles)
        for lo, hi in self.cuts[self.n_cut:]:
            parallel.allreduce_segment(self.optim.flat_grad, lo, hi, handles)
        parallel.wait_all(handles, self.optim.flat_grad)
        self.graph_opt.replay()
        return self.loss, self.grad_norm
