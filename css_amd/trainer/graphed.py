"""Whole training step as ONE HIP graph.

A css_amd training step is ~2 300 kernel launches (forward operators, PIT, every gradient GEMM, optimiser) issued
from Python; at 23 ms of GPU work the host cannot keep up.  ``GraphedTrainStep`` captures
zero_grad -> objective(model, batch) -> backward -> clip + Adam once and replays it per step.  In a multi-GPU job
(torch.distributed initialised, one process per GPU) the step is TWO graphs with the gradient all-reduce between them:
graph 1 = zero_grad, forward, PIT, backward; then the stretches of ``parallel.bucket_plan`` on FusedAdam's flat
gradient buffer (RCCL over xGMI, asynchronous, all in flight; issued eagerly - no collective is captured, so nothing
depends on the runtime's support for capturing RCCL); graph 2 = clip + Adam.  The schedule of collectives is the one
the eager step issues (same stretches, same order), so a rank that runs a step eagerly - a batch shape it has not
captured - stays paired with ranks that replay.  The warm-up steps before the capture issue NO collective.  Replaces the reference's per-epoch
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
        optim.capturable = True
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
                    optim.advance_hyper()
                    self._eager(collective=False)   # ranks capture at different times: no collective in here
        except Exception:   # a step that cannot run: leave model and optimiser as they were, eager-capable
            restore()
            optim.capturable = False
            raise
        torch.cuda.current_stream().wait_stream(side)
        restore()
        # thread_local: other threads of this process (the RCCL watchdog) may touch the runtime during the capture
        self.graph = torch.cuda.CUDAGraph()
        try:
            if not self.distributed:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.loss, self.grad_norm = self._eager()
            else:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.loss = self._forward_backward()
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), capture_error_mode="thread_local"):
                    self.grad_norm = self.optim.step(max_norm=self.max_norm)
        except Exception:   # the capture does not execute anything, but a failed one must not leave the optimiser capturable
            restore()
            optim.capturable = False
            raise

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
        self.static = {}

    def __call__(self, batch):
        """One optimiser step on ``batch``; returns (loss, grad_norm) as 0-dim device tensors of THIS step (they are
        overwritten by the next call)."""
        for k in self.keys:
            self.static[k].copy_(batch[k], non_blocking=True)
        self.optim.advance_hyper()
        self.graph.replay()
        if self.distributed:
            parallel.allreduce_plan(self.optim.flat_grad, self.plan)
            self.graph_opt.replay()
        return self.loss, self.grad_norm
