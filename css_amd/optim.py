"""Fused gradient clipping + Adam (csship_adam_clip) on flat parameter / gradient buffers.

``FusedAdam`` re-homes every parameter of the model as a view into ONE flat f32 buffer and gives every parameter a
persistent ``.grad`` view into a second one.  The step is then three kernel launches with no host synchronisation,
and the flat gradient buffer is the all-reduce bucket of ``css_amd.parallel`` (no gather / scatter copies).
It follows torch.optim.Adam's update rule and ``param_groups`` protocol (``LRScheduler`` sets ``group["lr"]``), so it
drops into ``train_one_epoch`` in place of ``torch.optim.Adam`` + ``clip_grad_norm_``.
"""
import torch

from . import bump_param_generation
from ._lib import check, lib
from .ops import _ptr, _stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0):
        params = [p for p in params]
        if not params:
            raise ValueError("FusedAdam got an empty parameter list")
        if any(isinstance(p, dict) for p in params):
            raise ValueError("css_amd.FusedAdam takes ONE parameter group (a flat list of tensors): the fused step "
                             "applies one lr / betas / weight_decay to the whole flat buffer")
        if any((not p.is_cuda) or p.dtype != torch.float32 for p in params):
            raise RuntimeError("css_amd.FusedAdam needs float32 parameters on the GPU (no CPU path exists)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_norm=max_norm))
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4  # 16-byte aligned segments
        self._n = n
        self._offs = offs
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros((), dtype=torch.float32, device=dev)
        self._ws = torch.empty(max(lib().csship_adam_clip_workspace(n), 16), dtype=torch.uint8, device=dev)
        self._step = 0
        # {lr, 1 - beta1^t, 1 - beta2^t} on the device: read by the kernel, so a captured step can be replayed
        self._hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros((16, 3), dtype=torch.float32).pin_memory()  # one staging slot per step in flight
        self._hyper_events = [None] * 16
        with torch.no_grad():
            for p, o in zip(params, offs):
                seg = self.flat_param[o:o + p.numel()].view(p.shape)
                seg.copy_(p.data)
                p.data = seg
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
                p.grad._csship_owner = self   # lets the explicit backward write a gradient IN PLACE while it is known zero
        # ids of the parameters whose .grad view has been written since the last zero_grad(): the explicit backward
        # (css_amd/autograd.py) stores a weight gradient straight into a view that is still zero instead of producing a
        # temporary and adding it (0 + g == g), and accumulates as before into one that is not
        self._dirty = set()
        # A store is only right while the view is still zero, and a torch-side write the explicit backward does not
        # know of (AccumulateGrad of a second loss term on the same parameters, a manual ``p.grad.add_``, gradients a
        # caller restores) breaks that.  Every ``.grad`` view shares the flat buffer's version counter, which any
        # in-place torch operation on any of them bumps - kernels that write through raw pointers do not - so "no
        # foreign write since zero_grad()" is "the counter still reads what zero_grad() left" (grads_untouched()).
        # AccumulateGrad is caught per parameter as well (post-accumulate hook below).
        self._grad_version = self.flat_grad._version
        self._foreign = False
        import weakref
        me = weakref.ref(self)

        def written(p):
            o = me()
            if o is not None:
                o._dirty.add(id(p))
        for p in params:
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(written)

    def zero_grad(self, set_to_none=False):
        """Gradients stay views of the flat buffer (set_to_none is ignored): one fill."""
        self.flat_grad.zero_()
        self._dirty.clear()
        self._grad_version = self.flat_grad._version
        self._foreign = False

    def grads_untouched(self):
        """True while no torch operation has written into the flat gradient buffer (any ``.grad`` view) since
        zero_grad() - the condition under which css_amd.autograd may STORE a gradient instead of accumulating it.
        Once False it stays False until the next zero_grad()."""
        if self._foreign:
            return False
        if self.flat_grad._version != self._grad_version:
            self._foreign = True
            return False
        return True

    def own_write(self, was_untouched):
        """css_amd's own torch-side writes into the buffer (the multi-tensor add fallback, an overlapped bucket
        all-reduce) re-arm the check - only if it held right before them."""
        if was_untouched and not self._foreign:
            self._grad_version = self.flat_grad._version

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """Clip to ``max_norm`` (default: the constructor's; <= 0 disables) and apply one Adam step.
        Returns the pre-clip gradient norm as a 0-dim device tensor (no synchronisation)."""
        if closure is not None:
            raise RuntimeError("css_amd.FusedAdam does not take a closure")
        g = self.param_groups[0]
        mn = g["max_norm"] if max_norm is None else max_norm
        # lr and the bias corrections (computed in double on the host, like torch.optim.Adam) are read from device
        # memory.  A step launched kernel by kernel uploads them here; a step that is being CAPTURED does not (the
        # host code runs once, the graph many times): GraphedTrainStep calls advance_hyper() before every replay, so
        # eager and replayed steps are bit-identical and may be mixed freely (the decision is per call, not a mode of
        # the optimiser: train_one_epoch runs shapes it has not captured eagerly between replays).
        if not torch.cuda.is_current_stream_capturing():
            self.advance_hyper()
        hyper, step = self._hyper, 0
        bump_param_generation()  # the kernel below rewrites the parameter storage: weight-shadow caches are stale
        check(lib().csship_adam_clip(_ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                     self._n, float(mn), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                     float(g["eps"]), step, float(g["weight_decay"]), _ptr(hyper), _ptr(self.grad_norm),
                                     _ptr(self._ws), self._ws.numel(), _stream()))
        return self.grad_norm

    def advance_hyper(self):
        """Count one step and upload {lr, bias corrections} for it (asynchronous pinned copy on the current stream,
        ordered before the kernel launch or graph replay that follows).  step() calls it unless it is being captured."""
        g = self.param_groups[0]
        self._step += 1
        bump_param_generation()  # replayed steps: the graph rewrites the parameter storage after this call
        # The copy is asynchronous and the host may be several steps ahead of the GPU: every step stages its values in
        # its own pinned slot, reused only after the copy that read it has completed (a single staging buffer would be
        # overwritten with a later step's values before an earlier copy ran).
        slot = self._step % len(self._hyper_events)
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()
        stage = self._hyper_host[slot]
        stage[0] = float(g["lr"])
        stage[1] = 1.0 - float(g["betas"][0]) ** self._step
        stage[2] = 1.0 - float(g["betas"][1]) ** self._step
        self._hyper.copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_events[slot] = ev

    def add_param_group(self, group):
        if getattr(self, "param_groups", None):
            raise ValueError("css_amd.FusedAdam supports a single parameter group")
        super().add_param_group(group)

    def state_dict(self):
        """torch.optim.Adam's format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``, as views of the flat
        moment buffers), so that the reference's train.py resume and combine_models.py read the moments, plus the
        flat buffers themselves under ``fused``."""
        sd = super().state_dict()
        params = self.param_groups[0]["params"]
        step = torch.tensor(float(self._step))
        sd["state"] = {i: {"step": step.clone(), "exp_avg": self.exp_avg[o:o + p.numel()].view(p.shape),
                           "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view(p.shape)}
                       for i, (p, o) in enumerate(zip(params, self._offs))}
        sd["fused"] = {"step": self._step, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}
        return sd

    def load_state_dict(self, sd):
        """Accepts this class's own checkpoints and torch.optim.Adam's (the reference's ``optimizer`` entry,
        train.py:209-225): the per-parameter moments are scattered into the flat buffers."""
        fused, state = sd.get("fused"), sd.get("state") or {}
        if len(sd.get("param_groups", [])) != 1:
            raise ValueError("css_amd.FusedAdam supports a single parameter group")
        super().load_state_dict({"state": {}, "param_groups": sd["param_groups"]})
        for k, v in self.defaults.items():   # a torch.optim.Adam checkpoint has no FusedAdam-only keys (max_norm)
            self.param_groups[0].setdefault(k, v)
        if fused is not None:
            self._step = int(fused["step"])
            self.exp_avg.copy_(fused["exp_avg"])
            self.exp_avg_sq.copy_(fused["exp_avg_sq"])
        elif state:
            params = self.param_groups[0]["params"]
            ids = sd["param_groups"][0]["params"]
            steps = set()
            with torch.no_grad():
                for p, o, pid in zip(params, self._offs, ids):
                    st = state.get(pid, state.get(str(pid)))
                    if st is None:
                        continue
                    self.exp_avg[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
                    self.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
                    steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError(f"css_amd.FusedAdam: per-parameter step counts differ ({sorted(steps)}); one shared "
                                 "step is required")
            self._step = steps.pop() if steps else 0
        bump_param_generation()
