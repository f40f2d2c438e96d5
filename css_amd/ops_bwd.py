"""Torch-tensor front door to the backward kernels of libcsship.so (the ``_bwd`` twins).

Same rules as ``ops.py``: HIP tensors only, shapes validated on the host, outputs from torch's caching
allocator, launches on the current stream.  The reference obtains these gradients from torch.autograd.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, lib
from .ops import _need, _ptr, _stream

_ws_cache = {}


_deferred_ws = []   # workspaces of the calls whose reductions are queued (csship_reduce_defer): alive until the flush
_defer = {"on": False, "group": False, "dev": None}
_NO_DEFER = bool(os.environ.get("CSSHIP_NO_DEFER"))
_NO_GROUP = bool(os.environ.get("CSSHIP_NO_GROUP_WGRAD"))   # A/B timing: every weight-gradient product as its own launch


class deferred_reductions:
    """``with deferred_reductions() as d: ... d.flush() ...``: inside, the slab reductions that produce PARAMETER gradients
    out of SMALL slab sets (layernorm_bwd, ffn_bwd_fused: dgamma / dbeta partials) are queued and run as one launch per
    ``flush()`` (include/csship.h, csship_reduce_defer).  Those calls get their own workspace, kept until the flush.  Nobody may read those gradients before the flush; the
    exit flushes what is left (or drops it when an exception is travelling)."""

    def __init__(self, group_wgrads=False):
        self.group = bool(group_wgrads) and not _NO_GROUP and not _NO_DEFER
        self.held = []     # flush(side=...): operands / slabs of launches that may still be running on the side stream
        self.side = None
        self.held_prev, self.held_cur = [], []   # reduce_side_stream(): the last two flushes' slabs

    def __enter__(self):
        if _NO_DEFER:   # A/B timing: every reduction as its own launch
            self.was = 1
            return self
        self.was = lib().csship_reduce_defer(1)
        _defer["on"] = True
        if self.group:   # weight-gradient products queued too: one launch per flush (csship_wgrad_hold)
            self.was_group = lib().csship_wgrad_hold(1)
            _defer["group"] = True
        return self

    def flush(self, side=None):
        """side: a second stream - the queued launches produce parameter gradients only, nothing on the data-gradient chain
        waits for them: they run on ``side`` behind everything the current stream has been given so far, beside what it is
        given next.  Their operands stay referenced (``held``) until ``join()``, which the caller owes before anyone reads
        those gradients (and before a stream capture ends)."""
        if side is not None and (lib().csship_wgrad_pending() or lib().csship_reduce_pending()):
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            self.side = side
            self.held.extend(_deferred_ws)
            with torch.cuda.stream(side):
                self.flush()
            return
        if _reduce_side["on"]:   # the launches of this flush may run on the side stream until the NEXT flush has begun (it joins first)
            self.held_prev, self.held_cur = self.held_cur, list(_deferred_ws)
        if lib().csship_wgrad_pending():
            nws = lib().csship_wgrad_group_workspace()
            dev = _defer["dev"]
            key = ("group", dev, torch.cuda.current_stream().cuda_stream)
            ws = _ws_cache.get(key)
            if ws is None or ws.numel() < nws:
                ws = _ws_cache[key] = torch.empty(nws, dtype=torch.uint8, device=dev)
            check(lib().csship_wgrad_flush(_ptr(ws), nws, _stream()))
        elif lib().csship_reduce_pending():
            check(lib().csship_reduce_flush(_stream()))
        _deferred_ws.clear()

    def join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
            self.side = None
        self.held.clear()

    def __exit__(self, et, ev, tb):
        try:
            if et is None:
                self.flush()
            self.join()
        finally:
            if self.group and not self.was_group:
                lib().csship_wgrad_hold(0)
                _defer["group"] = False
            if not self.was:
                lib().csship_reduce_defer(0)
                _defer["on"] = False
                _deferred_ws.clear()
        return False


def pos_bwd_max_T():
    """longest chunk csship_attn_pos_bwd(_ds) takes (the kernel's own constant, not a copy of it)"""
    return int(lib().csship_attn_pos_bwd_max_t())


def ffn_static_max_ffn():
    """widest FeedForward of the static row-panel forms (csship_ffn_bwd_ln, the chain kernel's static form)"""
    return int(lib().csship_ffn_static_max_ffn())


_hold = {"on": False, "ws": []}
_NO_PAIR = bool(os.environ.get("CSSHIP_NO_PAIR_REDUCE"))   # A/B timing: every weight gradient's slab reduction as its own launch


_reduce_side = {"on": False}


class reduce_side_stream:
    """``with reduce_side_stream(side):`` - inside, the library's parameter-gradient reductions (the block flushes, reductions too
    large to queue, the relative-position window sums) run on ``side`` beside the current stream (csship_reduce_side_stream).  The
    exit joins the current stream with it.  Slabs of a deferred_reductions flush stay referenced for one more flush."""

    def __init__(self, side):
        self.side = side

    def __enter__(self):
        if self.side is not None:
            check(lib().csship_reduce_side_stream(self.side.cuda_stream))
            _reduce_side["on"] = True
        return self

    def join(self):
        if self.side is not None:
            check(lib().csship_reduce_side_join(_stream()))

    def __exit__(self, et, ev, tb):
        if self.side is not None:
            try:
                self.join()
            finally:
                _reduce_side["on"] = False
                lib().csship_reduce_side_stream(None)
        return False


class paired_reductions:
    """``with paired_reductions(): dw2 = linear_wgrad(..); dw1 = linear_wgrad(..)``: the slab reductions of the products
    inside are queued (csship_reduce_hold) and run as ONE launch at the exit - for two weight gradients of one operator
    (2 x 16 MB of slabs, both still in the last-level cache), not for more (a whole block's slabs came back from HBM).
    Every product inside gets a workspace of its own, alive until the flush."""

    def __enter__(self):
        if not _NO_PAIR and not _defer["group"]:   # (grouped weight gradients: the products themselves are queued)
            lib().csship_reduce_hold(1)
            _hold["on"] = True
        return self

    def __exit__(self, et, ev, tb):
        if _hold["on"]:
            _hold["on"] = False
            lib().csship_reduce_hold(0)
            try:
                if et is None and lib().csship_reduce_pending():
                    check(lib().csship_reduce_flush(_stream()))
                    _deferred_ws.clear()   # (the flush ran every queued job, the deferred small ones included)
                elif et is not None and not _defer["on"]:
                    # an exception between the products: the held jobs point at the workspaces freed below.  Outside
                    # a deferred_reductions scope nobody else drops them (switching the deferred mode off does, and
                    # restores the mode it found); inside one, the outer scope's error path drops the whole queue.
                    lib().csship_reduce_defer(lib().csship_reduce_defer(0))
            finally:
                _hold["ws"].clear()
        return False


def _workspace(nbytes, dev, own=False):
    """Scratch reused across calls on one stream (kernels on a stream run in order).  own: the call's reduction is one
    of the deferrable ones - while reductions are deferred it gets a workspace of its own (the slabs live in it until
    the flush)."""
    if _hold["on"]:   # a held reduction reads this product's slabs after the NEXT product has run
        t = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _hold["ws"].append(t)
        return t
    if own and _defer["on"]:
        t = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _deferred_ws.append(t)
        return t
    key = (dev, torch.cuda.current_stream().cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = t
    return t


def _as4(t, name):
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"css_amd: '{name}' must be float32 or bfloat16")
    if not t.is_cuda:
        raise RuntimeError(f"css_amd: '{name}' must be a CUDA/HIP tensor (no CPU path exists)")
    if t.dim() == 2:
        t = t[None, None]
    elif t.dim() == 3:
        t = t[None]
    if t.dim() != 4 or t.stride(-1) != 1:
        raise RuntimeError(f"css_amd: '{name}' must be [.., rows, cols] with contiguous columns")
    return t


def bgemm(a, b, *, a_kmajor=False, b_kmajor=False, out=None, out_bf16=False, alpha=1.0, bias=None, mask_pos=None,
          add=None, splits=0, band=0, band_stride=0):
    """C = alpha * opA(a) @ opB(b)^T-style product, batched over leading dims.
    band > 0: a is banded and reduction tiles outside an output tile's band are skipped (csship_bgemm_params.band):
    row-major a zero unless m <= k <= m + band; kmajor a with band_stride and splits: rows k = z * (K / splits) +
    t * band_stride + h, zero unless t <= m <= t + band.

    a: [.., M, K] (or [.., K, M] when a_kmajor); b: [.., N, K] (or [.., K, N] when b_kmajor); result [.., M, N].
    mask_pos: bf16 [.., M, N], zeroes the result where it is <= 0; add: f32 [.., M, N] added to the result.
    """
    squeeze = a.dim()
    a4, b4 = _as4(a, "a"), _as4(b, "b")
    b1, b2 = max(a4.shape[0], b4.shape[0]), max(a4.shape[1], b4.shape[1])
    M, K = (a4.shape[3], a4.shape[2]) if a_kmajor else (a4.shape[2], a4.shape[3])
    N, Kb = (b4.shape[3], b4.shape[2]) if b_kmajor else (b4.shape[2], b4.shape[3])
    if K != Kb:
        raise RuntimeError(f"css_amd: bgemm reduction mismatch {K} vs {Kb}")

    def bstr(t):
        return (t.stride(0) if t.shape[0] > 1 else 0, t.stride(1) if t.shape[1] > 1 else 0)

    if out is None:
        out = torch.empty((b1, b2, M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
        out_ret = out.reshape((M, N)) if squeeze == 2 else (out[0] if squeeze == 3 else out)
    else:
        out_ret = out
        out = _as4(out, "out")
        if out.shape[-2:] != (M, N):
            raise RuntimeError("css_amd: bgemm out shape")
        out_bf16 = out.dtype == torch.bfloat16
    aux = None
    epi = _lib.BEPI_OUT_BF16 if out_bf16 else 0
    if mask_pos is not None:
        aux = _as4(_need(mask_pos, torch.bfloat16, "mask_pos"), "mask_pos")
        epi |= _lib.BEPI_MASK_POS
    if add is not None:
        aux = _as4(_need(add, torch.float32, "add"), "add")
        epi |= _lib.BEPI_ADD
    if aux is not None and aux.shape[-2:] != (M, N):
        raise RuntimeError("css_amd: bgemm aux must have the output's shape")
    p = _lib.BgemmParams(M=M, N=N, K=K, lda=a4.stride(2), ldb=b4.stride(2), ldc=out.stride(2),
                         ldaux=aux.stride(2) if aux is not None else 0,
                         a_kmajor=int(a_kmajor), b_kmajor=int(b_kmajor), a_f32=int(a4.dtype == torch.float32),
                         b_f32=int(b4.dtype == torch.float32), epilogue=epi, batch1=b1, batch2=b2, splits=splits,
                         sA1=bstr(a4)[0], sA2=bstr(a4)[1], sB1=bstr(b4)[0], sB2=bstr(b4)[1],
                         sC1=bstr(out)[0], sC2=bstr(out)[1],
                         sX1=bstr(aux)[0] if aux is not None else 0, sX2=bstr(aux)[1] if aux is not None else 0,
                         alpha=float(alpha), band=int(band), band_stride=int(band_stride) if a_kmajor else 0)
    nws = lib().csship_bgemm_workspace(C.byref(p))
    # (a product accumulated in place is a gradient accumulator: its slab sum may wait for the block's flush)
    ws = _workspace(nws, a.device, own=add is not None and out is not None and add.data_ptr() == out.data_ptr()) if nws else None
    check(lib().csship_bgemm(_ptr(a4), _ptr(b4), _ptr(out), _ptr(bias), _ptr(aux), C.byref(p), _ptr(ws), nws, _stream()))
    return out_ret


_SPLIT_WGRAD = os.environ.get("CSS_AMD_SPLIT_WGRAD", "0") == "1"  # A/B timing: the product and the column sums as two calls


def linear_wgrad(dy, x, alpha=1.0, dw_out=None, db_out=None, head=None):
    """(dW, db) of y = x W^T + b from dy [rows, M] and x [rows, N] (f32 or bf16, rows contiguous): dW = alpha * dy^T x
    f32 [M, N], db = alpha * column sums of dy f32 [M] - ONE pass over dy (csship_bgemm_wgrad) when the widths are
    multiples of 8, else the product and the column sums as two calls.  head = (rows, alpha_head): the first ``rows``
    output rows take alpha_head instead of alpha."""
    if dy.dim() != 2 or x.dim() != 2 or dy.shape[0] != x.shape[0]:
        raise RuntimeError("css_amd: linear_wgrad expects dy [rows, M] and x [rows, N]")
    K, M = dy.shape
    N = x.shape[1]
    if (M | N) & 7 or dy.stride(1) != 1 or x.stride(1) != 1 or dy.stride(0) % 8 or x.stride(0) % 8 or _SPLIT_WGRAD:
        dw = bgemm(dy, x, a_kmajor=True, b_kmajor=True, alpha=alpha, out=dw_out)
        db = colsum(dy, alpha, out=db_out)
        if head is not None and alpha != 0:
            dw[:head[0]] *= head[1] / alpha
            db[:head[0]] *= head[1] / alpha
        return dw, db
    dw = torch.empty((M, N), dtype=torch.float32, device=dy.device) if dw_out is None else dw_out
    db = torch.empty(M, dtype=torch.float32, device=dy.device) if db_out is None else db_out
    if dw.dtype != torch.float32 or dw.shape != (M, N) or dw.stride(1) != 1 or db.dtype != torch.float32 or \
            db.shape != (M,) or not db.is_contiguous():
        raise RuntimeError("css_amd: linear_wgrad outputs must be f32 [M, N] (contiguous columns) and contiguous f32 [M]")
    p = _lib.BgemmParams(M=M, N=N, K=K, lda=dy.stride(0), ldb=x.stride(0), ldc=dw.stride(0), ldaux=0, a_kmajor=1,
                         b_kmajor=1, a_f32=int(dy.dtype == torch.float32), b_f32=int(x.dtype == torch.float32), epilogue=0,
                         batch1=1, batch2=1, splits=0, sA1=0, sA2=0, sB1=0, sB2=0, sC1=0, sC2=0, sX1=0, sX2=0,
                         alpha=float(alpha))
    nws = lib().csship_bgemm_wgrad_workspace(C.byref(p))
    ws = _workspace(nws, dy.device) if nws else None
    if _defer["group"]:   # the product may be queued until the flush: operands and outputs stay alive till then
        _deferred_ws.append((dy, x, dw, db))
        _defer["dev"] = dy.device
    check(lib().csship_bgemm_wgrad(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), None, C.byref(p),
                                   float(head[1]) if head else 0.0, int(head[0]) if head else 0, _ptr(ws), nws, _stream()))
    return dw, db


def colsum(x, alpha=1.0, out=None):
    """alpha * column sums of a [M, N] f32/bf16 matrix (row stride free) -> f32 [N]."""
    if x.dim() != 2 or x.stride(1) != 1 or not x.is_cuda:
        raise RuntimeError("css_amd: colsum expects a HIP [M, N] matrix with contiguous columns")
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    elif out.dtype != torch.float32 or out.shape != (N,) or not out.is_contiguous():
        raise RuntimeError("css_amd: colsum out must be a contiguous f32 [N] vector")
    nws = lib().csship_colsum_workspace(M, N)
    ws = _workspace(nws, x.device)
    check(lib().csship_colsum(_ptr(x), int(x.dtype == torch.float32), M, N, x.stride(0), float(alpha), _ptr(out),
                              _ptr(ws), nws, _stream()))
    return out


def layernorm_bwd(x, dy, gamma, beta, dx_add=None, relu=False, out=None, want_y=False):
    """Backward of y = LayerNorm(x) [-> ReLU]: returns (dx (+ dx_add), dgamma, dbeta).  x, dy: f32 [M, D] dense.
    want_y: also LayerNorm(x) as bf16 [M, D] (the pass recomputes the normalised rows anyway)."""
    _need(x, torch.float32, "x"), _need(dy, torch.float32, "dy")
    M, D = x.shape
    if not (x.is_contiguous() and dy.is_contiguous()) or dy.shape != x.shape:
        raise RuntimeError("css_amd: layernorm_bwd expects dense [M, D] x and dy")
    dx = torch.empty_like(x) if out is None else out
    gb = torch.empty((2, D), dtype=torch.float32, device=x.device)
    nws = lib().csship_layernorm_bwd_workspace(M, D)
    ws = _workspace(nws, x.device, own=True)
    y = torch.empty((M, D), dtype=torch.bfloat16, device=x.device) if want_y else None
    check(lib().csship_layernorm_bwd_y(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), _ptr(dx_add), int(relu), M, D, _ptr(dx),
                                       _ptr(gb), _ptr(y), _ptr(ws), nws, _stream()))
    return (dx, gb[0], gb[1], y) if want_y else (dx, gb[0], gb[1])


def relu_bwd(dy, y, out=None):
    """dy where y > 0 else 0; 2-D f32 with free row strides."""
    _need(dy, torch.float32, "dy"), _need(y, torch.float32, "y")
    if dy.dim() != 2 or dy.shape != y.shape or dy.stride(1) != 1 or y.stride(1) != 1:
        raise RuntimeError("css_amd: relu_bwd expects matching 2-D tensors")
    out = torch.empty_strided(dy.shape, dy.stride(), dtype=dy.dtype, device=dy.device) if out is None else out
    check(lib().csship_relu_bwd(_ptr(dy), dy.stride(0), _ptr(y), y.stride(0), dy.shape[0], dy.shape[1], _ptr(out),
                                out.stride(0), _stream()))
    return out


def relu_bwd_bf16(dy, y, ld_out):
    """relu_bwd into bf16 [M, ld_out], the columns past dy's width zero (ld_out even)."""
    _need(dy, torch.float32, "dy"), _need(y, torch.float32, "y")
    if dy.dim() != 2 or dy.shape != y.shape or dy.stride(1) != 1 or y.stride(1) != 1 or ld_out < dy.shape[1] or ld_out & 1:
        raise RuntimeError("css_amd: relu_bwd_bf16 expects matching 2-D tensors and an even ld_out >= their width")
    out = torch.empty((dy.shape[0], ld_out), dtype=torch.bfloat16, device=dy.device)
    check(lib().csship_relu_bwd_bf16(_ptr(dy), dy.stride(0), _ptr(y), y.stride(0), dy.shape[0], dy.shape[1], _ptr(out),
                                     ld_out, _stream()))
    return out


def mvn_apply(f, mean, rinv, ld_out=None):
    """bf16 [B*T, ld_out] = MVN(f) (zero padded columns) for f f32 [B, T, F]."""
    _need(f, torch.float32, "f")
    B, T, F = f.shape
    if f.stride(2) != 1 or f.stride(0) != T * f.stride(1):
        f = f.contiguous()
    ld_out = (F + 7) // 8 * 8 if ld_out is None else ld_out
    out = torch.empty((B * T, ld_out), dtype=torch.bfloat16, device=f.device)
    check(lib().csship_mvn_apply(_ptr(f), B, T, F, f.stride(1), float("-inf"), _ptr(mean), _ptr(rinv), _ptr(out), ld_out,
                                 _stream()))
    return out


def _pad8(n):
    return (n + 7) // 8 * 8


def relpos_softmax(s_qk, rel, T):
    """s_qk f32 [B,H,T,>=T], rel f32 [B,H,T,>=2T-1] (dense) -> P bf16 [B,H,T,pad8(T)] (zero padded)."""
    B, H = s_qk.shape[:2]
    if not (s_qk.is_contiguous() and rel.is_contiguous()):
        raise RuntimeError("css_amd: relpos_softmax expects dense inputs")
    p = torch.empty((B, H, T, _pad8(T)), dtype=torch.bfloat16, device=s_qk.device)
    check(lib().csship_relpos_softmax(_ptr(s_qk), s_qk.shape[3], _ptr(rel), rel.shape[3], B, H, T, _ptr(p), p.shape[3],
                                      _stream()))
    return p


def relpos_softmax_bwd(p, dp, T, W):
    """P bf16 [B,H,T,ldp], dP f32 [B,H,T,ldd] -> dS bf16 [B,H,T,ldp], dS_skew bf16 [B,T,H,W]."""
    B, H = p.shape[:2]
    if not (p.is_contiguous() and dp.is_contiguous()):
        raise RuntimeError("css_amd: relpos_softmax_bwd expects dense inputs")
    ds = torch.empty_like(p)
    dsk = torch.empty((B, T, H, W), dtype=torch.bfloat16, device=p.device)
    check(lib().csship_relpos_softmax_bwd(_ptr(p), p.shape[3], _ptr(dp), dp.shape[3], B, H, T, _ptr(ds), _ptr(dsk), W,
                                          _stream()))
    return ds, dsk


def convmod_bwd(x, dy, weights, fwd_ws, kernel_size=33, bn_eval=False, drop=None, sync=False, want_next=False, next_drop=None,
                flat=None, stats_done=False):
    """Backward of ops.convmod(x, weights, batch_stats=True) (or keep_activations=True with bn_eval=True).  x, dy f32 [B,T,D]; fwd_ws: that call's workspace.
    Returns dx and a dict of f32 gradients keyed like the weights dict (dw_w as [k, D]).  want_next (not with sync): a
    third result bf16 [B*T, D] = dropout(dx) under ``next_drop`` (None: bf16(dx)), the operand of the next operator of the
    backward pass (csship_convmod_bwd_next).  flat + stats_done (not with sync): the gradient buffer in which
    ffn_bwd(conv=...) already left the statistics pass's sums (csship_convmod_bwd_after_stats)."""
    from .ops import conv_weights_struct
    _need(x, torch.float32, "x"), _need(dy, torch.float32, "dy")
    x, dy = x.contiguous(), dy.contiguous()
    B, T, D = x.shape
    if stats_done and (flat is None or sync):
        raise RuntimeError("css_amd: convmod_bwd(stats_done=True) needs the gradient buffer the statistics were left in (no SyncBatchNorm)")
    if flat is None:
        flat = torch.empty(38 * D + 8, dtype=torch.float32, device=x.device)
    cw = conv_weights_struct(weights)
    dx = torch.empty_like(x)
    nws = lib().csship_convmod_bwd_workspace(B, T, D)
    ws = _workspace(nws, x.device, own=True)   # (the last pass's partial sums may wait for the block's flush)
    if sync and not bn_eval:  # SyncBatchNorm: the two channel sums of the BN input gradient are over all ranks' rows
        from .parallel import allreduce_sum_
        import torch.distributed as dist
        args = (_ptr(x), _ptr(dy), C.byref(cw), B, T, D, kernel_size, _ptr(fwd_ws), _ptr(dx), _ptr(flat))
        check(lib().csship_convmod_bwd_stats(*args, drop, _ptr(ws), nws, _stream()))
        off = lib().csship_convmod_bwd_sums_offset(B, T, D)
        allreduce_sum_(ws[off:off + 8 * D].view(torch.float32))
        check(lib().csship_convmod_bwd_finish(*args, float(B * T * dist.get_world_size()), drop, _ptr(ws), nws, _stream()))
    elif stats_done:
        nxt = torch.empty((B * T, D), dtype=torch.bfloat16, device=x.device) if want_next else None
        check(lib().csship_convmod_bwd_after_stats(_ptr(x), _ptr(dy), C.byref(cw), B, T, D, kernel_size, _ptr(fwd_ws), _ptr(dx),
                                                   _ptr(flat), int(bn_eval), drop, next_drop, _ptr(nxt), _ptr(ws), nws, _stream()))
    elif want_next:
        nxt = torch.empty((B * T, D), dtype=torch.bfloat16, device=x.device)
        check(lib().csship_convmod_bwd_next(_ptr(x), _ptr(dy), C.byref(cw), B, T, D, kernel_size, _ptr(fwd_ws), _ptr(dx),
                                            _ptr(flat), int(bn_eval), drop, next_drop, _ptr(nxt), _ptr(ws), nws, _stream()))
    else:
        check(lib().csship_convmod_bwd_drop(_ptr(x), _ptr(dy), C.byref(cw), B, T, D, kernel_size, _ptr(fwd_ws), _ptr(dx),
                                            _ptr(flat), int(bn_eval), drop, _ptr(ws), nws, _stream()))
    o = 2 * D + 4
    grads = {"bn_b": flat[:D], "bn_g": flat[D:2 * D], "pw2_w": flat[2 * D:2 * D + 1], "pw2_b": flat[2 * D + 1:2 * D + 2],
             "dw_w": flat[o:o + kernel_size * D].view(kernel_size, D), "dw_b": flat[o + 33 * D:o + 34 * D],
             "ln_g": flat[o + 34 * D:o + 35 * D], "ln_b": flat[o + 35 * D:o + 36 * D],
             "pw1_w": flat[o + 36 * D:o + 36 * D + 2], "pw1_b": flat[o + 36 * D + 2:o + 36 * D + 4]}
    if want_next:
        if sync and not bn_eval:
            raise RuntimeError("css_amd: convmod_bwd(want_next=True) is not built for SyncBatchNorm")
        return dx, grads, nxt
    return dx, grads


def pack_weight(w, *, scale=1.0, row_major=None, frag=None, row0=0, want_row_major=False, want_frag=False):
    """One-launch bf16 shadows of a Linear weight w f32 [R, K]: row-major (rows row0.. of ``row_major``) and / or MFMA
    fragment order (rows row0.. of ``frag``).  Allocates what is asked for and not given.  Returns (row_major, frag)."""
    _need(w, torch.float32, "w")
    R, K = w.shape
    if w.stride(1) != 1:
        w = w.contiguous()
    if want_row_major and row_major is None:
        row_major = torch.empty((R, (K + 7) // 8 * 8), dtype=torch.bfloat16, device=w.device)
    if want_frag and frag is None:
        frag = torch.empty(((R + 15) // 16 * 16) * ((K + 31) // 32 * 32), dtype=torch.bfloat16, device=w.device)
    check(lib().csship_pack_weight(_ptr(w), R, K, w.stride(0), float(scale), _ptr(row_major),
                                   row_major.stride(0) if row_major is not None else 0, _ptr(frag), row0, _stream()))
    return row_major, frag


class PackPlan:
    """All bf16 weight shadows of a model in ONE launch per step (csship_pack_weights_multi).

    ``add`` registers a job and returns its destination tensors (allocated once, rewritten by every ``run``); the job
    table is uploaded to the device the first time and again only if a source pointer moved."""

    def __init__(self, device):
        self.device = device
        self.jobs = []      # (w, row_major, frag, scale, transposed, row0, k0, k_total)
        self.table = None
        self.ptrs = None
        self.max_chunks = 0

    def add(self, w, *, scale=1.0, transposed=False, row_major=None, frag=None, row0=0, want_row_major=False,
            want_frag=False, k0=0, k_total=0):
        """w: f32 [R, K] (transposed=False) or the [K, R] storage of the logical [R, K] weight (transposed=True).
        k0 / k_total: this weight is the k-slice [k0, k0 + K) of a fragment table k_total wide (``frag`` given)."""
        _need(w, torch.float32, "w")
        if w.dim() != 2 or w.stride(1) != 1:
            raise RuntimeError("css_amd: PackPlan needs 2-d weights with contiguous rows")
        R, K = (w.shape[1], w.shape[0]) if transposed else w.shape
        if want_row_major and row_major is None:
            row_major = torch.empty((R, (K + 7) // 8 * 8), dtype=torch.bfloat16, device=w.device)
        if want_frag and frag is None:
            frag = torch.empty(((R + 15) // 16 * 16) * ((K + 31) // 32 * 32), dtype=torch.bfloat16, device=w.device)
        # (a slice that starts at 0 may end inside a k-step: the job writes zeros up to the step's end, the table beyond it stays
        # as allocated - zeros for a K-padded table like the embedding's)
        if k_total and (frag is None or k0 % 32 or k_total % 32 or (K % 32 and k0) or k0 + (K + 31) // 32 * 32 > k_total):
            raise RuntimeError("css_amd: a k-slice job needs its fragment table and whole 32-wide k-steps")
        self.jobs.append((w, row_major, frag, float(scale), bool(transposed), int(row0), int(k0), int(k_total)))
        self.table = None
        return row_major, frag

    def _build(self):
        arr = (_lib.PackJob * len(self.jobs))()
        self.max_chunks = 0
        for j, (w, rm, fr, scale, tr, row0, k0, k_total) in zip(arr, self.jobs):
            R, K = (w.shape[1], w.shape[0]) if tr else w.shape
            Kp = (K + 31) // 32 * 32 if fr is not None else (K + 7) // 8 * 8
            j.w, j.row_major, j.frag = w.data_ptr(), (rm.data_ptr() if rm is not None else None), (fr.data_ptr() if fr is not None else None)
            j.ld_w, j.ld_row_major = w.stride(0), (rm.stride(0) if rm is not None else 0)
            j.R, j.K, j.dst_row0, j.Kp, j.scale, j.transposed = R, K, row0, Kp, scale, int(tr)
            j.k0, j.Kp_dst = k0, k_total
            self.max_chunks = max(self.max_chunks, ((R + 15) // 16 * 16) * (Kp // 8))
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.ptrs = [w.data_ptr() for w, *_ in self.jobs]

    def run(self):
        if self.table is None or self.ptrs != [w.data_ptr() for w, *_ in self.jobs]:
            if torch.cuda.is_current_stream_capturing() and self.table is not None:
                raise RuntimeError("css_amd: a parameter moved while a HIP graph is being captured")
            self._build()
        check(lib().csship_pack_weights_multi(_ptr(self.table), len(self.jobs), self.max_chunks, _stream()))


def lstm_bidir_train(gx, whh_packed, B, T, H):
    """BLSTM recurrence that also returns what the backward needs: (y [B,T,2H], gates [B,T,2,4H], c [B,T,2,H])."""
    from . import ops
    return ops.lstm_bidir(gx, whh_packed, B, T, H, keep=True)


def lstm_bidir_bwd(gates, c, dy, whhT_packed):
    """Back-propagation through time: dy f32 [B,T,2H] -> dgates f32 [B,T,2,4H] (gradient wrt pre-activation gates)."""
    from . import ops
    B, T, _, H4 = gates.shape
    H = H4 // 4
    dy = dy.contiguous()
    dg = torch.empty_like(gates)
    if ops.lstm_dist_enabled(H, B, backward=True):
        key = (gates.device.index, _stream() or 0, B, H)
        ws = ops._LSTM_WS.get(key)
        if ws is None:
            ws = ops._LSTM_WS[key] = torch.zeros(lib().csship_lstm_dist_workspace(B, H), dtype=torch.uint8, device=gates.device)
        check(lib().csship_lstm_bidir_bwd_dist(_ptr(gates), _ptr(c), _ptr(dy), _ptr(whhT_packed), _ptr(dg), B, T, H,
                                              _ptr(ws), ws.numel(), _stream()))
    else:
        check(lib().csship_lstm_bidir_bwd(_ptr(gates), _ptr(c), _ptr(dy), _ptr(whhT_packed), _ptr(dg), B, T, H, _stream()))
    return dg


def ffn_bwd(x, dy, ln_g, ln_b, w1p, b1, w2tp, w1tp, d_ffn, alpha=0.5, drop_hidden=None, drop_out=None, ln2=None, conv=None):
    """Activation side of the FeedForward backward in one kernel (csship_ffn_bwd).
    conv = (fwd_ws, pw2_w, drop_cfg, flat, B, T) (not with ln2): x is the output of a conv module and dx its output gradient -
    the statistics pass of that module's backward rides in the epilogue (csship_ffn_bwd_conv) and leaves its sums in ``flat``,
    the gradient buffer convmod_bwd(..., flat=flat, stats_done=True) then finishes.
    Returns (dx f32 [M,256], dgamma, dbeta, a bf16 [M,256], h bf16 [M,d_ffn], dh bf16 [M,d_ffn], dyd): with dropout
    cfgs h is the DROPPED hidden activation and dyd (bf16 [M,256]) = alpha * dy o output mask / (1 - p), else None.
    ln2 = (x2, gamma2) with x = LayerNorm2(x2): dx is carried on through that LayerNorm (csship_ffn_bwd_ln) and the
    tuple grows by (dgamma2, dbeta2)."""
    _need(x, torch.float32, "x"), _need(dy, torch.float32, "dy")
    M, D = x.shape
    if not (x.is_contiguous() and dy.is_contiguous()) or dy.shape != x.shape:
        raise RuntimeError("css_amd: ffn_bwd expects dense [M, 256] x and dy")
    dev = x.device
    a = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    h = torch.empty((M, d_ffn), dtype=torch.bfloat16, device=dev)
    dh = torch.empty((M, d_ffn), dtype=torch.bfloat16, device=dev)
    dx = torch.empty_like(x)
    gb = torch.empty((2, D), dtype=torch.float32, device=dev)
    dyd = torch.empty((M, D), dtype=torch.bfloat16, device=dev) if drop_out is not None else None
    if ln2 is not None:
        x2, g2 = ln2
        _need(x2, torch.float32, "ln2 x")
        if not x2.is_contiguous() or x2.shape != x.shape:
            raise RuntimeError("css_amd: ffn_bwd expects the second LayerNorm's input dense and shaped like x")
        gb2 = torch.empty((2, D), dtype=torch.float32, device=dev)
        nws = lib().csship_ffn_bwd_ln_workspace(M)
        ws = _workspace(nws, dev, own=True)
        check(lib().csship_ffn_bwd_ln(_ptr(x), _ptr(dy), _ptr(ln_g), _ptr(ln_b), _ptr(w1p), _ptr(b1), _ptr(w2tp), _ptr(w1tp),
                                      _ptr(a), _ptr(h), _ptr(dh), _ptr(dx), _ptr(gb), M, D, d_ffn, float(alpha), drop_hidden,
                                      drop_out, _ptr(dyd), _ptr(x2), _ptr(g2), _ptr(gb2), _ptr(ws), nws, _stream()))
        return dx, gb[0], gb[1], a, h, dh, dyd, gb2[0], gb2[1]
    if conv is not None:
        fws, pw2_w, cdrop, flat, B, T = conv
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.numel() < 38 * D + 8 or B * T != M:
            raise RuntimeError("css_amd: ffn_bwd(conv=...) expects the conv module's f32 gradient buffer [38 D + 8] and M = B * T")
        nws = lib().csship_ffn_bwd_conv_workspace(M)
        ws = _workspace(nws, dev, own=True)
        check(lib().csship_ffn_bwd_conv(_ptr(x), _ptr(dy), _ptr(ln_g), _ptr(ln_b), _ptr(w1p), _ptr(b1), _ptr(w2tp), _ptr(w1tp),
                                        _ptr(a), _ptr(h), _ptr(dh), _ptr(dx), _ptr(gb), M, D, d_ffn, float(alpha), drop_hidden,
                                        drop_out, _ptr(dyd), _ptr(fws), B, T, _ptr(pw2_w), cdrop, _ptr(flat), _ptr(ws), nws, _stream()))
        return dx, gb[0], gb[1], a, h, dh, dyd
    nws = lib().csship_ffn_bwd_workspace(M)
    ws = _workspace(nws, dev, own=True)
    check(lib().csship_ffn_bwd_drop(_ptr(x), _ptr(dy), _ptr(ln_g), _ptr(ln_b), _ptr(w1p), _ptr(b1), _ptr(w2tp), _ptr(w1tp),
                                    _ptr(a), _ptr(h), _ptr(dh), _ptr(dx), _ptr(gb), M, D, d_ffn, float(alpha), drop_hidden,
                                    drop_out, _ptr(dyd), _ptr(ws), nws, _stream()))
    return dx, gb[0], gb[1], a, h, dh, dyd


def qkv_in_bwd(dqkv, wqkvT_p, x, dy, ln_g, ln_b):
    """Input side of the q | k | v projection's backward in one kernel (csship_qkv_in_bwd): dqkv bf16 [M, 768],
    x / dy f32 [M, 256], M % 64 == 0.  Returns (dx, dgamma, dbeta, y) with y = LayerNorm(x) as bf16."""
    _need(dqkv, torch.bfloat16, "dqkv"), _need(x, torch.float32, "x"), _need(dy, torch.float32, "dy")
    M, D = x.shape
    if not (x.is_contiguous() and dy.is_contiguous() and dqkv.is_contiguous()) or dy.shape != x.shape or \
            dqkv.shape != (M, 3 * D):
        raise RuntimeError("css_amd: qkv_in_bwd expects dense x, dy [M, 256] and dqkv [M, 768]")
    dev = x.device
    y = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    dx = torch.empty_like(x)
    gb = torch.empty((2, D), dtype=torch.float32, device=dev)
    nws = lib().csship_qkv_in_bwd_workspace(M)
    ws = _workspace(nws, dev, own=True)
    check(lib().csship_qkv_in_bwd(_ptr(dqkv), _ptr(wqkvT_p), _ptr(x), _ptr(dy), _ptr(ln_g), _ptr(ln_b), _ptr(dx), _ptr(y),
                                  _ptr(gb), M, D, _ptr(ws), nws, _stream()))
    return dx, gb[0], gb[1], y


def qkv_ffn_bwd(dqkv, wqkvT_p, x_attn, dy, attn_ln_g, attn_ln_b, x, ln_g, ln_b, w1p, b1, w2tp, w1tp, d_ffn, alpha=0.5,
                drop_hidden=None, drop_out=None, ln2=None):
    """qkv_in_bwd and the ffn_bwd behind it in the backward pass as ONE kernel (csship_qkv_ffn_bwd): the q | k | v
    projection's dx is the FeedForward's dy and never leaves the registers.  x_attn / dy: the attention block's input and
    output gradient; x: the FeedForward's input; the rest as in qkv_in_bwd / ffn_bwd (ln2 included).  M % 64 == 0,
    d_ffn <= ffn_static_max_ffn(), drop_out given (the FeedForward's dW2 takes dyd: dy itself exists nowhere).
    Returns ((attn dgamma, attn dbeta, y), ffn_bwd's tuple)."""
    _need(dqkv, torch.bfloat16, "dqkv"), _need(x_attn, torch.float32, "x_attn"), _need(dy, torch.float32, "dy"), _need(x, torch.float32, "x")
    M, D = x.shape
    if not (x.is_contiguous() and dy.is_contiguous() and dqkv.is_contiguous() and x_attn.is_contiguous()) or \
            dy.shape != x.shape or x_attn.shape != x.shape or dqkv.shape != (M, 3 * D):
        raise RuntimeError("css_amd: qkv_ffn_bwd expects dense x, x_attn, dy [M, 256] and dqkv [M, 768]")
    if drop_out is None:
        raise RuntimeError("css_amd: qkv_ffn_bwd needs the FeedForward's output dropout cfg (dyd is its only copy of dy)")
    dev = x.device
    y = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    gbq = torch.empty((2, D), dtype=torch.float32, device=dev)
    a = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    h = torch.empty((M, d_ffn), dtype=torch.bfloat16, device=dev)
    dh = torch.empty((M, d_ffn), dtype=torch.bfloat16, device=dev)
    dx = torch.empty_like(x)
    gb = torch.empty((2, D), dtype=torch.float32, device=dev)
    dyd = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    x2 = g2 = gb2 = None
    if ln2 is not None:
        x2, g2 = ln2
        _need(x2, torch.float32, "ln2 x")
        if not x2.is_contiguous() or x2.shape != x.shape:
            raise RuntimeError("css_amd: qkv_ffn_bwd expects the second LayerNorm's input dense and shaped like x")
        gb2 = torch.empty((2, D), dtype=torch.float32, device=dev)
    nws = lib().csship_qkv_ffn_bwd_workspace(M)
    ws = _workspace(nws, dev, own=True)
    check(lib().csship_qkv_ffn_bwd(_ptr(dqkv), _ptr(wqkvT_p), _ptr(x_attn), _ptr(dy), _ptr(attn_ln_g), _ptr(attn_ln_b), _ptr(y),
                                   _ptr(gbq), _ptr(x), _ptr(ln_g), _ptr(ln_b), _ptr(w1p), _ptr(b1), _ptr(w2tp), _ptr(w1tp),
                                   _ptr(a), _ptr(h), _ptr(dh), _ptr(dx), _ptr(gb), M, D, d_ffn, float(alpha), drop_hidden,
                                   drop_out, _ptr(dyd), _ptr(x2), _ptr(g2), _ptr(gb2), _ptr(ws), nws, _stream()))
    res = (dx, gb[0], gb[1], a, h, dh, dyd) + ((gb2[0], gb2[1]) if ln2 is not None else ())
    return (gbq[0], gbq[1], y), res


def qkv_ffn2_bwd(dqkv, wqkvT_p, x_attn, dy, attn_ln_g, attn_ln_b, fin, fout, conv, d_ffn, alpha=0.5):
    """qkv_ffn_bwd with the FeedForward backward of the block BELOW behind it, the conv module's statistics pass included
    (csship_qkv_ffn2_bwd: three stages, one launch).  fin / fout: dicts with x, ln_g, ln_b, w1p, b1, w2tp, w1tp, drop_hidden,
    drop_out (fin also ln2 = (x2, gamma2): the chain runs through that LayerNorm); conv = (fwd_ws, pw2_w, drop_cfg, flat, B, T) as
    in ffn_bwd.  Returns ((attn dgamma, attn dbeta, y), ffn_bwd's tuple for fin (its dx is None: it never exists), ffn_bwd's tuple
    for fout)."""
    _need(dqkv, torch.bfloat16, "dqkv"), _need(x_attn, torch.float32, "x_attn"), _need(dy, torch.float32, "dy")
    M, D = x_attn.shape
    fws, pw2_w, cdrop, flat, B, T = conv
    x2, g2 = fin["ln2"]
    for t in (x_attn, dy, dqkv, fin["x"], fout["x"], x2):
        if not t.is_contiguous():
            raise RuntimeError("css_amd: qkv_ffn2_bwd expects dense rows")
    if dy.shape != x_attn.shape or fin["x"].shape != x_attn.shape or fout["x"].shape != x_attn.shape or x2.shape != x_attn.shape or \
            dqkv.shape != (M, 3 * D) or B * T != M:
        raise RuntimeError("css_amd: qkv_ffn2_bwd expects x_attn, dy, both FeedForward inputs and the LayerNorm's input as [M, 256], dqkv [M, 768], M = B * T")
    if fin.get("drop_out") is None or fout.get("drop_out") is None:
        raise RuntimeError("css_amd: qkv_ffn2_bwd needs both FeedForwards' output dropout cfgs (dyd is the only copy of their dy)")
    if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.numel() < 38 * D + 8:
        raise RuntimeError("css_amd: qkv_ffn2_bwd expects the conv module's f32 gradient buffer [38 D + 8]")
    dev = x_attn.device
    bf = torch.bfloat16
    y = torch.empty((M, D), dtype=bf, device=dev)
    gbq = torch.empty((2, D), dtype=torch.float32, device=dev)
    qa = _lib.QkvBwdArgs(_ptr(dqkv), _ptr(wqkvT_p), _ptr(x_attn), _ptr(dy), _ptr(attn_ln_g), _ptr(attn_ln_b), _ptr(y), _ptr(gbq))

    def block(f, want_dx):
        a = torch.empty((M, D), dtype=bf, device=dev)
        h = torch.empty((M, d_ffn), dtype=bf, device=dev)
        dh = torch.empty((M, d_ffn), dtype=bf, device=dev)
        dyd = torch.empty((M, D), dtype=bf, device=dev)
        gb = torch.empty((2, D), dtype=torch.float32, device=dev)
        dx = torch.empty((M, D), dtype=torch.float32, device=dev) if want_dx else None
        st = _lib.FfnBwdArgs()
        st.x, st.ln_g, st.ln_b, st.w1_packed, st.b1 = _ptr(f["x"]), _ptr(f["ln_g"]), _ptr(f["ln_b"]), _ptr(f["w1p"]), _ptr(f["b1"])
        st.w2t_packed, st.w1t_packed = _ptr(f["w2tp"]), _ptr(f["w1tp"])
        st.a_bf16, st.h_bf16, st.dh_bf16, st.dx, st.dgamma_dbeta = _ptr(a), _ptr(h), _ptr(dh), _ptr(dx), _ptr(gb)
        st.drop_hidden, st.drop_out, st.dyd_bf16 = f.get("drop_hidden"), f["drop_out"], _ptr(dyd)
        return st, (dx, gb[0], gb[1], a, h, dh, dyd)

    fi, res_in = block(fin, False)
    gb2 = torch.empty((2, D), dtype=torch.float32, device=dev)
    fi.ln2_x, fi.ln2_g, fi.dgamma_dbeta2 = _ptr(x2), _ptr(g2), _ptr(gb2)
    fo, res_out = block(fout, True)
    cv = _lib.ConvStatsArgs(_ptr(fws), B, T, _ptr(pw2_w), cdrop, _ptr(flat))
    nws = lib().csship_qkv_ffn2_bwd_workspace(M)
    ws = _workspace(nws, dev, own=True)
    check(lib().csship_qkv_ffn2_bwd(C.byref(qa), C.byref(fi), C.byref(fo), C.byref(cv), M, D, d_ffn, float(alpha), _ptr(ws), nws,
                                    _stream()))
    return (gbq[0], gbq[1], y), res_in + (gb2[0], gb2[1]), res_out


def grad_accumulate(pairs):
    """dst += src for every (dst, src) pair in ONE launch per 96 pairs (csship_grad_accumulate).  dst: dense f32; src: f32
    with dst's element count, dense in dst's order, or a 2-d [R, C] view with stride(0) == 1 (the transpose of a matrix
    with dense rows: the depthwise conv weight gradient) or with rows further apart than C.  Returns the pairs it could not take (other layouts / dtypes)."""
    jobs, rest, keep = [], [], []
    for dst, src in pairs:
        ok = (dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.is_contiguous() and src.is_cuda and dst.is_cuda
              and src.numel() == dst.numel() and src.numel() > 0)
        if ok and src.is_contiguous():
            jobs.append((src.data_ptr(), dst.data_ptr(), src.numel(), 0, 0))
        elif ok and src.dim() == 2 and src.stride(0) == 1 and src.stride(1) >= src.shape[0]:
            jobs.append((src.data_ptr(), dst.data_ptr(), src.numel(), src.stride(1), src.shape[1]))
        elif ok and src.dim() == 2 and src.stride(1) == 1 and src.stride(0) >= src.shape[1]:   # padded row stride
            jobs.append((src.data_ptr(), dst.data_ptr(), src.numel(), src.stride(0), -src.shape[1]))
        else:
            rest.append((dst, src))
            continue
        keep.append(src)
    if jobs:
        arr = (_lib.GradJob * len(jobs))()
        for a, (sp, dp, n, ld, cols) in zip(arr, jobs):
            a.src, a.dst, a.n, a.ld_src, a.cols = sp, dp, n, ld, cols
        check(lib().csship_grad_accumulate(arr, len(jobs), _stream()))
    return rest


def ema_update(dsts_head, srcs_head, w_head, dsts_tail, srcs_tail, w_tail, keep):
    """dst = keep * dst + w * src for two lists of dense f32 tensors (w_head / w_tail) in ONE launch per 96 tensors
    (csship_ema_update): BatchNorm's running statistics of every block."""
    pairs = [(d, s_, True) for d, s_ in zip(dsts_head, srcs_head)] + [(d, s_, False) for d, s_ in zip(dsts_tail, srcs_tail)]
    for d, s_, _ in pairs:
        _need(d, torch.float32, "dst"), _need(s_, torch.float32, "src")
        if not (d.is_contiguous() and s_.is_contiguous() and d.numel() == s_.numel() and d.numel() > 0):
            raise RuntimeError("css_amd: ema_update takes dense tensors of equal size")
    for at in range(0, len(pairs), 96):   # (a chunk keeps its head jobs in front: the lists are already in that order)
        chunk = pairs[at:at + 96]
        arr = (_lib.GradJob * len(chunk))()
        for a, (d, s_, _) in zip(arr, chunk):
            a.src, a.dst, a.n, a.ld_src, a.cols = s_.data_ptr(), d.data_ptr(), d.numel(), 0, 0
        check(lib().csship_ema_update(arr, len(chunk), float(keep), float(w_head), sum(1 for p in chunk if p[2]),
                                      float(w_tail), _stream()))


def attn_rowdot(d_out, out, B, T, H):
    """D[b,h,i] = sum_d dO[b,i,h,d] * O[b,i,h,d] for bf16 [B*T, H*64] operands -> f32 [B,H,T]."""
    _need(d_out, torch.bfloat16, "d_out"), _need(out, torch.bfloat16, "out")
    d = torch.empty((B, H, T), dtype=torch.float32, device=out.device)
    check(lib().csship_attn_rowdot(_ptr(d_out.contiguous()), _ptr(out.contiguous()), B, T, H, _ptr(d), _stream()))
    return d


def mha_bwd_core(qkv, pe_pack, lse, d_out, d, skew, n_head, drop=None, out=None, want_q_rows=False):
    """Probabilities, softmax backward and dS k in one pass (csship_mha_relpos_bwd_core) -> (p_v, dS, skew, dq1[, q_rows]):
    p_v bf16 [B,H,T,pad8(T)] the operand of dV (the dropped probabilities with a dropout cfg), dS bf16 alike, skew
    bf16 [B,T,H,W] filled in place inside the band, dq1 f32 [B,H,T,64] = dS k.  d = rowsum(dO o O) f32 [B,H,T]
    (attn_rowdot), or None with ``out`` = the forward's output bf16 [B*T, H*64]: the pass computes it.  want_q_rows:
    also the q rows as bf16 [B*T*H, 64] in skew's row order."""
    _need(qkv, torch.bfloat16, "qkv"), _need(lse, torch.float32, "lse"), _need(d_out, torch.bfloat16, "d_out")
    if skew is not None:   # None: no skewed copy is written (attn_pos_bwd_ds skews the row-major dS itself)
        _need(skew, torch.bfloat16, "skew")
    if d is not None:
        _need(d, torch.float32, "d")
    elif out is None:
        raise RuntimeError("css_amd: mha_bwd_core needs d or the forward's output")
    else:
        _need(out, torch.bfloat16, "out")
    packed, q_min, q_max = pe_pack
    B, T, D3 = qkv.shape
    if not (qkv.is_contiguous() and d_out.is_contiguous() and lse.is_contiguous()) \
            or (skew is not None and (not skew.is_contiguous() or skew.shape[:3] != (B, T, n_head))) \
            or (d is not None and not d.is_contiguous()) or (out is not None and not out.is_contiguous()) \
            or D3 != 3 * n_head * 64:
        raise RuntimeError("css_amd: mha_bwd_core expects dense qkv [B,T,3*H*64], d_out [B*T,H*64], skew [B,T,H,W]")
    ldp = (T + 7) // 8 * 8
    p_v = torch.empty(B, n_head, T, ldp, device=qkv.device, dtype=torch.bfloat16)
    ds = slack_bf16((B, n_head, T, ldp), qkv.device)   # (16 bytes of slack on both sides: attn_pos_bwd_ds)
    dq1 = torch.empty(B, n_head, T, 64, device=qkv.device, dtype=torch.float32)
    q_rows = torch.empty(B * T * n_head, 64, device=qkv.device, dtype=torch.bfloat16) if want_q_rows else None
    check(lib().csship_mha_relpos_bwd_core(_ptr(qkv), _ptr(packed), q_min, q_max, _ptr(lse), _ptr(d_out), _ptr(d),
                                           _ptr(out) if d is None else None, _ptr(p_v), _ptr(ds), _ptr(skew), _ptr(dq1),
                                           _ptr(q_rows), ldp, skew.shape[3] if skew is not None else 0, B, T, n_head, drop, _stream()))
    return (p_v, ds, skew, dq1, q_rows) if want_q_rows else (p_v, ds, skew, dq1)


def attn_dvdk(p_v, ds, d_out, qkv, dqkv, n_head):
    """dV = P'^T dO and dK = dS^T q of the attention in one streaming pass over the two [B,H,T,ldp] bf16 operands
    (csship_attn_dvdk); writes the k and v thirds of dqkv bf16 [B,T,3*H*64]."""
    _need(p_v, torch.bfloat16, "p_v"), _need(ds, torch.bfloat16, "ds"), _need(qkv, torch.bfloat16, "qkv")
    _need(d_out, torch.bfloat16, "d_out"), _need(dqkv, torch.bfloat16, "dqkv")
    B, T, _ = qkv.shape
    if not (p_v.is_contiguous() and ds.is_contiguous() and d_out.is_contiguous() and qkv.is_contiguous() and dqkv.is_contiguous()) \
            or p_v.shape != ds.shape or tuple(p_v.shape[:3]) != (B, n_head, T) or dqkv.shape != qkv.shape:
        raise RuntimeError("css_amd: attn_dvdk expects dense [B,H,T,ldp] operands and [B,T,3*H*64] projections")
    check(lib().csship_attn_dvdk(_ptr(p_v), _ptr(ds), p_v.shape[3], _ptr(d_out), _ptr(qkv), _ptr(dqkv), B, T, n_head, _stream()))
    return dqkv


_CONST = {}            # key -> tensor(s), least recently used first
_CONST_KEEP = set()    # keys a captured graph reads through their address: never evicted
_CONST_MAX = 64


def cached_const(key, build):
    """Step-invariant index / scale tensors: built once per key, OUTSIDE graph capture only (a tensor allocated while a
    step is being captured lives in the graph's pool and must not outlive the capture; such a step rebuilds it).  At
    most _CONST_MAX entries (variable-length batches would otherwise keep one index per length for ever); an entry a
    captured graph has read stays - the graph replays its address."""
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    t = _CONST.pop(key, None)
    if t is None:
        t = build()
        if capturing:
            return t
        if len(_CONST) >= _CONST_MAX:
            for k in list(_CONST):
                if k not in _CONST_KEEP:
                    del _CONST[k]
                    break
    _CONST[key] = t
    if capturing:
        _CONST_KEEP.add(key)
    return t


def relpos_tables(pe, T, maxlen=1000, want_t=True):
    """Every bf16 form of the relative-position table pe f32 [2*maxlen, 64] that one training step of length T reads,
    by THREE launches (table + zero row -> bf16 -> one gather through a cached index) instead of the ~27 small torch
    launches of ops.pack_relpos + backward.relpos_window + pe_window_t, whose results these are bit for bit:
    {"pe_bf": bf16 table, "pack": (packed, q_min, q_max) of ops.pack_relpos, "window": (window, idx[, window^T]) of
    backward.relpos_window / pe_window_t}."""
    _need(pe, torch.float32, "pe")
    dk = pe.shape[1]
    if tuple(pe.shape) != (2 * maxlen, 64):
        raise RuntimeError("css_amd: relpos_tables expects the [2 * maxlen, 64] table")
    dev = pe.device
    q_min, q_max = -((maxlen + 15) // 16) - 2, (maxlen + 15) // 16 + 1   # ops.pack_relpos
    W = (2 * T - 1 + 7) // 8 * 8
    nblk, nks = (T + 63) // 64, (T + 63 + 31) // 32
    ld = (max(W, (nblk - 1) * 64 + nks * 32) + 7) // 8 * 8               # pe_window_t

    def build():
        zero_row = 2 * maxlen
        E = torch.arange((2 * maxlen + 1) * dk, device=dev).view(2 * maxlen + 1, dk)
        q, m = torch.arange(q_min, q_max + 1, device=dev), torch.arange(16, device=dev)
        rel = (16 * q[:, None] + 15 - m[None, :]).clamp(-maxlen, maxlen - 1) + maxlen
        parts = [E[rel].view(-1, 16, 2, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)]
        idx = (torch.arange(2 * T - 1, device=dev) - (T - 1)).clamp(-maxlen, maxlen - 1) + maxlen
        rows = torch.cat([idx, idx.new_full((W - (2 * T - 1),), zero_row)])
        parts.append(E[rows].reshape(-1))
        if want_t:
            cols = torch.cat([rows, rows.new_full((ld - W,), zero_row)])
            parts.append(E[cols].t().reshape(-1))
        return torch.cat(parts).to(torch.int32), idx, parts[0].numel()

    flat, idx, n_pack = cached_const(("relpos", dev.index, T, maxlen, bool(want_t)), build)
    zrow = cached_const(("zrow", dev.index, dk), lambda: torch.zeros((1, dk), dtype=torch.float32, device=dev))
    ext = torch.cat([pe.detach(), zrow]).to(torch.bfloat16)
    vals = ext.view(-1).index_select(0, flat)
    window = (vals[n_pack:n_pack + W * dk].view(W, dk), idx)
    if want_t:
        window = window + (vals[n_pack + W * dk:].view(dk, ld),)
    return {"pe_bf": ext[:2 * maxlen], "pack": (vals[:n_pack], int(q_min), int(q_max)), "window": window}


def pe_window_t(win, T):
    """pe_window^T for attn_pos_bwd: bf16 [64, ld] with the window's rows as columns, zero beyond them (ld covers every
    k-step of the last 64-frame block)."""
    W = win.shape[0]
    nblk, nks = (T + 63) // 64, (T + 63 + 31) // 32
    ld = (max(W, (nblk - 1) * 64 + nks * 32) + 7) // 8 * 8
    out = torch.zeros((win.shape[1], ld), dtype=torch.bfloat16, device=win.device)
    out[:, :W] = win.t()
    return out


def slack_bf16(shape, device):
    """A dense bf16 tensor with 16 readable bytes in front of and behind it (csship_attn_pos_bwd_ds reads its rows in
    16-byte pieces at 2-byte alignment)."""
    n = 1
    for d in shape:
        n *= d
    flat = torch.empty(n + 16, dtype=torch.bfloat16, device=device)
    return flat[8:8 + n].view(shape)


def attn_pos_bwd_ds(ds, win_t, q_rows, dq1, dqkv, dwin_acc, n_head):
    """attn_pos_bwd from the ROW-MAJOR dS (bf16 [B,H,T,ldp], allocated by slack_bf16): the kernel skews it while staging,
    no skewed copy exists (csship_attn_pos_bwd_ds)."""
    _need(ds, torch.bfloat16, "ds"), _need(win_t, torch.bfloat16, "win_t"), _need(q_rows, torch.bfloat16, "q_rows")
    _need(dq1, torch.float32, "dq1"), _need(dqkv, torch.bfloat16, "dqkv"), _need(dwin_acc, torch.float32, "dwin_acc")
    B, H, T, ldp = ds.shape
    W = dwin_acc.shape[0]
    st = ds.untyped_storage()
    lo = ds.storage_offset() * 2
    if not (ds.is_contiguous() and win_t.is_contiguous() and q_rows.is_contiguous() and dq1.is_contiguous() and
            dqkv.is_contiguous() and dwin_acc.is_contiguous()) or H != n_head or q_rows.shape != (B * T * H, 64) or \
            dq1.shape != (B, H, T, 64) or dqkv.shape != (B, T, 3 * H * 64) or dwin_acc.shape[1] != 64 or win_t.shape[0] != 64:
        raise RuntimeError("css_amd: attn_pos_bwd_ds expects dense dS [B,H,T,ldp], q rows [B*T*H,64], dq1 [B,H,T,64], dqkv [B,T,3*H*64]")
    if lo < 16 or st.nbytes() - lo - ds.numel() * 2 < 16:
        raise RuntimeError("css_amd: attn_pos_bwd_ds needs 16 bytes of slack on both sides of dS (ops_bwd.slack_bf16)")
    nws = lib().csship_attn_pos_bwd_workspace(B, T)
    ws = _workspace(nws, ds.device)
    check(lib().csship_attn_pos_bwd_ds(_ptr(ds), ldp, _ptr(win_t), win_t.shape[1], _ptr(q_rows), _ptr(dq1), _ptr(dqkv),
                                       _ptr(dwin_acc), W, B, T, H, _ptr(ws), nws, _stream()))
    return dqkv, dwin_acc


def mha_bwd_fused(qkv, pe_pack, lse, d_out, out, n_head, drop=None, dqkv=None):
    """The attention backward up to the window axis in one key-stationary pass (csship_mha_relpos_bwd_fused): recomputes
    the probabilities from q, k, the packed position blocks and the forward's log-sum-exp and returns (dS bf16
    [B,H,T,pad8(T)] - with slack_bf16's slack, the operand of attn_pos_bwd_parts -, dq1 partials f32 [G,B,H,T,64], dqkv bf16
    [B,T,3*H*64] with its k and v thirds written).  No probabilities and no second copy of dS ever reach memory."""
    _need(qkv, torch.bfloat16, "qkv"), _need(lse, torch.float32, "lse"), _need(d_out, torch.bfloat16, "d_out")
    _need(out, torch.bfloat16, "out")
    packed, q_min, q_max = pe_pack
    B, T, D3 = qkv.shape
    if not (qkv.is_contiguous() and d_out.is_contiguous() and lse.is_contiguous() and out.is_contiguous()) \
            or D3 != 3 * n_head * 64 or d_out.numel() != B * T * n_head * 64 or out.numel() != d_out.numel():
        raise RuntimeError("css_amd: mha_bwd_fused expects dense qkv [B,T,3*H*64], d_out / out [B*T,H*64], lse [B,H,T]")
    ldp = (T + 7) // 8 * 8
    ds = slack_bf16((B, n_head, T, ldp), qkv.device)
    G = lib().csship_mha_relpos_bwd_fused_parts(B, T, n_head)
    dq1 = torch.empty(G, B, n_head, T, 64, device=qkv.device, dtype=torch.float32)
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    check(lib().csship_mha_relpos_bwd_fused(_ptr(qkv), _ptr(packed), q_min, q_max, _ptr(lse), _ptr(d_out), _ptr(out), _ptr(ds), ldp,
                                            _ptr(dq1), _ptr(dqkv), B, T, n_head, drop, _stream()))
    return ds, dq1, dqkv


def attn_pos_bwd_parts(ds, win_t, qkv, dq1_parts, dqkv, dwin_acc, n_head):
    """attn_pos_bwd_ds behind mha_bwd_fused: dq1 as 1 or 2 partials [G,B,H,T,64] added in the pass, q rows read in place
    from qkv (csship_attn_pos_bwd_ds_parts)."""
    _need(ds, torch.bfloat16, "ds"), _need(win_t, torch.bfloat16, "win_t"), _need(qkv, torch.bfloat16, "qkv")
    _need(dq1_parts, torch.float32, "dq1_parts"), _need(dqkv, torch.bfloat16, "dqkv"), _need(dwin_acc, torch.float32, "dwin_acc")
    B, H, T, ldp = ds.shape
    W = dwin_acc.shape[0]
    st = ds.untyped_storage()
    lo = ds.storage_offset() * 2
    if not (ds.is_contiguous() and win_t.is_contiguous() and qkv.is_contiguous() and dq1_parts.is_contiguous() and
            dqkv.is_contiguous() and dwin_acc.is_contiguous()) or H != n_head or qkv.shape != (B, T, 3 * H * 64) or \
            tuple(dq1_parts.shape[1:]) != (B, H, T, 64) or dqkv.shape != qkv.shape or dwin_acc.shape[1] != 64 or win_t.shape[0] != 64:
        raise RuntimeError("css_amd: attn_pos_bwd_parts expects dense dS [B,H,T,ldp], qkv / dqkv [B,T,3*H*64], dq1 parts [G,B,H,T,64]")
    if lo < 16 or st.nbytes() - lo - ds.numel() * 2 < 16:
        raise RuntimeError("css_amd: attn_pos_bwd_parts needs 16 bytes of slack on both sides of dS (ops_bwd.slack_bf16)")
    nws = lib().csship_attn_pos_bwd_workspace(B, T)
    ws = _workspace(nws, ds.device)
    check(lib().csship_attn_pos_bwd_ds_parts(_ptr(ds), ldp, _ptr(win_t), win_t.shape[1], _ptr(qkv), 3 * H * 64, _ptr(dq1_parts),
                                             dq1_parts.shape[0], _ptr(dqkv), _ptr(dwin_acc), W, B, T, H, _ptr(ws), nws, _stream()))
    return dqkv, dwin_acc


def attn_pos_bwd(dsk, win_t, q_rows, dq1, dqkv, dwin_acc, n_head):
    """The relative-position term's two gradient products in one pass over the skewed dS (csship_attn_pos_bwd):
    the q third of dqkv = dq1 + dS_skew pe_window, and dwin_acc (f32 [W, 64]) += dS_skew^T q.  dsk bf16 [B,T,H,W]."""
    _need(dsk, torch.bfloat16, "dsk"), _need(win_t, torch.bfloat16, "win_t"), _need(q_rows, torch.bfloat16, "q_rows")
    _need(dq1, torch.float32, "dq1"), _need(dqkv, torch.bfloat16, "dqkv"), _need(dwin_acc, torch.float32, "dwin_acc")
    B, T, H, W = dsk.shape
    if not (dsk.is_contiguous() and win_t.is_contiguous() and q_rows.is_contiguous() and dq1.is_contiguous() and
            dqkv.is_contiguous() and dwin_acc.is_contiguous()) or H != n_head or q_rows.shape != (B * T * H, 64) or \
            dq1.shape != (B, H, T, 64) or dqkv.shape != (B, T, 3 * H * 64) or dwin_acc.shape != (W, 64) or win_t.shape[0] != 64:
        raise RuntimeError("css_amd: attn_pos_bwd expects dense dS_skew [B,T,H,W], q rows [B*T*H,64], dq1 [B,H,T,64], dqkv [B,T,3*H*64]")
    nws = lib().csship_attn_pos_bwd_workspace(B, T)
    ws = _workspace(nws, dsk.device)
    check(lib().csship_attn_pos_bwd(_ptr(dsk), W, _ptr(win_t), win_t.shape[1], _ptr(q_rows), _ptr(dq1), _ptr(dqkv), _ptr(dwin_acc),
                                    W, B, T, H, _ptr(ws), nws, _stream()))
    return dqkv, dwin_acc


def attn_softmax_bwd(qkv, d_out, p, d, skew, n_head, drop=None):
    """dS = P o (dO v^T - D) -> (dS bf16 like p, skew) ; skew: bf16 [B,T,H,W], zero outside the band, filled in place."""
    _need(qkv, torch.bfloat16, "qkv"), _need(p, torch.bfloat16, "p")
    B, T, _ = qkv.shape
    if not (qkv.is_contiguous() and d_out.is_contiguous() and p.is_contiguous() and d.is_contiguous() and skew.is_contiguous()):
        raise RuntimeError("css_amd: attn_softmax_bwd expects dense operands")
    ds = torch.empty_like(p)
    check(lib().csship_attn_softmax_bwd_drop(_ptr(qkv), _ptr(d_out), _ptr(p), p.shape[3], _ptr(d), _ptr(ds), _ptr(skew),
                                             skew.shape[3], B, T, n_head, drop, _stream()))
    return ds, skew
