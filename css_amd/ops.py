"""Torch-tensor front door to libcsship.so.

Each function validates device / dtype / shape on the host (a wrong shape must never reach a
kernel), allocates outputs with torch's caching allocator, and enqueues the kernel on the
current torch HIP stream.  CPU tensors are refused: there is no CPU implementation.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, lib

NBIN = 257
FRAME_LEN, FRAME_HOP = 512, 256
F32_EPS = 1.1920928955078125e-07  # torch.finfo(torch.float32).eps (feature.py:9, conformer.py:21)
NEG_INF = float("-inf")


def _need(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"css_amd: '{name}' must be a CUDA/HIP tensor (no CPU path exists)")
    if t.dtype != dtype:
        raise TypeError(f"css_amd: '{name}' must be {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def num_frames(n_samples):
    return (n_samples - FRAME_LEN) // FRAME_HOP + 1


def stft(x, want=("mag",), logpow_eps=1e-10, pad_rows=False):
    """x: [B,n] or [B,D,n] f32 (last dim contiguous) -> dict of [B,(D,)T,257] tensors (TF layout).  want: any of
    "mag", "re", "im", "pha", "logpow" (= log(max(|X|^2, logpow_eps))).  pad_rows: the outputs are [..., :257] views of
    [..., 260] buffers - 16-byte aligned rows for the consumers that take a row stride (mvn_stats, the models'
    forward / forward_loss, head_pit's targets); anything that needs dense rows gets them from .contiguous()."""
    _need(x, torch.float32, "x")
    if x.dim() not in (2, 3):
        raise RuntimeError(f"STFT expect 2D/3D tensor, but got {x.dim()}D signal")  # feature.py:54-59
    if x.stride(-1) != 1:
        x = x.contiguous()
    if x.dim() == 2:
        n0, s0, n1, s1 = x.shape[0], x.stride(0), 1, 0
    else:
        n0, s0, n1, s1 = x.shape[0], x.stride(0), x.shape[1], x.stride(1)
    n = x.shape[-1]
    if n < FRAME_LEN:
        raise RuntimeError("signal shorter than one frame")
    if x.data_ptr() % 16 or s0 % 4 or s1 % 4:
        x = x.contiguous()
        if x.dim() == 2:
            s0, s1 = x.stride(0), 0
        else:
            s0, s1 = x.stride(0), x.stride(1)
        if s0 % 4 or s1 % 4:  # odd sample count: pad rows to a multiple of 4 floats
            pad = (-n) % 4
            x = torch.nn.functional.pad(x, (0, pad))
            s0 = x.stride(0)
            s1 = x.stride(1) if x.dim() == 3 else 0
    T = num_frames(n)
    pad_rows = bool(pad_rows) and "logpow" not in want
    ld = (NBIN + 3) // 4 * 4 if pad_rows else NBIN
    shape = tuple(x.shape[:-1]) + (T, ld)
    out = {k: torch.empty(shape, device=x.device, dtype=torch.float32) for k in want}
    if pad_rows:
        check(lib().csship_stft_ld(_ptr(x), n0, s0, n1, s1, n, _ptr(out.get("mag")), _ptr(out.get("re")),
                                   _ptr(out.get("im")), _ptr(out.get("pha")), ld, _stream()))
        return {k: v[..., :NBIN] for k, v in out.items()}
    if "logpow" in out:  # log(max(|X|^2, eps)): an epilogue flag of the STFT kernel (training features)
        check(lib().csship_stft_logpower(_ptr(x), n0, s0, n1, s1, n, float(logpow_eps), _ptr(out["logpow"]), _stream()))
        if len(out) == 1:
            return out
    check(lib().csship_stft(_ptr(x), n0, s0, n1, s1, n, _ptr(out.get("mag")), _ptr(out.get("re")),
                            _ptr(out.get("im")), _ptr(out.get("pha")), _stream()))
    return out


MIX_FIELDS = ("u1", "u2", "w0", "delay", "len1", "len2", "rir_on", "d1", "d2", "noise_on", "nu", "n0", "h0", "h1", "h2", "snr_db")


def mix_params(records):
    """list of dicts keyed by MIX_FIELDS -> host uint8 tensor [B, 64] in csship_mix_batch's record layout."""
    import numpy as np
    a = np.zeros((len(records), 16), dtype=np.int32)
    for i, r in enumerate(records):
        a[i, :12] = [int(r[k]) for k in MIX_FIELDS[:12]]
        a[i, 12:] = np.array([r[k] for k in MIX_FIELDS[12:]], dtype=np.float32).view(np.int32)
    return torch.from_numpy(a)


def mix_batch(pool, params_dev, n, noise_pool=None):
    """On-device mixtures: pool f32 [U, L], params_dev int32 [B, 16] (mix_params(...) on the device), window length n
    -> (mix, source1, source2) f32 [B, n].  separation.py:163-267 semantics; see include/csship.h."""
    _need(pool, torch.float32, "pool")
    assert pool.is_contiguous() and params_dev.is_cuda and params_dev.dtype == torch.int32 and params_dev.is_contiguous()
    B = params_dev.shape[0]
    ld = (n + 3) // 4 * 4
    out = torch.empty((3, B, ld), dtype=torch.float32, device=pool.device)
    nws = lib().csship_mix_workspace(B)
    ws = torch.empty(nws, dtype=torch.uint8, device=pool.device)
    npool = noise_pool.contiguous() if noise_pool is not None else None
    check(lib().csship_mix_batch(_ptr(pool), pool.shape[1], pool.shape[0], _ptr(npool), npool.shape[1] if npool is not None else 0,
                                 npool.shape[0] if npool is not None else 0, _ptr(params_dev), B, n, _ptr(out[0]), _ptr(out[1]),
                                 _ptr(out[2]), ld, _ptr(ws), nws, _stream()))
    return out[0, :, :n], out[1, :, :n], out[2, :, :n]


def mvn_stats(x, clamp_min=NEG_INF, eps=F32_EPS):
    """x [B,T,F] (row stride may exceed F) -> (mean [B,F], rinv [B,F]) over the time axis."""
    _need(x, torch.float32, "x")
    B, T, F = x.shape
    if x.stride(2) != 1 or x.stride(0) != T * x.stride(1):
        x = x.contiguous()
    mean = torch.empty(B, F, device=x.device, dtype=torch.float32)
    rinv = torch.empty_like(mean)
    check(lib().csship_mvn_stats(_ptr(x), B, T, F, x.stride(1), clamp_min, eps, _ptr(mean), _ptr(rinv), _stream()))
    return mean, rinv


def transpose_norm(x, mean=None, rinv=None, clamp_min=NEG_INF, out=None, f_off=0):
    """x [B,T,F] -> out[b, f_off+f, t] = (max(x,clamp) - mean) * rinv   (out [B, Ftot, T])."""
    _need(x, torch.float32, "x")
    x = x.contiguous()
    B, T, F = x.shape
    if out is None:
        out = torch.empty(B, F, T, device=x.device, dtype=torch.float32)
    assert out.is_contiguous() and out.shape[0] == B and out.shape[2] == T and out.shape[1] >= f_off + F
    check(lib().csship_transpose_norm(_ptr(x), B, T, F, F, clamp_min, _ptr(mean), _ptr(rinv), _ptr(out),
                                      out.stride(0), f_off, _stream()))
    return out


def layernorm(x, gamma, beta, out_bf16=True, out_f32=False):
    _need(x, torch.float32, "x")
    x = x.contiguous()
    D = x.shape[-1]
    M = x.numel() // D
    ob = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if out_bf16 else None
    of = torch.empty_like(x) if out_f32 else None
    check(lib().csship_layernorm(_ptr(x), M, D, _ptr(_need(gamma, torch.float32, "gamma")),
                                 _ptr(_need(beta, torch.float32, "beta")), _ptr(ob), _ptr(of), _stream()))
    return ob, of


def layernorm_relu(x, gamma, beta, relu=True, out_bf16=False, out_f32=True):
    """LayerNorm over the last dim (any width) followed by an optional ReLU -> (bf16 or None, f32 or None)."""
    _need(x, torch.float32, "x")
    x = x.contiguous()
    D = x.shape[-1]
    M = x.numel() // D
    ob = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if out_bf16 else None
    of = torch.empty_like(x) if out_f32 else None
    check(lib().csship_layernorm_relu(_ptr(x), M, D, _ptr(_need(gamma, torch.float32, "gamma")),
                                      _ptr(_need(beta, torch.float32, "beta")), int(relu), _ptr(ob), _ptr(of), _stream()))
    return ob, of


def pad_weight_bf16(w, scale_rows=None):
    """[N,K] f32 -> bf16 [N, ceil64(K)] zero padded (the W operand layout of csship_gemm)."""
    N, K = w.shape
    Kp = (K + 63) // 64 * 64
    w = w.detach().float()
    if scale_rows is not None:
        w = w * scale_rows[:, None]
    out = torch.zeros(N, Kp, device=w.device, dtype=torch.bfloat16)
    out[:, :K] = w.to(torch.bfloat16)
    return out


def gemm(a, w_bf16, bias=None, *, K=None, relu=False, residual=None, alpha=1.0, ln=None, out_bf16=False,
         mvn=None, rows_per_batch=1, out=None):
    """C = epi(A[M,K] @ W[N,K]^T + bias).  a: bf16 or f32 [M, >=K]; w_bf16: [N, ldw] zero padded."""
    if a.dtype == torch.bfloat16:
        _need(a, torch.bfloat16, "a")
        mode = _lib.A_BF16
    else:
        _need(a, torch.float32, "a")
        mode = _lib.A_F32_MVN if mvn is not None else _lib.A_F32
    _need(w_bf16, torch.bfloat16, "w")
    assert a.dim() == 2 and a.stride(1) == 1 and w_bf16.is_contiguous()
    M = a.shape[0]
    K = a.shape[1] if K is None else K
    N, ldw = w_bf16.shape
    epi = (_lib.EPI_RELU if relu else 0) | (_lib.EPI_RESIDUAL if residual is not None else 0) | \
          (_lib.EPI_LAYERNORM if ln is not None else 0) | (_lib.EPI_OUT_BF16 if out_bf16 else 0)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    if mvn is None and ln is None and K % 8 == 0 and a.stride(0) % 8 == 0 and K * max(M, N) * 4 < 2 ** 31:
        # the batched-GEMM kernel (back-to-back descriptor loads, vector epilogue) is 1.1-2x faster than the classic tiled
        # kernel wherever its loader applies; same arithmetic (bf16 operands, f32 accumulation in the same order)
        q = _lib.BgemmParams(M=M, N=N, K=K, lda=a.stride(0), ldb=ldw, ldc=out.stride(0),
                             ldaux=residual.stride(0) if residual is not None else 0, a_kmajor=0, b_kmajor=0,
                             a_f32=int(a.dtype == torch.float32), b_f32=0,
                             epilogue=(_lib.BEPI_OUT_BF16 if out_bf16 else 0) | (_lib.BEPI_RELU if relu else 0) |
                             ((_lib.BEPI_ADD | _lib.BEPI_SCALE_BIAS) if residual is not None else 0),
                             batch1=1, batch2=1, splits=1, alpha=float(alpha) if residual is not None else 1.0)
        check(lib().csship_bgemm(_ptr(a), _ptr(w_bf16), _ptr(out), _ptr(bias), _ptr(residual), C.byref(q), None, 0, _stream()))
        return out
    p = _lib.GemmParams(M=M, N=N, K=K, lda=a.stride(0), ldw=ldw, ldc=out.stride(0),
                        ldr=residual.stride(0) if residual is not None else N, a_mode=mode, epilogue=epi,
                        rows_per_batch=rows_per_batch, alpha=alpha)
    check(lib().csship_gemm(_ptr(a), _ptr(w_bf16), _ptr(bias), _ptr(residual), _ptr(out),
                            _ptr(mvn[0]) if mvn else None, _ptr(mvn[1]) if mvn else None,
                            _ptr(ln[0]) if ln else None, _ptr(ln[1]) if ln else None, C.byref(p), _stream()))
    return out


def pack_relpos(pe_k, maxlen=1000):
    """pe_k [2*maxlen, 64] -> (packed bf16, q_min, q_max) for csship_mha_relpos (layout: include/csship.h)."""
    assert tuple(pe_k.shape) == (2 * maxlen, 64)
    q_min = -((maxlen + 15) // 16) - 2   # 16*q_min + 15 <= -maxlen: the whole block clamps low
    q_max = (maxlen + 15) // 16 + 1      # 16*q_max >= maxlen - 1: the whole block clamps high
    q = torch.arange(q_min, q_max + 1, device=pe_k.device)
    m = torch.arange(16, device=pe_k.device)
    rel = (16 * q[:, None] + 15 - m[None, :]).clamp(-maxlen, maxlen - 1) + maxlen  # [nq, 16]
    blk = pe_k.detach().float()[rel]                      # [nq, m, 64]
    blk = blk.view(-1, 16, 2, 4, 8).permute(0, 2, 3, 1, 4)  # [nq, ks, g, m, j]
    return blk.contiguous().to(torch.bfloat16).view(-1), int(q_min), int(q_max)


def mha_relpos(qkv, pe_pack, n_head, q_log2e=False):
    """qkv bf16 [B,T,3*H*64] (q pre-scaled by 1/sqrt(d_k); q_log2e: by log2(e)/sqrt(d_k), the inference forward's
    log2-domain form), pe_pack = pack_relpos(pe_k) -> bf16 [B,T,H*64]."""
    _need(qkv, torch.bfloat16, "qkv")
    packed, q_min, q_max = pe_pack
    _need(packed, torch.bfloat16, "pe_packed")
    B, T, D3 = qkv.shape
    assert qkv.is_contiguous() and packed.is_contiguous()
    assert D3 == 3 * n_head * 64 and packed.numel() == (q_max - q_min + 1) * 2 * 512
    out = torch.empty(B, T, n_head * 64, device=qkv.device, dtype=torch.bfloat16)
    check(lib().csship_mha_relpos_ex(_ptr(qkv), _ptr(packed), q_min, q_max, _ptr(out), B, T, n_head,
                                     1 if q_log2e else 0, _stream()))
    return out


def mha_relpos_train(qkv, pe_pack, n_head, drop=None):
    """mha_relpos that also returns the per-row log-sum-exp f32 [B,H,T] (saved for the backward).  drop: dropout cfg of
    the attention probabilities (training)."""
    _need(qkv, torch.bfloat16, "qkv")
    packed, q_min, q_max = pe_pack
    B, T, D3 = qkv.shape
    assert qkv.is_contiguous() and D3 == 3 * n_head * 64
    out = torch.empty(B, T, n_head * 64, device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty(B, n_head, T, device=qkv.device, dtype=torch.float32)
    check(lib().csship_mha_relpos_train_drop(_ptr(qkv), _ptr(packed), q_min, q_max, _ptr(out), _ptr(lse), B, T, n_head,
                                             drop, _stream()))
    return out, lse


def mha_relpos_probs(qkv, pe_pack, lse, n_head, drop=None):
    """softmax probabilities bf16 [B,H,T,pad8(T)] recomputed from q, k, the packed position table and lse.  With a
    dropout cfg returns (p, p_dropped): p_dropped = p o mask / (1 - p_drop), the operand of dV."""
    _need(qkv, torch.bfloat16, "qkv"), _need(lse, torch.float32, "lse")
    packed, q_min, q_max = pe_pack
    B, T, D3 = qkv.shape
    assert qkv.is_contiguous() and lse.is_contiguous() and D3 == 3 * n_head * 64
    ldp = (T + 7) // 8 * 8
    p = torch.empty(B, n_head, T, ldp, device=qkv.device, dtype=torch.bfloat16)
    pd = torch.empty_like(p) if drop is not None else None  # the kernel writes every element p has
    check(lib().csship_mha_relpos_probs_drop(_ptr(qkv), _ptr(packed), q_min, q_max, _ptr(lse), _ptr(p), _ptr(pd), ldp, B,
                                             T, n_head, drop, _stream()))
    return p if drop is None else (p, pd)


def conv_weights_struct(t):
    """t: dict of f32 CUDA tensors keyed like _lib.ConvWeights fields."""
    cw = _lib.ConvWeights()
    for name, typ in _lib.ConvWeights._fields_:
        v = t.get(name)
        if typ is not C.c_void_p:     # "causal": a flag, not a tensor
            setattr(cw, name, int(bool(v)) if v is not None else 0)
        else:
            setattr(cw, name, v.data_ptr() if v is not None else None)
    return cw


def convmod(x, weights, kernel_size=33, batch_stats=False, return_workspace=False, keep_activations=False, drop=None,
            sync=False):
    """x f32 [B,T,D] -> x + ConvModule(x).  weights: dict for conv_weights_struct.
    return_workspace (batch_stats only): also return the workspace the backward needs (pre-BN activations; its last
    4*D floats are scale | shift | batch mean | biased batch variance).
    sync (batch_stats only; SyncBatchNorm, train.py:70): the statistics are those of all ranks' rows - two launches
    with an all-reduce of the per-channel f64 sums in between; every rank must hold the same B*T."""
    _need(x, torch.float32, "x")
    x = x.contiguous()
    B, T, D = x.shape
    y = torch.empty_like(x)
    cw = conv_weights_struct(weights)
    ws = None
    mode = 1 if batch_stats else (2 if keep_activations else 0)  # 2: running statistics, activations kept for the backward
    if mode:
        ws = torch.empty(lib().csship_convmod_workspace(B, T, D), device=x.device, dtype=torch.uint8)
    if sync and batch_stats:
        from .parallel import allreduce_sum_
        import torch.distributed as dist
        check(lib().csship_convmod_stats(_ptr(x), C.byref(cw), B, T, D, kernel_size, _ptr(ws), _stream()))
        off = lib().csship_convmod_sums_offset(B, T, D)
        allreduce_sum_(ws[off:off + 16 * D].view(torch.float64))
        check(lib().csship_convmod_finish(_ptr(x), _ptr(y), C.byref(cw), B, T, D, float(B * T * dist.get_world_size()),
                                          _ptr(ws), drop, _stream()))
        return (y, ws) if return_workspace else y
    check(lib().csship_convmod_drop(_ptr(x), _ptr(y), C.byref(cw), B, T, D, kernel_size, mode,
                                    _ptr(ws), drop, _stream()))
    return (y, ws) if return_workspace else y


def outproj_conv_train(attn_out, wo_packed, bo, x, weights, B, T, kernel_size=33, drop_proj=None, drop_conv=None, apply=True):
    """Training form of the attention output projection + conv module (csship_outproj_conv_train, d_model 256, centred
    taps, batch-statistics BatchNorm): x2 = x + D(attn_out Wo^T + bo), y = x2 + ConvModule_train(x2) in three launches.
    attn_out bf16 [B*T, 256], x f32 [B*T, 256].  Returns (x2, y, workspace): the workspace is the one ops.convmod(...,
    batch_stats=True, return_workspace=True) returns (ops_bwd.convmod_bwd takes it)."""
    _need(attn_out, torch.bfloat16, "attn_out"), _need(x, torch.float32, "x")
    D = x.shape[-1]
    if not (attn_out.is_contiguous() and x.is_contiguous()) or attn_out.shape != (B * T, D) or x.shape != (B * T, D):
        raise RuntimeError("css_amd: outproj_conv_train expects dense [B*T, 256] operands")
    cw = conv_weights_struct(weights)
    # apply=False: the BatchNorm statistics are reduced into the workspace and the module's tail (BatchNorm apply, ReLU, affine,
    # dropout, residual) is left to the chain kernel that consumes the rows (encoder_chain_train(bn=...)); y is None then
    x2, y = torch.empty_like(x), (torch.empty_like(x) if apply else None)
    ws = torch.empty(lib().csship_convmod_workspace(B, T, D), device=x.device, dtype=torch.uint8)
    check(lib().csship_outproj_conv_train(_ptr(attn_out), _ptr(wo_packed), _ptr(bo), _ptr(x), _ptr(x2), _ptr(y), C.byref(cw),
                                          B, T, D, kernel_size, drop_proj, drop_conv if apply else None, _ptr(ws), _stream()))
    return x2, y, ws


def _targets_array(targets, B, T, F, strided=False):
    """HOST array of the targets' device pointers (+ the tensors kept alive).  strided: rows may be padded (stft's
    pad_rows) as long as every target has the same row stride; returns (arr, keep, row stride) then."""
    arr = (C.c_void_p * len(targets))()
    keep = []
    ld = None
    for i, t in enumerate(targets):
        _need(t, torch.float32, f"target{i}")
        if tuple(t.shape) != (B, T, F):
            raise AssertionError(f"{tuple(t.shape)} != {(B, T, F)}")  # mse.py:29 / pit.py:46
        ok = strided and t.stride(2) == 1 and t.stride(0) == T * t.stride(1) and (ld is None or t.stride(1) == ld)
        if not ok:
            t = t.contiguous()
            if strided and ld is not None and ld != F:   # mixed strides: everything dense
                return _targets_array([u.contiguous() for u in targets], B, T, F, strided=True)
        ld = t.stride(1) if strided else F
        keep.append(t)
        arr[i] = t.data_ptr()
    return (arr, keep, ld) if strided else (arr, keep)


_LSTM_WS = {}  # (device index, stream, B, H) -> workspace of the distributed recurrence


def lstm_dist_enabled(H, B=1, backward=False):
    """The distributed recurrence needs all its workgroups resident at once (256 CUs): the forward runs H/32 two-wave
    workgroups per (16 batch rows, direction), two to a CU; the backward H/16 one-wave workgroups holding W_hh^T in
    LDS (64 KB at H = 512: two to a CU; 32 KB at H = 256: four).  Larger batches take the single-CU kernels."""
    if H not in (256, 512) or os.environ.get("CSS_AMD_LSTM_DIST", "1") == "0":
        return False
    groups = 2 * ((B + 15) // 16)
    if backward:
        return groups * (H // 16) <= 256 * (2 if H == 512 else 4)
    return groups * (H // 32) <= 512


def lstm_bidir(gx, whh_packed, B, T, H, keep=False):
    """BLSTM recurrence (both directions).  gx f32 [B*T, 8H] -> y f32 [B, T, 2H]; with keep=True also the activated
    gates [B,T,2,4H] and cell states [B,T,2,H] for the backward pass.  H = 256 / 512 run the distributed kernel (W_hh
    resident in registers across H/32 CUs per batch tile and direction), other sizes the single-CU kernel."""
    _need(gx, torch.float32, "gx")
    dev = gx.device
    y = torch.empty((B, T, 2 * H), dtype=torch.float32, device=dev)
    gates = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=dev) if keep else None
    c = torch.empty((B, T, 2, H), dtype=torch.float32, device=dev) if keep else None
    if lstm_dist_enabled(H, B):
        key = (dev.index, _stream() or 0, B, H)
        ws = _LSTM_WS.get(key)
        if ws is None:
            ws = _LSTM_WS[key] = torch.zeros(lib().csship_lstm_dist_workspace(B, H), dtype=torch.uint8, device=dev)
        check(lib().csship_lstm_bidir_dist(_ptr(gx), _ptr(whh_packed), _ptr(y), _ptr(gates), _ptr(c), B, T, H, _ptr(ws),
                                          ws.numel(), _stream()))
    elif keep:
        check(lib().csship_lstm_bidir_train(_ptr(gx), _ptr(whh_packed), _ptr(y), _ptr(gates), _ptr(c), B, T, H, _stream()))
    else:
        check(lib().csship_lstm_bidir(_ptr(gx), _ptr(whh_packed), _ptr(y), B, T, H, _stream()))
    return (y, gates, c) if keep else y


def lstm_dist_check():
    """Synchronise and raise if any distributed-recurrence launch since the last check timed out on a hand-off.

    The error word is sticky (the last 64 bytes of the workspace, never cleared by a launch): a timeout in ANY layer /
    step since the previous check is seen here.  Called where the host synchronises anyway: the ``loss.item()`` of
    ``train_one_epoch`` / ``validate`` and the end of ``Separator.separate``.  A no-op when the distributed kernels
    were never used."""
    if not _LSTM_WS:
        return
    torch.cuda.synchronize()
    bad = False
    for (idx, _, B, H), ws in _LSTM_WS.items():
        off = lib().csship_lstm_dist_workspace_error_offset(B, H)
        word = ws[off:off + 4].view(torch.int32)
        if int(word.item()) != 0:
            word.zero_()
            bad = True
    if bad:
        raise RuntimeError("css_amd: BLSTM distributed recurrence timed out waiting for a workgroup: outputs and "
                           "gradients since the last check are invalid (are other processes using this GPU?); "
                           "set CSS_AMD_LSTM_DIST=0 to use the single-CU recurrence")


def pit_mse(masks, f_orig, targets, want_y_pred=False, want_pairs=False):
    """Fused mask-apply + PIT-MSE.  masks f32 [B,T,>=K*F]; f_orig f32 [B,T,>=F]; targets: K x [B,T,F].

    Returns (loss 0-dim, perm int32 [B,K], y_pred or None, pair_sse or None).
    """
    _need(masks, torch.float32, "masks")
    _need(f_orig, torch.float32, "f_orig")
    K = len(targets)
    B, T, F = targets[0].shape
    assert masks.stride(2) == 1 and masks.stride(0) == T * masks.stride(1) and masks.shape[:2] == (B, T)
    assert f_orig.stride(2) == 1 and f_orig.stride(0) == T * f_orig.stride(1) and f_orig.shape[:2] == (B, T)
    arr, keep = _targets_array(targets, B, T, F)
    dev = masks.device
    loss = torch.empty((), device=dev, dtype=torch.float32)
    perm = torch.empty(B, K, device=dev, dtype=torch.int32)
    y_pred = torch.empty(B, K, T, F, device=dev, dtype=torch.float32) if want_y_pred else None
    pairs = torch.empty(B, K, K, device=dev, dtype=torch.float32) if want_pairs else None
    ws = torch.empty(lib().csship_pit_workspace(B, T, F, K), device=dev, dtype=torch.uint8)
    check(lib().csship_pit_mse(_ptr(masks), masks.stride(1), _ptr(f_orig), f_orig.stride(1), arr, B, T, F, K,
                               _ptr(y_pred), _ptr(loss), _ptr(perm), _ptr(pairs), _ptr(ws), _stream()))
    return loss, perm, y_pred, pairs


def pack_head_pit(w, b, K, F):
    """Head rows for csship_head_pit: the first K*F rows of the mask head ([n_out, 256] weight, [n_out] bias) reordered as
    (speaker 0 bins 0..255 | speaker 1 bins 0..255 | ... | one 16-row tile: bin 256 of speaker 0 .. K-1, zero padded),
    fragment-packed; returns (w_packed bf16, bias_perm f32)."""
    assert F == 257 and w.shape[1] == 256 and w.shape[0] >= K * F
    idx = torch.cat([torch.arange(k * F, k * F + 256) for k in range(K)] + [torch.tensor([k * F + 256 for k in range(K)])]).to(w.device)
    wp = torch.zeros(K * 256 + 16, 256, device=w.device, dtype=torch.float32)
    bp = torch.zeros(K * 256 + 16, device=w.device, dtype=torch.float32)
    wp[:idx.numel()] = w.detach().float()[idx]
    bp[:idx.numel()] = b.detach().float()[idx]
    return pack_frag(wp), bp.contiguous()


def head_pit(x_enc, w_packed, bias_perm, f_orig, targets, want_pairs=False):
    """csship_head_pit: x_enc f32 [B,T,256] -> (loss 0-dim, perm int32 [B,K], pair_sse or None); masks never materialised."""
    _need(x_enc, torch.float32, "x_enc")
    _need(f_orig, torch.float32, "f_orig")
    K = len(targets)
    B, T, F = targets[0].shape
    assert x_enc.is_contiguous() and f_orig.stride(2) == 1 and f_orig.stride(0) == T * f_orig.stride(1)
    arr, keep, ldt = _targets_array(targets, B, T, F, strided=True)
    dev = x_enc.device
    loss = torch.empty((), device=dev, dtype=torch.float32)
    perm = torch.empty(B, K, device=dev, dtype=torch.int32)
    pairs = torch.empty(B, K, K, device=dev, dtype=torch.float32) if want_pairs else None
    nws = lib().csship_head_pit_workspace(B, T)
    ws = torch.empty(nws, device=dev, dtype=torch.uint8)
    check(lib().csship_head_pit(_ptr(x_enc), _ptr(w_packed), _ptr(bias_perm), _ptr(f_orig), f_orig.stride(1), arr, ldt, B, T, F, K,
                                x_enc.shape[2], _ptr(loss), _ptr(perm), _ptr(pairs), _ptr(ws), nws, _stream()))
    return loss, perm, pairs


def pit_mse_bwd(masks, f_orig, targets, perm, n_masks, grad_loss=1.0, grad_loss_dev=None):
    """grad_loss_dev: optional 0-dim f32 device tensor, a second factor read by the kernel (an autograd backward's
    upstream gradient) - the result equals ``pit_mse_bwd(..., grad_loss) * grad_loss_dev`` bit for bit."""
    K = len(targets)
    B, T, F = targets[0].shape
    arr, keep = _targets_array(targets, B, T, F)
    if grad_loss_dev is not None:
        _need(grad_loss_dev, torch.float32, "grad_loss_dev")
        if grad_loss_dev.numel() != 1:
            raise RuntimeError("css_amd: grad_loss_dev must hold one value")
    # same (possibly padded) row stride as the masks: the kernel addresses both with ld_mask
    grad = torch.empty_strided(masks.shape, masks.stride(), device=masks.device, dtype=masks.dtype)
    check(lib().csship_pit_mse_bwd_dev(_ptr(masks), masks.stride(1), _ptr(f_orig), f_orig.stride(1), arr, _ptr(perm),
                                       B, T, F, K, n_masks, float(grad_loss), _ptr(grad_loss_dev), _ptr(grad), _stream()))
    return grad


def apply_masks(masks, f_orig, K, F):
    """y_pred [B,K,T,F] = masks[..., k*F:(k+1)*F] * f_orig[..., :F]."""
    _need(masks, torch.float32, "masks")
    _need(f_orig, torch.float32, "f_orig")
    B, T = masks.shape[:2]
    y = torch.empty(B, K, T, F, device=masks.device, dtype=torch.float32)
    check(lib().csship_apply_masks(_ptr(masks), masks.stride(1), _ptr(f_orig), f_orig.stride(1), B, T, F, K,
                                   _ptr(y), _stream()))
    return y


def ipd(re_tf, im_tf, index_l, index_r, out, f_off):
    """IPD features from TF-layout spectra [B,C,T,257] into out[b, f_off + p*257 + f, t]."""
    _need(re_tf, torch.float32, "re")
    _need(im_tf, torch.float32, "im")
    B, Cn, T, F = re_tf.shape
    P = len(index_l)
    assert re_tf.is_contiguous() and im_tf.is_contiguous() and out.is_contiguous()
    assert all(0 <= i < Cn for i in list(index_l) + list(index_r))
    assert out.shape[0] == B and out.shape[2] == T and out.shape[1] >= f_off + P * F
    li = (C.c_int32 * P)(*index_l)
    ri = (C.c_int32 * P)(*index_r)
    ws = torch.empty(2 * B * P * F, device=re_tf.device, dtype=torch.float32)
    check(lib().csship_ipd(_ptr(re_tf), _ptr(im_tf), B, Cn, T, li, ri, P, _ptr(out), out.stride(0), f_off,
                           _ptr(ws), _stream()))
    return out


def ipd_tf(re_tf, im_tf, index_l, index_r, out, f_off):
    """IPD features from TF-layout spectra [B,C,T,257] into the frame-major out[b, t, f_off + p*257 + f]."""
    _need(re_tf, torch.float32, "re")
    _need(im_tf, torch.float32, "im")
    B, Cn, T, F = re_tf.shape
    P = len(index_l)
    assert F == 257 and re_tf.is_contiguous() and im_tf.is_contiguous()
    assert all(0 <= i < Cn for i in list(index_l) + list(index_r))
    assert out.dim() == 3 and out.shape[:2] == (B, T) and out.stride(2) == 1 and out.stride(0) == T * out.stride(1)
    assert out.shape[2] >= f_off + P * F
    li = (C.c_int32 * P)(*index_l)
    ri = (C.c_int32 * P)(*index_r)
    nws = lib().csship_ipd_tf_workspace(B, P)
    ws = torch.empty(nws, device=re_tf.device, dtype=torch.uint8)
    check(lib().csship_ipd_tf(_ptr(re_tf), _ptr(im_tf), B, Cn, T, li, ri, P, _ptr(out), out.stride(1), f_off, _ptr(ws), nws,
                              _stream()))
    return out


def pack_frag(w):
    """[R, K] float -> bf16 in MFMA fragment order (include/csship.h, "row-panel kernels").

    Blocks [ceil(R/16)][K/32], block(rt, ks)[lane = 16*g + m][j] = W[16*rt + m][32*ks + 8*g + j].
    """
    R, K = w.shape
    assert K % 32 == 0
    Rp = (R + 15) // 16 * 16
    wp = torch.zeros(Rp, K, device=w.device, dtype=torch.float32)
    wp[:R] = w.detach().float()
    wp = wp.view(Rp // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4)  # [rt, ks, g, m, j]
    return wp.contiguous().to(torch.bfloat16).view(-1)


def il8_row_order(R, device=None):
    """Row order of the 8-feature interleaved packing (CSSHIP_CHAIN_QKV_IL8, include/csship.h): packed row 16*tile + m
    holds feature 32*(tile//2) + 8*(m//4) + 4*(tile%2) + m%4, so that the chain kernel's lane (g, token row) - which
    owns rows 4g..4g+3 of its wave's two tiles - ends up with the 8 consecutive features 32*wave + 8g .. + 7."""
    assert R % 32 == 0
    r = torch.arange(R, device=device)
    tile, m = r // 16, r % 16
    return 32 * (tile // 2) + 8 * (m // 4) + 4 * (tile % 2) + m % 4


def pack_frag_il8(w):
    """pack_frag of the rows of ``w`` [R, K] in il8_row_order (R % 32 == 0); the bias goes through the same order."""
    return pack_frag(w.detach().float()[il8_row_order(w.shape[0], w.device)])


def fold_layernorm(w, b, gamma, beta):
    """Linear(LayerNorm(x)) with the LayerNorm's affine part folded into the Linear (f32):
    (xhat * gamma + beta) W^T + b = xhat (W diag(gamma))^T + (b + W beta) -> (W', b')."""
    w, b = w.detach().float(), b.detach().float()
    return w * gamma.detach().float()[None, :], b + (w * beta.detach().float()[None, :]).sum(1)   # (no BLAS call: pack time only)


def pack_mx(w):
    """W [R, K] f32 -> (e4m3 bytes, e8m0 scale dwords) in the operand layout of csship_ffn_mx (include/csship.h):
    OCP MXFP8 along K (32 consecutive elements per block, shared exponent floor(log2(max|block|)) - 8, round to nearest
    even, saturating at +-448).  R % 16 == 0, K % 256 == 0.
      bytes  uint8 [R/16][K/128][2][64][16]: lane 16*g + m <- W8[16*tile + m][128*ks + 64*half + 16*g + (0..15)]
      scales int32 [R/16][K/256][64]: byte q of lane 16*b + m <- scale of block 4*(2*kp + q) + b of row 16*tile + m"""
    R, K = w.shape
    assert R % 16 == 0 and K % 256 == 0
    x = w.detach().float().reshape(R, K // 32, 32)
    amax = x.abs().amax(-1).clamp_min(2.0 ** -126)
    e = torch.floor(torch.log2(amax)) - 8.0                                   # shared exponent per block
    q = (x / torch.exp2(e)[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).reshape(R, K)
    sb = (e + 127.0).clamp(0, 254).to(torch.int32)                            # e8m0 byte [R, K/32]
    qb = q.view(R // 16, 16, K // 128, 2, 4, 16).permute(0, 2, 3, 4, 1, 5)    # [tile, ks, half, g, m, 16]
    sc = sb.view(R // 16, 16, K // 256, 2, 4).permute(0, 2, 4, 1, 3)          # [tile, kp, b, m, q]
    dw = sc[..., 0] | (sc[..., 1] << 8) | (127 << 16) | (127 << 24)
    return qb.contiguous().view(-1), dw.contiguous().to(torch.int32).view(-1)


def ffn_wide(x, ln, w1p, b1, w2p, b2, d_ffn, alpha=0.5, final_ln=None, out=None):
    """csship_ffn_wide: out = x + alpha * FFN(LN(x)) [-> LN] for x f32 [M, 256 | 512]; w1p / w2p from pack_frag."""
    _need(x, torch.float32, "x")
    assert x.dim() == 2 and x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(lib().csship_ffn_wide(_ptr(x), _ptr(out), _ptr(ln[0]), _ptr(ln[1]), _ptr(w1p), _ptr(b1), _ptr(w2p), _ptr(b2),
                                _ptr(final_ln[0]) if final_ln else None, _ptr(final_ln[1]) if final_ln else None,
                                x.shape[0], x.shape[1], d_ffn, float(alpha), _stream()))
    return out


def ln_linear_wide(x, ln, w, b):
    """csship_ln_linear_wide: bf16(LayerNorm(x) W^T + b) for x f32 [M, 256 | 512], W f32 [N, D] (N % 256 == 0); packs W / b in
    the IL8 order itself (callers on a hot path keep the packed form)."""
    _need(x, torch.float32, "x")
    N, D = w.shape
    order = il8_row_order(N, w.device)
    wp, bp = pack_frag(w.detach().float()[order]), b.detach().float()[order].contiguous()
    out = torch.empty((x.shape[0], N), dtype=torch.bfloat16, device=x.device)
    check(lib().csship_ln_linear_wide(_ptr(x), _ptr(ln[0]), _ptr(ln[1]), _ptr(wp), _ptr(bp), _ptr(out), x.shape[0], D, N, _stream()))
    return out


def linear_res_wide(a_bf16, w, b, res):
    """csship_linear_res_wide: res + a W^T + b for a bf16 [M, D], W f32 [D, D], res f32 [M, D]."""
    _need(a_bf16, torch.bfloat16, "a")
    _need(res, torch.float32, "res")
    out = torch.empty_like(res)
    check(lib().csship_linear_res_wide(_ptr(a_bf16), _ptr(pack_frag(w)), _ptr(b), _ptr(res), _ptr(out), a_bf16.shape[0],
                                       w.shape[0], _stream()))
    return out


def ffn_mx(x, ln, w1mx, b1, w2mx, b2, d_ffn, alpha=0.5, final_ln=None, out=None):
    """csship_ffn_mx: the fused FeedForward with MXFP8 operands; w1mx / w2mx = pack_mx(W) tuples (bytes, scales)."""
    _need(x, torch.float32, "x")
    assert x.dim() == 2 and x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(lib().csship_ffn_mx(_ptr(x), _ptr(out), _ptr(ln[0]), _ptr(ln[1]), _ptr(w1mx[0]), _ptr(w1mx[1]), _ptr(b1),
                              _ptr(w2mx[0]), _ptr(w2mx[1]), _ptr(b2), _ptr(final_ln[0]) if final_ln else None,
                              _ptr(final_ln[1]) if final_ln else None, x.shape[0], x.shape[1], d_ffn, float(alpha),
                              _stream()))
    return out


def ffn_fused(x, ln, w1p, b1, w2p, b2, d_ffn, alpha=0.5, final_ln=None, out=None, drop_hidden=None, drop_out=None,
              want_pre_ln=False):
    """out = x + alpha * FFN(LN(x)) [-> LN]; x f32 [M,256]; w1p/w2p from pack_frag.  drop_hidden / drop_out: dropout cfg
    pointers (css_amd.dropout.DropState.cfg) of the training forward, None = no dropout.  want_pre_ln (with final_ln):
    returns (rows in front of the final LayerNorm, out) - the training forward keeps both."""
    _need(x, torch.float32, "x")
    assert x.dim() == 2 and x.is_contiguous()
    M, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    pre = torch.empty_like(x) if want_pre_ln else None
    check(lib().csship_ffn_fused_train(_ptr(x), _ptr(out), _ptr(pre), _ptr(ln[0]), _ptr(ln[1]), _ptr(w1p), _ptr(b1), _ptr(w2p),
                                       _ptr(b2), _ptr(final_ln[0]) if final_ln else None,
                                       _ptr(final_ln[1]) if final_ln else None, M, D, d_ffn, alpha, drop_hidden, drop_out,
                                       _stream()))
    return (pre, out) if want_pre_ln else out


def dropout_apply(x, cfg, out=None):
    """y = x o mask / (1 - p) on an f32 [rows, cols] tensor (row stride free, cols % 4 == 0); in place by default."""
    _need(x, torch.float32, "x")
    x2 = x.view(-1, x.shape[-1]) if x.is_contiguous() else x
    assert x2.dim() == 2 and x2.stride(1) == 1
    y = x2 if out is None else out.view(-1, out.shape[-1])
    check(lib().csship_dropout_apply(_ptr(x2), x2.stride(0), _ptr(y), y.stride(0), x2.shape[0], x2.shape[1], cfg, _stream()))
    return x if out is None else out


def dropout_add(x, y, cfg, alpha=1.0, out=None):
    """out = x + alpha * dropout(y), dense f32 [rows, cols]."""
    _need(x, torch.float32, "x"), _need(y, torch.float32, "y")
    assert x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib().csship_dropout_add(_ptr(x), _ptr(y), _ptr(out), float(alpha), x.numel() // x.shape[-1], x.shape[-1], cfg,
                                   _stream()))
    return out


def dropout_mask(n, cfg, device="cuda"):
    """The keep mask itself (uint8 [n]) - tests compare it with oracle.dropout.keep_mask."""
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    check(lib().csship_dropout_mask(_ptr(keep), n, cfg, _stream()))
    return keep


def panel_linear(a, wp, bias, N, *, ln=None, relu=False, residual=None, alpha=1.0, out_bf16=False, out=None, drop=None):
    """out[M,N] = epi(pro(a)[M,256] @ W^T + bias); a: f32 (LN if ln given, else rounded) or bf16 [M,256].
    drop (with residual, f32 output): out = residual + alpha * dropout(a W^T + bias), masks as dropout_add's."""
    assert a.dim() == 2 and a.is_contiguous() and a.shape[1] == 256
    M = a.shape[0]
    if a.dtype == torch.bfloat16:
        pro = _lib.PRO_BF16
    else:
        _need(a, torch.float32, "a")
        pro = _lib.PRO_LN if ln is not None else _lib.PRO_F32
    epi = (_lib.EPI_RELU if relu else 0) | (_lib.EPI_RESIDUAL if residual is not None else 0) | \
          (_lib.EPI_OUT_BF16 if out_bf16 else 0)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    check(lib().csship_panel_linear_drop(_ptr(a), pro, _ptr(ln[0]) if ln else None, _ptr(ln[1]) if ln else None,
                                         _ptr(wp), _ptr(bias), _ptr(residual),
                                         residual.stride(0) if residual is not None else 0, alpha, _ptr(out),
                                         out.stride(0), M, N, 256, epi, drop, _stream()))
    return out


def istft(re, im, length=None):
    """re/im f32 [S, 257, T] (the reference's [B, F, T] layout) -> wav [S, length]; torch.istft(center=True, hann)."""
    _need(re, torch.float32, "re")
    _need(im, torch.float32, "im")
    re, im = re.contiguous(), im.contiguous()
    S, F, T = re.shape
    assert F == NBIN and im.shape == re.shape
    if length is None:
        length = FRAME_HOP * (T - 1)
    wav = torch.empty(S, length, device=re.device, dtype=torch.float32)
    check(lib().csship_istft(_ptr(re), _ptr(im), S, T, length, _ptr(wav), _stream()))
    return wav


def _ffn_struct(ln, w1p, b1, w2p, b2):
    w = _lib.FfnWeights()
    w.ln_g, w.ln_b = (ln[0].data_ptr(), ln[1].data_ptr()) if ln is not None else (None, None)
    w.w1, w.b1, w.w2, w.b2 = (t.data_ptr() for t in (w1p, b1, w2p, b2))
    return w


def embed_train(f, mean, rinv, w_packed, bias, ln_g, ln_b, drop=None):
    """Training form of the embedding (csship_embed_train): x0 = dropout(ReLU(LayerNorm(Linear(MVN_time(f))))) from f f32
    [B,T,idim], one kernel; also returns what the backward keeps.  w_packed: fragment-packed [256, Kp] with Kp = idim rounded
    up to 256, zero padded.  -> (x0 f32 [B*T,256], fn bf16 [B*T, pad8(idim)], e f32 [B*T,256])."""
    _need(f, torch.float32, "f")
    B, T, F = f.shape
    if f.stride(2) != 1 or f.stride(0) != T * f.stride(1):
        f = f.contiguous()
    M, ldfn = B * T, (F + 7) // 8 * 8
    out = torch.empty((M, 256), dtype=torch.float32, device=f.device)
    e = torch.empty_like(out)
    fn = torch.empty((M, ldfn), dtype=torch.bfloat16, device=f.device)
    check(lib().csship_embed_train(_ptr(f), f.stride(1), _ptr(mean), _ptr(rinv), _ptr(w_packed), _ptr(bias), _ptr(ln_g), _ptr(ln_b),
                                   _ptr(out), _ptr(fn), ldfn, _ptr(e), B, T, F, 256, drop, _stream()))
    return out, fn, e


def encoder_chain_train(x, d_ffn, ffn_b, qkv, ffn_a=None, final_ln=None, drops=(None, None, None, None), bn=None):
    """csship_encoder_chain_train on x f32 [M,256] (the training forward through the chain kernel).  ffn_a / ffn_b:
    (ln=(g,b), w1p, b1, w2p, b2); final_ln: (g, b) behind ffn_a; qkv: (ln=(g,b), wqkv_packed, bqkv, n_qkv); drops: dropout
    cfg pointers (a.hidden, a.out, b.hidden, b.out).  Returns (x_out, qkv_bf16, pre_ln_rows, ln_rows) - the last two
    None without ffn_a.
    bn = (x2, workspace, weights, drop_conv) behind outproj_conv_train(..., apply=False): x is ignored, the kernel forms its
    input rows x3 = x2 + dropout(pw2(relu(BatchNorm(z)))) itself (whole 64-row panels only) and a fifth result, x3, is
    returned."""
    x3 = None
    if bn is not None:
        x2, ws, cwd, drop_conv = bn
        _need(x2, torch.float32, "x2")
        x = x2
    _need(x, torch.float32, "x")
    assert x.dim() == 2 and x.is_contiguous() and x.shape[1] == 256
    M = x.shape[0]
    out = torch.empty_like(x)
    fa = _ffn_struct(*ffn_a) if ffn_a is not None else None
    fb = _ffn_struct(*ffn_b)
    pre = torch.empty_like(x) if ffn_a is not None else None
    lnr = torch.empty_like(x) if ffn_a is not None else None
    n_qkv = qkv[3]
    q_out = torch.empty(M, n_qkv, device=x.device, dtype=torch.bfloat16)
    tr = _lib.ChainTrain()
    tr.drop_a_hidden, tr.drop_a_out, tr.drop_b_hidden, tr.drop_b_out = (d if d is not None else None for d in drops)
    tr.pre_ln_out, tr.ln_out = (pre.data_ptr() if pre is not None else None), (lnr.data_ptr() if lnr is not None else None)
    if bn is not None:
        wsf = ws.view(torch.float32)
        x3 = torch.empty_like(x)
        tr.bn_z, tr.bn_x2, tr.bn_scale_shift = wsf.data_ptr(), x2.data_ptr(), wsf[-4 * 256:].data_ptr()
        tr.bn_pw2_w, tr.bn_pw2_b = _need(cwd["pw2_w"], torch.float32, "pw2_w").data_ptr(), _need(cwd["pw2_b"], torch.float32, "pw2_b").data_ptr()
        tr.drop_bn, tr.bn_out = drop_conv, x3.data_ptr()
    check(lib().csship_encoder_chain_train(
        _ptr(x), _ptr(out), C.byref(fa) if fa is not None else None,
        _ptr(final_ln[0]) if final_ln else None, _ptr(final_ln[1]) if final_ln else None, C.byref(fb),
        _ptr(qkv[0][0]), _ptr(qkv[0][1]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(q_out), M, 256, d_ffn, n_qkv, C.byref(tr), _stream()))
    return (out, q_out, pre, lnr, x3) if bn is not None else (out, q_out, pre, lnr)


def encoder_chain(x, d_ffn, ffn_a=None, final_ln=None, ffn_b=None, qkv=None, out=None, flags=0):
    """csship_encoder_chain on x f32 [M,256].  ffn_a / ffn_b: (ln=(g,b) or None, w1p, b1, w2p, b2) with packed weights
    (ln None: the affine part is folded into w1p / b1, fold_layernorm); final_ln: (g, b) applied after ffn_a; qkv:
    (ln=(g,b) or None, wqkv_packed, bqkv, n_qkv); flags: _lib.CHAIN_QKV_IL8.  Returns (x_out, qkv_bf16|None)."""
    _need(x, torch.float32, "x")
    assert x.dim() == 2 and x.is_contiguous() and x.shape[1] == 256
    M = x.shape[0]
    if out is None:
        out = torch.empty_like(x)
    fa = _ffn_struct(*ffn_a) if ffn_a is not None else None
    fb = _ffn_struct(*ffn_b) if ffn_b is not None else None
    q_out, n_qkv = None, 0
    if qkv is not None:
        n_qkv = qkv[3]
        q_out = torch.empty(M, n_qkv, device=x.device, dtype=torch.bfloat16)
    check(lib().csship_encoder_chain(
        _ptr(x), _ptr(out), None, None, 1, C.byref(fa) if fa is not None else None,
        _ptr(final_ln[0]) if final_ln else None, _ptr(final_ln[1]) if final_ln else None,
        C.byref(fb) if fb is not None else None, _ptr(qkv[0][0]) if qkv and qkv[0] else None,
        _ptr(qkv[0][1]) if qkv and qkv[0] else None,
        _ptr(qkv[1]) if qkv else None, _ptr(qkv[2]) if qkv else None, _ptr(q_out), M, 256, d_ffn, n_qkv, int(flags), _stream()))
    return out, q_out
