"""Operator backwards of the Conformer path, composed from the HIP kernels behind ``ops_bwd`` / ``ops``.

The reference gets all of these from torch.autograd over css/models/conformer.py; here each operator has an explicit
backward built from: the batched MFMA GEMM (csship_bgemm: dX = dY.W, dW = dY^T.X, attention products), LayerNorm /
ReLU / softmax backward row kernels, and column sums for the biases.  Nothing in this file computes on the CPU and
nothing falls back to torch arithmetic beyond gathers / scatters of parameter-sized tensors.
"""
import os

import torch

from . import ops, ops_bwd as ob

BF = torch.bfloat16


def linear_bwd(dy, x, w_bf, *, need_dx=True, dx_bf16=False, alpha=1.0, dw_out=None, db_out=None):
    """y = x @ w^T + b.  dy: [M, N] f32/bf16, x: [M, K] f32/bf16, w_bf: bf16 [N, K] -> (dx, dw, db).
    dw_out / db_out: f32 destinations (e.g. the parameters' zeroed .grad views) instead of fresh tensors."""
    dw, db = ob.linear_wgrad(dy, x, alpha, dw_out=dw_out, db_out=db_out)
    dx = ob.bgemm(dy, w_bf, b_kmajor=True, alpha=alpha, out_bf16=dx_bf16) if need_dx else None
    return dx, dw, db


def head_bwd(dmasks, masks, x, w_bf, dw_out=None, db_out=None):
    """masks = relu(x @ w^T + b) (conformer.py:98-99).  dmasks/masks: f32 [M, n_out] (row stride free)."""
    n_out = dmasks.shape[1]
    if w_bf.shape[0] > n_out and w_bf.shape[0] % 8 == 0 and not _NO_HEAD_PAD:
        # w_bf carries zero rows up to a multiple of 8: dz goes to bf16 rows of that width and both products take the
        # whole-group kernels (dw_out / db_out cannot: the product is [rows of w_bf, K])
        dz = ob.relu_bwd_bf16(dmasks, masks, w_bf.shape[0])
        dx, dw, db = linear_bwd(dz, x, w_bf)
        return dx, dw[:n_out], db[:n_out]
    dz = ob.relu_bwd(dmasks, masks)
    return linear_bwd(dz, x, w_bf[:n_out], dw_out=dw_out, db_out=db_out)


def ffn_bwd(x, dy, ln_g, ln_b, b1, w1_bf, w2_bf, alpha=0.5, drop_hidden=None, drop_out=None):
    """out = x + alpha * (relu(LN(x) @ w1^T + b1) @ w2^T + b2)  (conformer.py:322-326 with the half-step residual).

    x, dy: f32 [M, 256].  Returns dx and the parameter gradients (ln_g, ln_b, w1, b1, w2, b2).
    The hidden activations are recomputed (LN + first Linear), not stored by the forward.
    """
    a, _ = ops.layernorm(x, ln_g, ln_b)
    if drop_hidden is not None:   # training with dropout: h is the dropped hidden activation; its positive entries are
        # the survivors of ReLU AND dropout, each carrying 1 / (1 - p); the output branch gradient is dy o mask / (1 - p)
        h = ops.dropout_apply(ops.gemm(a, w1_bf, b1, relu=True), drop_hidden).to(BF)
        dyb = ops.dropout_apply(dy, drop_out, out=torch.empty_like(dy))
        sc = 65536.0 / (65536 - int(drop_hidden.contents.p * 65536.0 + 0.5))
        dw2, db2 = ob.linear_wgrad(dyb, h, alpha)
        dh = ob.bgemm(dyb, w2_bf, b_kmajor=True, alpha=alpha * sc, mask_pos=h, out_bf16=True)
        da, dw1, db1 = linear_bwd(dh, a, w1_bf)
        dx, dg, dbeta = ob.layernorm_bwd(x, da, ln_g, ln_b, dx_add=dy)
        return dx, (dg, dbeta, dw1, db1, dw2, db2)
    h = ops.gemm(a, w1_bf, b1, relu=True, out_bf16=True)
    dw2, db2 = ob.linear_wgrad(dy, h, alpha)
    dh = ob.bgemm(dy, w2_bf, b_kmajor=True, alpha=alpha, mask_pos=h, out_bf16=True)
    da, dw1, db1 = linear_bwd(dh, a, w1_bf)
    dx, dg, dbeta = ob.layernorm_bwd(x, da, ln_g, ln_b, dx_add=dy)
    return dx, (dg, dbeta, dw1, db1, dw2, db2)


def ffn_bwd_fused(x, dy, ln_g, ln_b, b1, w1p, w2tp, w1tp, d_ffn, alpha=0.5, drop_hidden=None, drop_out=None,
                  outs=(None, None, None, None), ln2=None, conv=None):
    """ffn_bwd with the activation side in ONE row-panel kernel (d_model = 256): LN + hidden recompute, masked dh,
    da and the LayerNorm backward never leave the CU; the four weight / bias gradients are GEMMs / column sums over
    the B*T rows.  w1p / w2tp / w1tp: fragment-packed W1, W2^T, W1^T.  ln2 = (x2, gamma2) when x = LayerNorm2(x2): dx
    comes back as the gradient wrt x2 and the gradient tuple ends with that LayerNorm's (dgamma2, dbeta2).  conv: see
    ob.ffn_bwd (the statistics pass of the conv module in front rides in the kernel's epilogue)."""
    res = ob.ffn_bwd(x, dy, ln_g, ln_b, w1p, b1, w2tp, w1tp, d_ffn, alpha, drop_hidden, drop_out, ln2=ln2, conv=conv)
    return _ffn_wgrads(res, dy, alpha, outs)


def _ffn_wgrads(res, dy, alpha, outs):
    """the four weight / bias gradients behind ob.ffn_bwd's (or ob.qkv_ffn_bwd's) activation-side results"""
    dx, dg, dbeta, a, h, dh, dyd = res[:7]
    dw1_o, db1_o, dw2_o, db2_o = outs   # optional f32 destinations (dW1, db1, dW2, db2), e.g. zeroed .grad views
    with ob.paired_reductions():   # the slabs of both products summed by one launch
        if dyd is not None:   # output dropout: the second Linear saw h (dropped) and its gradient is alpha * dy o mask / (1 - p)
            dw2, db2 = ob.linear_wgrad(dyd, h, dw_out=dw2_o, db_out=db2_o)
        else:
            dw2, db2 = ob.linear_wgrad(dy, h, alpha, dw_out=dw2_o, db_out=db2_o)
        dw1, db1 = ob.linear_wgrad(dh, a, dw_out=dw1_o, db_out=db1_o)
    return dx, (dg, dbeta, dw1, db1, dw2, db2) + tuple(res[7:])


_NO_HEAD_PAD = bool(os.environ.get("CSSHIP_NO_HEAD_PAD"))   # A/B timing: the mask head's backward on f32 dz of width n_out
_NO_BAND = bool(os.environ.get("CSSHIP_NO_BAND"))   # A/B timing: the skewed-dS product over the whole window axis
_NO_LN_Y = bool(os.environ.get("CSSHIP_NO_LN_Y"))   # A/B timing: LayerNorm(x) for the q/k/v weight gradient by its own launch
# Round 6: the key-stationary fused pass (csship_mha_relpos_bwd_fused: probabilities recomputed, dS written once, dK / dV in
# registers, no P' in memory).  OPT-IN: it moves 67 MB where the two kernels it replaces move 177, and measures the same -
# 66 against 41 + 29 us alone, 8.21 against 8.16 ms per training step (tools/bench_attn_bwd.py, profiles/r06_attn_fb_ablate.txt):
# with two waves per SIMD both forms are bound by instruction issue, not by bytes.
_ATTN_FUSED_BWD = bool(os.environ.get("CSSHIP_ATTN_FUSED_BWD"))
_SPLIT_ATTN_BWD = bool(os.environ.get("CSSHIP_ATTN_BWD_SPLIT"))   # A/B timing: probabilities / softmax backward / dS k as separate launches
_NO_DVDK = bool(os.environ.get("CSSHIP_NO_DVDK"))   # A/B timing: dV and dK as two general batched products
_NO_POS_FROM_DS = bool(os.environ.get("CSSHIP_NO_POS_FROM_DS"))   # A/B timing: the backward core also writes the skewed dS and the position pass reads that
_NO_POS_FUSED = bool(os.environ.get("CSSHIP_NO_POS_FUSED"))   # A/B timing: the two position products of the attention backward as two tiled products
_NO_PANEL_DX = bool(os.environ.get("CSSHIP_NO_PANEL_DX"))   # A/B timing: the output projection's input gradient by the tiled kernel


def relpos_window(pe_table_bf, T, maxlen):
    """rows of the relative-position table seen by a length-T sequence: window[r] = pe[clamp(r - (T-1)) + maxlen]
    (conformer.py:157-159,179-185), zero padded to a multiple of 8 rows.  Returns (window bf16 [W, d_k], index)."""
    idx = (torch.arange(2 * T - 1, device=pe_table_bf.device) - (T - 1)).clamp(-maxlen, maxlen - 1) + maxlen
    W = (2 * T - 1 + 7) // 8 * 8
    win = torch.zeros((W, pe_table_bf.shape[1]), dtype=BF, device=pe_table_bf.device)
    win[:2 * T - 1] = pe_table_bf[idx]
    return win, idx


def mha_core_bwd(qkv, dao, pe_table_bf, n_head, maxlen=1000, window=None, probs=None, ao=None, skew_buf=None, dwin_acc=None,
                 drop_p=None):
    """Backward of o = softmax(q k^T + q pe[i-j]^T) v (conformer.py:229-253; q already carries 1/sqrt(d_k)).

    qkv: bf16 [B, T, 3*H*64]; dao: bf16 [B*T, H*64] (gradient of the head-concatenated output).
    Returns dqkv bf16 [B, T, 3*H*64] and dpe f32 [2*maxlen, 64].  Scores are re-materialised per (b, h)
    ([B,H,T,T] fits HBM comfortably at these lengths) so every product is one batched GEMM.
    """
    B, T, _ = qkv.shape
    H = n_head
    g = qkv.view(B, T, 3, H, 64)
    q, k, v = (g[:, :, i].permute(0, 2, 1, 3) for i in range(3))  # [B,H,T,64] strided views
    win, idx = relpos_window(pe_table_bf, T, maxlen) if window is None else window[:2]
    W = win.shape[0]
    p_v = None              # operand of dV: the dropped probabilities when the forward dropped them (conformer.py:249)
    do = dao.view(B, T, H, 64).permute(0, 2, 1, 3)
    if (probs is not None and ao is not None and not _SPLIT_ATTN_BWD and _ATTN_FUSED_BWD and window is not None
            and len(window) > 2 and window[2] is not None and T <= ob.pos_bwd_max_T() and not _NO_POS_FUSED and qkv.is_contiguous()
            and dao.is_contiguous()):
        # round 6: ONE key-stationary pass recomputes the probabilities and forms dS (written once), dK, dV (in registers) and
        # dq1 (a partial per key group); the position pass adds the partials and reads q in place.  No P', no second dS.
        ds, parts, dqkv = ob.mha_bwd_fused(qkv, probs[0], probs[1], dao, ao.reshape(B * T, H * 64), H, drop=drop_p)
        acc = dwin_acc if dwin_acc is not None else torch.zeros((window[0].shape[0], 64), dtype=torch.float32, device=qkv.device)
        ob.attn_pos_bwd_parts(ds, window[2], qkv, parts, dqkv, acc, H)
        return dqkv, acc
    if probs is not None and ao is not None and not _SPLIT_ATTN_BWD:
        # one pass over the scores: probabilities, softmax backward (row-major and skewed dS) and dS k
        from_ds = (window is not None and len(window) > 2 and window[2] is not None and T <= ob.pos_bwd_max_T() and not _NO_POS_FUSED
                   and not _NO_POS_FROM_DS)   # the position products skew the row-major dS themselves: no skewed copy
        dsk = None if from_ds else (torch.zeros((B, T, H, W), dtype=BF, device=qkv.device) if skew_buf is None else skew_buf)
        p_v, ds, dsk, dq1, q_rows = ob.mha_bwd_core(qkv, probs[0], probs[1], dao, None, dsk, H, drop=drop_p,
                                                    out=ao.reshape(B * T, H * 64), want_q_rows=True)
        return _mha_core_products(qkv, g, q, k, do, p_v, ds, dsk, dq1, win, idx, window, dwin_acc, maxlen, q_rows, dao=dao)
    if probs is not None:   # (pe_pack, lse) from the training forward: one attention-shaped kernel recomputes P
        if drop_p is not None:
            p, p_v = ops.mha_relpos_probs(qkv, probs[0], probs[1], H, drop=drop_p)
        else:
            p = ops.mha_relpos_probs(qkv, probs[0], probs[1], H)
    else:                   # generic route: materialise the scores and the window products, then a row softmax
        s = ob.bgemm(q, k)                               # q k^T            [B,H,T,T]
        rel = ob.bgemm(q, win[None, None])               # q pe_window^T    [B,H,T,W]
        p = ob.relpos_softmax(s, rel, T)                 # bf16 [B,H,T,Tp]
    if ao is not None:      # attention-shaped softmax backward: no [B,H,T,T] f32 dP buffer
        d = ob.attn_rowdot(dao, ao, B, T, H)
        dsk = torch.zeros((B, T, H, W), dtype=BF, device=qkv.device) if skew_buf is None else skew_buf
        ds, dsk = ob.attn_softmax_bwd(qkv, dao, p, d, dsk, H, drop=drop_p)
    else:                   # generic route
        dp = ob.bgemm(do, v)                             # dO v^T           [B,H,T,T]
        ds, dsk = ob.relpos_softmax_bwd(p, dp, T, W)     # dS, and dS skewed onto the window axis [B,T,H,W]
    dq1 = ob.bgemm(ds[..., :T], k, b_kmajor=True)                       # dS k
    return _mha_core_products(qkv, g, q, k, do, p if p_v is None else p_v, ds, dsk, dq1, win, idx, window, dwin_acc, maxlen,
                              dao=dao)


def _mha_core_products(qkv, g, q, k, do, p_v, ds, dsk, dq1, win, idx, window, dwin_acc, maxlen, q_rows=None, dao=None):
    """the batched GEMMs behind dS: dV = p_v^T dO, dK = dS^T q, dq = dS_skew pe_window + dq1, d(pe_window) = dS_skew^T q"""
    B, T, _ = qkv.shape
    H, W = g.shape[3], win.shape[0]
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (dqkv.view(B, T, 3, H, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    streamed = (not _NO_DVDK and dao is not None and p_v.dim() == 4 and p_v.is_contiguous() and ds.is_contiguous()
                and p_v.shape == ds.shape and p_v.shape[3] % 8 == 0 and dao.is_contiguous())
    if streamed:   # both in one pass over P' and dS (csship_attn_dvdk)
        ob.attn_dvdk(p_v, ds, dao.view(B * T, H * 64), qkv, dqkv, H)
    else:
        ob.bgemm(p_v[..., :T], do, a_kmajor=True, b_kmajor=True, out=dv)    # dV = P^T dO
        ob.bgemm(ds[..., :T], q, a_kmajor=True, b_kmajor=True, out=dk)      # dK = dS^T q
    win_t = window[2] if window is not None and len(window) > 2 else None
    if (win_t is not None and not _NO_POS_FUSED and T <= ob.pos_bwd_max_T() and q_rows is not None and (dsk is None or dsk.is_contiguous())
            and dq1.is_contiguous() and dq1.dtype == torch.float32 and tuple(dq1.shape) == (B, H, T, 64)):
        # both position products in one pass (csship_attn_pos_bwd): over the skewed dS, or - no skewed copy written -
        # over the row-major dS, skewed while the kernel stages its tiles
        acc = dwin_acc if dwin_acc is not None else torch.zeros((W, 64), dtype=torch.float32, device=qkv.device)
        if dsk is None:
            ob.attn_pos_bwd_ds(ds, win_t, q_rows, dq1, dqkv, acc, H)
        else:
            ob.attn_pos_bwd(dsk, win_t, q_rows, dq1, dqkv, acc, H)
        return dqkv, acc
    if dsk is None:
        raise RuntimeError("css_amd: the backward core was asked for no skewed dS, and the fused position pass is off")
    # + dS_skew pe_window; row i of dS_skew is zero outside columns i .. i + T - 1: the product skips those reduction tiles
    ob.bgemm(dsk.permute(0, 2, 1, 3), win[None, None], b_kmajor=True, add=dq1, out=dq, band=0 if _NO_BAND else T - 1)
    if q_rows is None:
        q_rows = g[:, :, 0].reshape(B * T * H, 64)                      # rows ordered (b, t, h) like dsk
    # [W, 64]; dwin_acc (the sum over the blocks already back-propagated) is added by the product's epilogue, in place
    # (rows (b, t, h) of dS_skew are zero outside columns t .. t + T - 1: one K-slab per batch entry, banded)
    banded = not _NO_BAND and (T * H) % 64 == 0 and B > 1
    dwin = ob.bgemm(dsk.view(B * T * H, W), q_rows, a_kmajor=True, b_kmajor=True, add=dwin_acc, out=dwin_acc,
                    **({"band": T - 1, "band_stride": H, "splits": B} if banded else {}))
    if window is not None:
        return dqkv, dwin  # the caller accumulates window gradients over the layers and scatters once
    dpe = torch.zeros((2 * maxlen, 64), dtype=torch.float32, device=qkv.device)
    dpe.index_add_(0, idx, dwin[:2 * T - 1])
    return dqkv, dpe


def mha_bwd(x, dy, qkv, ao, ln_g, ln_b, wqkv_bf, wo_bf, pe_table_bf, n_head, q_scale=0.125, window=None, probs=None,
            skew_buf=None, fused_softmax_bwd=True, drop_p=None, drop_out=None, dwin_acc=None, dyb_bf16=None,
            dwo_out=None, dbo_out=None, wo_tp=None, wqkv_tp=None, ffn_next=None, ffn_below=None):
    """out = x + attention(LN(x)) @ wo^T + bo  (conformer.py:213-254,356).

    wqkv_bf holds the q rows already multiplied by q_scale (as the forward uses them); the returned dwqkv / dbqkv are
    gradients of the UNSCALED parameters.  Returns dx and (ln_g, ln_b, wqkv, bqkv, wo, bo, pe) gradients; with
    ``window`` (from relpos_window) the last one is the gradient of the window rows instead of the whole table.
    ffn_next: keyword arguments of ffn_bwd_fused (without dy) for the FeedForward whose output x is (the block's
    feed_forward_in): when the row-panel forms apply, the q | k | v projection's input-side backward and that FeedForward's
    run as ONE kernel (ob.qkv_ffn_bwd) and the result is (dx of the FeedForward, attention gradients, FeedForward gradients);
    ``chained_ffn(...)`` says beforehand whether they will.  ffn_below (with ffn_next that has ``ln2``): (kwargs of the
    FeedForward BELOW - the next block of the backward pass' feed_forward_out, whose output is ln2's input -, conv tuple of
    ob.ffn_bwd): a third stage of the same kernel (ob.qkv_ffn2_bwd); the result grows by that stage's raw ob.ffn_bwd tuple (its
    weight gradients are the caller's to queue when that block's turn comes).
    """
    D = x.shape[1]
    # training: out = x + D_out(ao wo^T + bo) (conformer.py:254): the branch gradient is dy o mask / (1 - p)
    # dyb_bf16: that product already formed (and rounded to bf16, as the GEMMs round it) by the operator in front
    if dyb_bf16 is not None:
        dyb = dyb_bf16
    else:
        dyb = dy if drop_out is None else ops.dropout_apply(dy, drop_out, out=torch.empty_like(dy))
    if wo_tp is not None and not _NO_PANEL_DX and dyb.dtype == BF and dyb.is_contiguous() and D == 256:
        # dao = dyb Wo as a row-panel Linear against the fragment-packed Wo^T (a 12 800 x 256 x 256 product is all launch,
        # first-tile latency and epilogue for the tiled kernel: 17.5 us against ~11)
        dwo, dbo = ob.linear_wgrad(dyb, ao, dw_out=dwo_out, db_out=dbo_out)
        dao = ops.panel_linear(dyb, wo_tp, None, D, out_bf16=True)
    else:
        dao, dwo, dbo = linear_bwd(dyb, ao, wo_bf, dx_bf16=True, dw_out=dwo_out, db_out=dbo_out)
    B = qkv.shape[0]
    if drop_p is not None and (probs is None or not fused_softmax_bwd):
        raise NotImplementedError("css_amd: dropout of the attention probabilities needs the attention-shaped backward")
    dqkv, dpe = mha_core_bwd(qkv, dao, pe_table_bf, n_head, window=window, probs=probs, dwin_acc=dwin_acc,
                             ao=ao if fused_softmax_bwd else None, skew_buf=skew_buf, drop_p=drop_p)
    dq2 = dqkv.view(-1, 3 * D)
    dw = torch.empty((3 * D, D), dtype=torch.float32, device=x.device)
    db = torch.empty(3 * D, dtype=torch.float32, device=x.device)
    if ffn_next is not None:
        if not chained_ffn(x, wqkv_tp, ffn_next) or not dq2.is_contiguous():
            raise RuntimeError("css_amd: mha_bwd(ffn_next=...) outside the shapes of the chained kernel (ask chained_ffn first)")
        f = dict(ffn_next)
        outs = f.pop("outs", (None, None, None, None))
        if ffn_below is not None:
            fb, conv = ffn_below
            (dg, dbeta, a), res, res_below = ob.qkv_ffn2_bwd(dq2, wqkv_tp, x, dy, ln_g, ln_b, f, fb, conv, f["d_ffn"], f.get("alpha", 0.5))
            ob.linear_wgrad(dq2, a, 1.0, dw_out=dw, db_out=db, head=(D, q_scale))
            _, gf = _ffn_wgrads(res, None, 0.5, outs)
            return res_below[0], (dg, dbeta, dw, db, dwo, dbo, dpe), gf, res_below
        (dg, dbeta, a), res = ob.qkv_ffn_bwd(dq2, wqkv_tp, x, dy, ln_g, ln_b, f.pop("x"), f.pop("ln_g"), f.pop("ln_b"), f.pop("w1p"),
                                            f.pop("b1"), f.pop("w2tp"), f.pop("w1tp"), f.pop("d_ffn"), f.pop("alpha", 0.5),
                                            drop_hidden=f.pop("drop_hidden", None), drop_out=f.pop("drop_out"), ln2=f.pop("ln2", None))
        ob.linear_wgrad(dq2, a, 1.0, dw_out=dw, db_out=db, head=(D, q_scale))
        dxf, gf = _ffn_wgrads(res, None, 0.5, outs)
        return dxf, (dg, dbeta, dw, db, dwo, dbo, dpe), gf
    if wqkv_tp is not None and not _NO_PANEL_DX and D == 256 and x.shape[0] % 64 == 0 and dq2.is_contiguous():
        # da = dqkv Wqkv, the LayerNorm backward, the residual gradient and a = LayerNorm(x) in one row-panel kernel
        dx, dg, dbeta, a = ob.qkv_in_bwd(dq2, wqkv_tp, x, dy, ln_g, ln_b)
        ob.linear_wgrad(dq2, a, 1.0, dw_out=dw, db_out=db, head=(D, q_scale))
        return dx, (dg, dbeta, dw, db, dwo, dbo, dpe)
    da = ob.bgemm(dq2, wqkv_bf, b_kmajor=True)
    # the LayerNorm backward recomputes the normalised rows: it also hands back a = LayerNorm(x), the projection's input
    if D in (256, 512, 1024) and not _NO_LN_Y:
        dx, dg, dbeta, a = ob.layernorm_bwd(x, da, ln_g, ln_b, dx_add=dy, want_y=True)
    else:
        dx, dg, dbeta = ob.layernorm_bwd(x, da, ln_g, ln_b, dx_add=dy)
        a, _ = ops.layernorm(x, ln_g, ln_b)
    ob.linear_wgrad(dq2, a, 1.0, dw_out=dw, db_out=db, head=(D, q_scale))   # q rows carry the projection's 1/sqrt(d_k)
    return dx, (dg, dbeta, dw, db, dwo, dbo, dpe)


_NO_QKV_FFN = bool(os.environ.get("CSSHIP_NO_QKV_FFN"))   # A/B timing: qkv_in_bwd and the FeedForward backward behind it as two launches
# ... and the block below's FeedForward backward (+ its conv module's statistics) as a THIRD stage of that kernel (csship_qkv_ffn2_bwd).
# OPT-IN: measured slower, 7.69-7.81 against 7.64-7.71 ms per training step (profiles/r06_qkv_ffn2_bwd_ab.txt) - 113 us without a
# launch boundary let the 200 workgroups drift out of step, and their weight streams stop sharing the L2.
_QKV_FFN2 = bool(os.environ.get("CSSHIP_QKV_FFN2"))


def chained_ffn(x, wqkv_tp, ffn_next):
    """will mha_bwd(..., ffn_next=ffn_next) run the chained kernel?  (whole 64-row panels of d_model 256, fragment-packed
    Wqkv^T, a FeedForward of the static width with its output dropout on)"""
    return (not _NO_QKV_FFN and not _NO_PANEL_DX and wqkv_tp is not None and x.shape[1] == 256 and x.shape[0] % 64 == 0
            and ffn_next.get("drop_out") is not None and ffn_next["d_ffn"] <= ob.ffn_static_max_ffn()
            and ffn_next["d_ffn"] % 256 == 0)


def embed_bwd(f, dy, mean, rinv, w_bf_pad, bias, ln_g, ln_b, e=None, fn=None):
    """x = relu(LN(MVN(f) @ w^T + b)) (conformer.py:95,124-129).  f: f32 [B,T,idim]; dy: f32 [M, D].
    w_bf_pad: bf16 [D, pad8(idim)].  e / fn: the forward's pre-LN activations / normalised features bf16 [M, Kp], when
    it kept them.  Returns (dw [D, idim], db, dln_g, dln_b); the input needs no gradient."""
    B, T, F = f.shape
    if fn is None:
        fn = ob.mvn_apply(f, mean, rinv, w_bf_pad.shape[1])              # bf16 [M, Kp]
    if e is None:
        e = ob.bgemm(fn, w_bf_pad, bias=bias)                            # pre-LN activations, recomputed
    de, dg, dbeta = ob.layernorm_bwd(e, dy, ln_g, ln_b, relu=True)
    dw, db = ob.linear_wgrad(de, fn)
    return dw[:, :F], db, dg, dbeta
