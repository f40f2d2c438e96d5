"""Autograd glue for training.

Forward values ALWAYS come from the HIP kernels.  Backward:
  * PIT-MSE -> masks: hand-written kernel (csship_pit_mse_bwd);
  * masks -> parameters, Conformer with d_model = 256 (``ConformerTrain``): explicit HIP backward of every operator
    (css_amd/backward.py: batched MFMA GEMMs, LayerNorm / softmax / ConvModule backward kernels).  The training
    forward runs operator by operator (same kernels as inference, minus the cross-operator chaining) and keeps each
    operator's input; hidden activations are recomputed in the backward;
  * masks -> parameters, BLSTM (``BlstmTrain``): back-propagation-through-time kernel for the recurrence, the same
    GEMM / LayerNorm backward kernels for the rest;
  * anything else (Conformer widths other than 256 / 512, BLSTM hidden sizes the BPTT kernel does not tile) raises
    NotImplementedError: there is no torch-op arithmetic path in this package (the fp32 autograd references the tests
    compare against live in oracle/ and tests/).
"""
import os

import torch

from . import backward as bw, ops, ops_bwd as ob
from .dropout import DropState, model_drop_state

BF = torch.bfloat16
_NO_TRAIN_CHAIN = bool(os.environ.get("CSSHIP_NO_TRAIN_CHAIN"))   # A/B timing: the training forward operator by operator
_NO_FFN_LN = bool(os.environ.get("CSSHIP_NO_FFN_LN"))       # A/B timing: the block's final LayerNorm as its own launch
_NO_CONV_NEXT = bool(os.environ.get("CSSHIP_NO_CONV_NEXT"))  # A/B timing: dropout(dx) for the output projection's backward as its own pass
_NO_DROP_EPI = bool(os.environ.get("CSSHIP_NO_DROP_EPI"))   # A/B timing: output-projection dropout as its own pass
_NO_LN2 = bool(os.environ.get("CSSHIP_NO_LN2_FUSE"))   # A/B timing: every block's final LayerNorm backward as its own launch
_NO_EMBED_TRAIN = bool(os.environ.get("CSSHIP_NO_EMBED_TRAIN"))   # A/B timing: the training embedding as four launches
_NO_BN_IN_CHAIN = bool(os.environ.get("CSSHIP_NO_BN_IN_CHAIN"))   # A/B timing: the conv module's BatchNorm-apply pass as its own launch
_NO_OC_TRAIN = bool(os.environ.get("CSSHIP_NO_OC_TRAIN"))   # A/B timing: output projection and conv module's first kernel as two launches


_NO_DIRECT = bool(os.environ.get("CSS_AMD_NO_DIRECT_GRAD"))   # A/B timing: every gradient through a temporary + the multi-tensor add


def _direct_grad(p):
    """The preset ``.grad`` view of ``p`` if a gradient may be STORED into it (FusedAdam owns it and it has not been written
    since the optimiser's last zero_grad(): 0 + g == g), else None.  Claims the view: a second contribution accumulates."""
    g = p.grad
    own = getattr(g, "_csship_owner", None) if g is not None else None
    if (_NO_DIRECT or own is None or not p.requires_grad or id(p) in own._dirty or g.dtype != torch.float32
            or not g.is_contiguous() or not own.grads_untouched()):
        return None
    own._dirty.add(id(p))
    return g


def _mark_written(ps):
    for p in ps:
        own = getattr(p.grad, "_csship_owner", None)
        if own is not None:
            own._dirty.add(id(p))


def _owners(ps):
    """the FusedAdam instances that own the ``.grad`` views of ps, each with whether its buffer was still untouched by
    torch-side writes (checked BEFORE one of our own follows: FusedAdam.own_write re-arms the check afterwards)"""
    seen = {}
    for p in ps:
        own = getattr(p.grad, "_csship_owner", None) if p.grad is not None else None
        if own is not None and id(own) not in seen:
            seen[id(own)] = (own, own.grads_untouched())
    return list(seen.values())


# Experiment (round 6): a block's grouped weight-gradient launch + parameter reductions on a second stream (one fork per block,
# one join at the end of the backward pass) beside the next block's data-gradient chain.  Single-process steps only.
_NO_FFN_CONV = bool(os.environ.get("CSSHIP_NO_FFN_CONV"))   # A/B timing: the conv module's statistics pass as its own launch
_WGRAD_SIDE = bool(os.environ.get("CSSHIP_WGRAD_SIDE"))
# Experiment (round 6): only the small parameter-gradient REDUCTIONS (block flush, conv tap partials, window sums) on the side stream
_REDUCE_SIDE = bool(os.environ.get("CSSHIP_REDUCE_SIDE"))
_side_streams = {}


def _side_stream(device):
    key = str(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _deliver(model, G):
    """Hand the parameter gradients G to autograd.  Parameters whose ``.grad`` already exists (FusedAdam presets them
    as views of its flat gradient buffer) are accumulated in place by ONE multi-tensor add and get None back, instead
    of one accumulate launch per parameter (~550 for Conformer-base); the rest go back through autograd as usual."""
    params = list(model.parameters())
    direct = [p for p in params if p.requires_grad and p in G and p.grad is not None
              and p.grad.dtype == G[p].dtype and p.grad.is_contiguous()]
    if direct:
        _accumulate(direct, G)
        _mark_written(direct)
        for p in direct:
            del G[p]
    # (a leftover may still be the transposed [D, K] view put() keeps for csship_grad_accumulate: autograd wants p.shape)
    return tuple((G[p].reshape(p.shape) if p in G else None) if p.requires_grad else None for p in params)


_NO_GRAD_ACC = bool(os.environ.get("CSS_AMD_NO_GRAD_ACC"))   # A/B timing: torch._foreach_add_ (+ contiguous copies) instead


def _accumulate(ps, G):
    """p.grad += G[p] for the parameters ps: one csship_grad_accumulate launch per 96 of them (transposed 2-d views
    included), torch._foreach_add_ for whatever layout that does not take."""
    with torch.no_grad():
        pairs = [(p.grad, G[p]) for p in ps]
        rest = pairs if _NO_GRAD_ACC else ob.grad_accumulate(pairs)
        if rest:
            owners = _owners(ps)
            torch._foreach_add_([d for d, _ in rest], [s.reshape(d.shape).contiguous() for d, s in rest])
            for own, clean in owners:   # our own torch-side write: it must not read as a foreign one
                own.own_write(clean)


def _deliver_group(model, module, G):
    """Gradients of ``module``'s parameters are final while the backward pass is still running: add them into their
    preset ``.grad`` views now and tell the trainer's bucket hook (css_amd.parallel.OverlappedGradReducer starts the
    all-reduce of that stretch of the flat gradient buffer while the remaining layers are back-propagated)."""
    hook = getattr(model, "_grad_bucket_hook", None)
    if hook is None:
        return
    ps = [p for p in module.parameters() if p.requires_grad and p in G and p.grad is not None
          and p.grad.dtype == G[p].dtype and p.grad.is_contiguous()]
    if ps:
        _accumulate(ps, G)
        _mark_written(ps)
        for p in ps:
            del G[p]
    # the hook may start an in-place all-reduce on a stretch of the flat gradient buffer (a torch-side write of ours)
    owners = _owners([p for p in module.parameters() if p.grad is not None][:1])
    hook(module)
    for own, clean in owners:
        own.own_write(clean)


def _skew_buffer(model, shape, device):
    """The skewed-dS buffer [B, T, H, W] of the attention backward: the kernels write only inside the band (the same
    positions every layer, every step), everything else must be zero - so it is zeroed ONCE per shape and kept on the
    model (it was an 82 MB fill per training step).  A step that is being captured must NOT take it from the graph's
    pool: the pool hands one block to several tensors of a replay in turn, and whoever held the block earlier in the
    step fills it with its own data on every replay.  It takes the buffer an eager step of the same shape left on the
    model (GraphedTrainStep's warm-up steps do) and pins it through ``model._skew_captured``, the list the capturer
    installs and keeps; without either, a captured step fills a fresh buffer every replay, as before."""
    key = (tuple(shape), str(device))
    ent = getattr(model, "_skew_cache", None)
    if torch.cuda.is_current_stream_capturing():
        keep = getattr(model, "_skew_captured", None)
        if keep is None or ent is None or ent[0] != key:
            return torch.zeros(shape, dtype=BF, device=device)
        keep.append(ent[1])
        return ent[1]
    if ent is None or ent[0] != key:
        ent = (key, torch.zeros(shape, dtype=BF, device=device))
        model._skew_cache = ent
    return ent[1]


class PitMse(torch.autograd.Function):
    """loss = mean_b min_perm MSE(mask_k * f_orig, target_{perm})  with the hand-written backward."""

    @staticmethod
    def forward(ctx, masks, f_orig, n_masks, *targets):
        loss, perm, _, _ = ops.pit_mse(masks, f_orig, list(targets))
        ctx.save_for_backward(masks, f_orig, perm, *targets)
        ctx.n_masks = n_masks
        ctx.mark_non_differentiable(perm)
        return loss, perm

    @staticmethod
    def backward(ctx, grad_loss, _grad_perm):
        masks, f_orig, perm, *targets = ctx.saved_tensors
        # the upstream gradient is a device scalar the kernel reads itself (a `g.mul_(grad_loss)` behind it was a second
        # pass over 39 MB); the product keeps the padded row stride of the masks, so the encoder's backward takes it
        # as it is
        if grad_loss.dtype == torch.float32 and grad_loss.numel() == 1 and grad_loss.is_cuda:
            g = ops.pit_mse_bwd(masks, f_orig, targets, perm, ctx.n_masks, 1.0, grad_loss_dev=grad_loss)
        else:
            g = ops.pit_mse_bwd(masks, f_orig, targets, perm, ctx.n_masks, 1.0).mul_(grad_loss)
        return (g,) + (None,) * (2 + len(targets))


def _conformer_pack_plan(model, wide, device):
    """bf16 shadows (row-major / MFMA fragment order / fragment-packed transposes) of every Linear weight of the model,
    produced by ONE launch per step.  Built once per (model, kernel family, device); the destinations are rewritten in
    place by every forward."""
    ent = getattr(model, "_train_pack", None)
    if ent is not None and ent["wide"] == wide and ent["device"] == device:
        return ent
    enc = model.conformer
    D = enc.d_model
    plan = ob.PackPlan(device)
    ent = {"wide": wide, "device": device, "plan": plan, "layers": []}
    we = enc.embed[0].weight.detach()
    ent["we_p"] = None
    if not wide:   # + the fragment table of the fused training embedding (csship_embed_train): K zero-padded to a multiple of 256
        kp = (we.shape[1] + 255) // 256 * 256
        ent["we_p"] = torch.zeros(we.shape[0] * kp, dtype=BF, device=device)
        ent["we_bf"] = plan.add(we, want_row_major=True)[0]
        plan.add(we, frag=ent["we_p"], k0=0, k_total=kp)
    else:
        ent["we_bf"] = plan.add(we, want_row_major=True)[0]
    wh = model.linear.weight.detach()   # row-major shadow with zero rows up to a multiple of 8 (bw.head_bwd)
    wh_rm = torch.zeros(((wh.shape[0] + 7) // 8 * 8, (wh.shape[1] + 7) // 8 * 8), dtype=torch.bfloat16, device=device)
    ent["wh_bf"], ent["wh_p"] = plan.add(wh, row_major=wh_rm, want_frag=not wide)
    for lay in enc.encoders:
        h = {}
        for tag, m in (("ffn_in", lay.feed_forward_in), ("ffn_out", lay.feed_forward_out)):
            w1, w2 = m.net[0].weight.detach(), m.net[3].weight.detach()
            if wide:
                h[tag] = (plan.add(w1, want_row_major=True)[0], plan.add(w2, want_row_major=True)[0])
            else:  # fragment-packed weights and, for the fused backward kernel, their transposes
                h[tag] = (plan.add(w1, want_frag=True)[1], plan.add(w2, want_frag=True)[1],
                          plan.add(w2, transposed=True, want_frag=True)[1], plan.add(w1, transposed=True, want_frag=True)[1])
        a = lay.self_attn
        wqkv_bf = torch.empty((3 * D, D), dtype=BF, device=device)
        wqkv_p = None if wide else torch.empty(3 * D * D, dtype=BF, device=device)
        for i, lin in enumerate((a.linear_q, a.linear_k, a.linear_v)):
            plan.add(lin.weight.detach(), scale=a.d_k ** -0.5 if i == 0 else 1.0, row_major=wqkv_bf, frag=wqkv_p,
                     row0=i * D)
        h["wqkv_bf"], h["wqkv_p"] = wqkv_bf, wqkv_p
        # fragment-packed (scaled) Wqkv^T [D, 3D]: the projection's input gradient as a row-panel kernel (d_model 256)
        wqkv_tp = None if wide else torch.empty(3 * D * D, dtype=BF, device=device)
        if not wide:
            for i, lin in enumerate((a.linear_q, a.linear_k, a.linear_v)):
                plan.add(lin.weight.detach(), scale=a.d_k ** -0.5 if i == 0 else 1.0, transposed=True, frag=wqkv_tp,
                         k0=i * D, k_total=3 * D)
        h["wqkv_tp"] = wqkv_tp
        h["wo_bf"], h["wo_p"] = plan.add(a.linear_out.weight.detach(), want_row_major=True, want_frag=not wide)
        # fragment-packed Wo^T: the input gradient of the output projection as a row-panel Linear (d_model 256)
        h["wo_tp"] = None if wide else plan.add(a.linear_out.weight.detach(), transposed=True, want_frag=True)[1]
        ent["layers"].append(h)
    model._train_pack = ent
    return ent


def bn_counters(model):
    """ONE int64 tensor behind the `num_batches_tracked` buffers of all conv modules (the buffers become views of it, same
    state_dict keys and values): the per-step increment is one launch instead of one per BatchNorm (torch's foreach ops
    have no multi-tensor path for int64).  Rebuilt when the module was moved or its buffers were replaced."""
    bns = [lay.conv.BN for lay in model.conformer.encoders]
    flat = getattr(model, "_bn_cnt_flat", None)
    if flat is None or flat.device != bns[0].num_batches_tracked.device or \
            any(b.num_batches_tracked._base is not flat for b in bns):
        with torch.no_grad():
            flat = torch.stack([b.num_batches_tracked for b in bns])
        for i, b in enumerate(bns):
            b._buffers["num_batches_tracked"] = flat[i]
        model._bn_cnt_flat = flat
    return flat


class ConformerTrain(torch.autograd.Function):
    """masks = Conformer(f) with the explicit HIP backward.  model.training selects BatchNorm batch statistics (running
    statistics updated) and the reference's dropouts (css_amd/dropout.py) or, in eval mode, the frozen running
    statistics and no dropout.  d_k = 64, kernel 33;
    d_model = 256 runs the tuned row-panel kernels (fused FeedForward forward / backward, LN-prologue projections),
    other widths (Conformer-large: 512) the same operators from the size-generic kernels."""

    @staticmethod
    def forward(ctx, model, f, *params):
        enc = model.conformer
        D, H, FF, K = enc.d_model, enc.n_head, enc.d_ffn, enc.kernel_size
        B, T, F = f.shape
        M = B * T
        f = f.contiguous()
        wide = D != 256          # size-generic kernels
        batch_bn = model.training
        from .parallel import is_distributed
        sync_bn = bool(batch_bn and getattr(model, "sync_batchnorm", False) and is_distributed())
        if sync_bn and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("SyncBatchNorm all-reduces inside the step: use the eager step, not GraphedTrainStep")
        M_bn = f.shape[0] * f.shape[1] * (torch.distributed.get_world_size() if sync_bn else 1)
        mean, rinv = ops.mvn_stats(f)
        # dropout (training only): the step's seed is snapshotted so that the backward regenerates the same masks even if
        # another forward runs in between
        ds = model_drop_state(model, getattr(enc, "dropout_rate", 0.0))
        if ds.active:
            ds = DropState(ds.seed.clone(), ds.p, ds.offset)
        S = {"f": f, "mean": mean, "rinv": rinv, "layers": [], "wide": wide, "batch_bn": batch_bn, "drop": ds, "sync_bn": sync_bn}
        Kp = (F + 7) // 8 * 8
        pack = _conformer_pack_plan(model, wide, f.device)
        pack["plan"].run()
        we_bf = pack["we_bf"]
        if pack["we_p"] is not None and not _NO_EMBED_TRAIN:   # MVN + Linear + LayerNorm + ReLU + dropout: one kernel (round 6)
            x, fn, e = ops.embed_train(f, mean, rinv, pack["we_p"], enc.embed[0].bias.detach(), enc.embed[1].weight.detach(),
                                       enc.embed[1].bias.detach(), drop=ds.cfg("embed"))
        else:
            fn = ob.mvn_apply(f, mean, rinv, Kp)
            e = ob.bgemm(fn, we_bf, bias=enc.embed[0].bias.detach())
            _, x = ops.layernorm_relu(e, enc.embed[1].weight.detach(), enc.embed[1].bias.detach())
            if ds.active:  # conformer.py:127 drops before the ReLU; relu(drop(v)) == drop(relu(v)) (the scale is positive)
                ops.dropout_apply(x, ds.cfg("embed"))
        S["we_bf"], S["fn"] = we_bf, fn   # (the normalised features, 6.7 MB in bf16, kept for the embed weight gradient)
        pe = model.position_table()
        # packed table of the forward, bf16 table, window (+ its transpose for csship_attn_pos_bwd up to 449 frames) of
        # the backward: three launches
        rt = ob.relpos_tables(pe, T, 1000, want_t=T <= ob.pos_bwd_max_T())
        pe_pack = rt["pack"]
        S["pe_bf"], S["pe_pack"], S["window"] = rt["pe_bf"], pe_pack, rt["window"]
        bn_means, bn_vars, bn_new_m, bn_new_v = [], [], [], []
        # small per-block operands for ALL blocks in a handful of launches (was 3 tiny launches per block): the fused
        # q | k | v bias with the q part pre-scaled, and the depthwise taps as [k][D] (lanes read consecutive channels)
        lays = list(enc.encoders)
        qs = lays[0].self_attn.d_k ** -0.5
        bqkv_all = torch.stack([getattr(l_.self_attn, n_).bias.detach() for l_ in lays
                                for n_ in ("linear_q", "linear_k", "linear_v")]).view(len(lays), 3 * D)
        bqkv_all.mul_(ob.cached_const(("qscale", f.device.index, D, qs), lambda: torch.cat(
            [torch.full((D,), qs, dtype=torch.float32, device=f.device), torch.ones(2 * D, dtype=torch.float32, device=f.device)])))
        dww_all = torch.stack([l_.conv.dw_conv_1d.weight.detach().view(D, K) for l_ in lays]).transpose(1, 2).contiguous()
        chained = not wide and not _NO_TRAIN_CHAIN
        carried = None

        def chain_ffn(i, tag):
            m = getattr(lays[i], "feed_forward_in" if tag == "ffn_in" else "feed_forward_out")
            w1p, w2p = pack["layers"][i][tag][0], pack["layers"][i][tag][1]
            return ((m.layer_norm.weight.detach(), m.layer_norm.bias.detach()), w1p, m.net[0].bias.detach(), w2p,
                    m.net[3].bias.detach())

        def chain_qkv(i):
            a_ = lays[i].self_attn
            return ((a_.layer_norm.weight.detach(), a_.layer_norm.bias.detach()), pack["layers"][i]["wqkv_p"], bqkv_all[i], 3 * D)

        def chain_drops(i, tag):
            return (ds.cfg(f"{i}.{tag}.hidden"), ds.cfg(f"{i}.{tag}.out"))

        for li, (lay, wp) in enumerate(zip(enc.encoders, pack["layers"])):
            L = {}

            def ffn(m, xin, tag, final_ln=None):
                dh, do = ds.cfg(f"{li}.{tag}.hidden"), ds.cfg(f"{li}.{tag}.out")
                if wide:  # LayerNorm -> tiled GEMM (ReLU, bf16) -> tiled GEMM (+ residual)
                    w1_bf, w2_bf = wp[tag]
                    L[tag] = (xin, w1_bf, w2_bf)
                    a_, _ = ops.layernorm(xin, m.layer_norm.weight.detach(), m.layer_norm.bias.detach())
                    if not ds.active:
                        h_ = ops.gemm(a_, w1_bf, m.net[0].bias.detach(), relu=True, out_bf16=True)
                        return ops.gemm(h_, w2_bf, m.net[3].bias.detach(), residual=xin, alpha=0.5)
                    h_ = ops.dropout_apply(ops.gemm(a_, w1_bf, m.net[0].bias.detach(), relu=True), dh)   # f32 hidden
                    return ops.dropout_add(xin, ops.gemm(h_, w2_bf, m.net[3].bias.detach()), do, alpha=0.5)
                w1p, w2p, w2tp, w1tp = wp[tag]
                L[tag] = (xin, w1p, w2tp, w1tp)
                return ops.ffn_fused(xin, (m.layer_norm.weight.detach(), m.layer_norm.bias.detach()), w1p,
                                     m.net[0].bias.detach(), w2p, m.net[3].bias.detach(), FF, 0.5, drop_hidden=dh, drop_out=do,
                                     final_ln=final_ln, want_pre_ln=final_ln is not None)

            a = lay.self_attn
            scale = a.d_k ** -0.5
            wqkv_bf, wqkv_p, wo_bf, wo_p = wp["wqkv_bf"], wp["wqkv_p"], wp["wo_bf"], wp["wo_p"]
            bqkv = bqkv_all[li]
            if chained and li == 0:   # first FeedForward + LayerNorm + q | k | v of the first block: one launch
                x1, qkv, _, _ = ops.encoder_chain_train(x, FF, chain_ffn(li, "ffn_in"), chain_qkv(li),
                                                        drops=(None, None) + chain_drops(li, "ffn_in"))
                L["ffn_in"] = (x,) + tuple(wp["ffn_in"][i] for i in (0, 2, 3))
            elif chained:             # came out of the chained launch at the end of the previous block
                x1, qkv = carried
                L["ffn_in"] = (x,) + tuple(wp["ffn_in"][i] for i in (0, 2, 3))
            else:
                x1 = ffn(lay.feed_forward_in, x, "ffn_in")
                if wide:
                    a2, _ = ops.layernorm(x1, a.layer_norm.weight.detach(), a.layer_norm.bias.detach())
                    qkv = ops.gemm(a2, wqkv_bf, bqkv, out_bf16=True)
                else:
                    qkv = ops.panel_linear(x1, wqkv_p, bqkv, 3 * D,
                                           ln=(a.layer_norm.weight.detach(), a.layer_norm.bias.detach()), out_bf16=True)
            ao, lse = ops.mha_relpos_train(qkv.view(B, T, 3 * D), pe_pack, H, drop=ds.cfg(f"{li}.attn.p"))
            ao = ao.view(M, D)
            c = lay.conv
            fused_oc = (batch_bn and not wide and not sync_bn and not _NO_OC_TRAIN and K == 33 and not getattr(c, "causal", False))
            if fused_oc:
                pass   # output projection + dropout + residual inside the conv module's first kernel (below)
            elif ds.active and not wide and not _NO_DROP_EPI:   # x2 = x1 + D(ao wo^T + bo): the mask in the projection's epilogue
                x2 = ops.panel_linear(ao, wo_p, a.linear_out.bias.detach(), D, residual=x1, drop=ds.cfg(f"{li}.attn.out"))
            elif ds.active:   # the projection without its fused residual, then one add pass
                y2 = ops.gemm(ao, wo_bf, a.linear_out.bias.detach()) if wide else \
                    ops.panel_linear(ao, wo_p, a.linear_out.bias.detach(), D)
                x2 = ops.dropout_add(x1, y2, ds.cfg(f"{li}.attn.out"))
            elif wide:
                x2 = ops.gemm(ao, wo_bf, a.linear_out.bias.detach(), residual=x1, alpha=1.0)
            else:
                x2 = ops.panel_linear(ao, wo_p, a.linear_out.bias.detach(), D, residual=x1)
            L["mha"] = (x1, qkv, ao, wqkv_bf, wo_bf, scale, lse)
            L["wo_tp"], L["wqkv_tp"] = wp.get("wo_tp"), wp.get("wqkv_tp")
            W = {"ln_g": c.layer_norm.weight.detach(), "ln_b": c.layer_norm.bias.detach(),
                 "pw1_w": c.pw_conv_1.weight.detach().view(2), "pw1_b": c.pw_conv_1.bias.detach(),
                 "dw_w": dww_all[li], "dw_b": c.dw_conv_1d.bias.detach(),
                 "bn_g": c.BN.weight.detach(), "bn_b": c.BN.bias.detach(),
                 "bn_mean": c.BN.running_mean, "bn_var": c.BN.running_var,
                 "pw2_w": c.pw_conv_2.weight.detach().view(1), "pw2_b": c.pw_conv_2.bias.detach(),
                 "causal": getattr(c, "causal", False)}
            # round 6: behind the fused kernel the module's tail (BatchNorm apply, ReLU, affine, dropout, residual) is formed by
            # the chain kernel that consumes the rows - its prologue reads z and x2 - instead of a launch of its own
            bn_in_chain = (fused_oc and chained and li + 1 < len(lays) and M % 64 == 0 and FF <= ob.ffn_static_max_ffn()
                           and not _NO_BN_IN_CHAIN)
            if fused_oc:   # x2 = x1 + D(ao wo^T + bo) and the conv module with batch statistics: three launches instead of five
                x2, x3, fws = ops.outproj_conv_train(ao, wo_p, a.linear_out.bias.detach(), x1, W, B, T, K,
                                                     drop_proj=ds.cfg(f"{li}.attn.out"), drop_conv=ds.cfg(f"{li}.conv.out"),
                                                     apply=not bn_in_chain)
            else:
                x3, fws = ops.convmod(x2.view(B, T, D), W, K, batch_stats=batch_bn, keep_activations=not batch_bn,
                                      return_workspace=True, drop=ds.cfg(f"{li}.conv.out"), sync=sync_bn)
                x3 = x3.view(M, D)
            L["conv"] = (x2, W, fws)
            if batch_bn:
                stats = fws.view(torch.float32)[-4 * D:]
                bn_means.append(c.BN.running_mean), bn_vars.append(c.BN.running_var)
                bn_new_m.append(stats[2 * D:3 * D]), bn_new_v.append(stats[3 * D:])
            if chained and li + 1 < len(lays):
                # this block's last FeedForward + its LayerNorm, the next block's first FeedForward and its LayerNorm +
                # q | k | v as ONE launch (the inference forward's chain kernel with the dropouts and the saved rows)
                if bn_in_chain:
                    x1n, qkvn, x4, x, x3 = ops.encoder_chain_train(
                        None, FF, chain_ffn(li + 1, "ffn_in"), chain_qkv(li + 1), ffn_a=chain_ffn(li, "ffn_out"),
                        final_ln=(lay.layer_norm.weight.detach(), lay.layer_norm.bias.detach()),
                        drops=chain_drops(li, "ffn_out") + chain_drops(li + 1, "ffn_in"),
                        bn=(x2, fws, W, ds.cfg(f"{li}.conv.out")))
                else:
                    x1n, qkvn, x4, x = ops.encoder_chain_train(
                        x3, FF, chain_ffn(li + 1, "ffn_in"), chain_qkv(li + 1), ffn_a=chain_ffn(li, "ffn_out"),
                        final_ln=(lay.layer_norm.weight.detach(), lay.layer_norm.bias.detach()),
                        drops=chain_drops(li, "ffn_out") + chain_drops(li + 1, "ffn_in"))
                carried = (x1n, qkvn)
                L["ffn_out"] = (x3,) + tuple(wp["ffn_out"][i] for i in (0, 2, 3))
            elif wide or _NO_FFN_LN:
                x4 = ffn(lay.feed_forward_out, x3, "ffn_out")
                _, x = ops.layernorm(x4, lay.layer_norm.weight.detach(), lay.layer_norm.bias.detach(), out_bf16=False,
                                     out_f32=True)
            else:   # the block's LayerNorm in the FeedForward's row epilogue; the backward needs the rows in front of it too
                x4, x = ffn(lay.feed_forward_out, x3, "ffn_out",
                            final_ln=(lay.layer_norm.weight.detach(), lay.layer_norm.bias.detach()))
            L["x4"] = x4
            S["layers"].append(L)
        if bn_means:  # BatchNorm1d bookkeeping for all blocks at once (momentum 0.1, unbiased running variance)
            with torch.no_grad():
                ob.ema_update(bn_means, bn_new_m, 0.1, bn_vars, bn_new_v, 0.1 * M_bn / max(M_bn - 1, 1), 0.9)
                bn_counters(model).add_(1)
        n_out = model.linear.out_features
        wh_bf, wh_p = pack["wh_bf"], pack["wh_p"]
        ld = (n_out + 3) // 4 * 4
        masks = torch.empty((M, ld), dtype=torch.float32, device=f.device)[:, :n_out]
        if wide:
            ob.bgemm(x, wh_bf[:n_out], bias=model.linear.bias.detach(), out=masks)
            torch.relu_(masks)
        else:
            ops.panel_linear(x, wh_p, model.linear.bias.detach(), n_out, relu=True, out=masks)
        S.update(e=e, x_enc=x, masks=masks, wh_bf=wh_bf)
        ctx.model, ctx.S = model, S
        return masks.view(B, T, n_out)

    @staticmethod
    def backward(ctx, grad_masks):
        model, S = ctx.model, ctx.S
        ctx.S = None  # release the saved activations with this call
        enc = model.conformer
        D, H = enc.d_model, enc.n_head
        B, T, F = S["f"].shape
        M = B * T
        G = {}  # parameter -> gradient

        def put(p, g):
            if g is p.grad:   # stored in place by its producer (_direct_grad)
                return
            # (a transposed 2-d view - the depthwise conv weight, kept [taps, channels] by the kernels - stays a view:
            # csship_grad_accumulate transposes while it adds; reshape() would copy it)
            if not (g.dim() == 2 and g.stride(0) == 1 and g.numel() == p.numel() and p not in G and p.grad is not None):
                g = g.reshape(p.shape)
            if p in G:
                red.flush()   # the sum reads both gradients
                red.join()
                red.rside.join()
                G[p] = G[p].reshape(p.shape) + g
            else:
                G[p] = g

        # the reductions behind the parameter gradients of a block run as ONE launch at its end (ops_bwd.deferred_reductions)
        side = (_side_stream(grad_masks.device) if _REDUCE_SIDE and not S["wide"] and getattr(model, "_grad_bucket_hook", None) is None
                else None)
        with ob.reduce_side_stream(side) as rside:
            with ob.deferred_reductions(group_wgrads=not S["wide"]) as red:
                red.rside = rside
                return ConformerTrain._backward(ctx, grad_masks, model, S, G, put, red)

    @staticmethod
    def _backward(ctx, grad_masks, model, S, G, put, red):
        enc = model.conformer
        D, H = enc.d_model, enc.n_head
        B, T, F = S["f"].shape
        M = B * T
        masks = S["masks"]
        dm = grad_masks.reshape(M, -1)
        if dm.stride(1) != 1 or dm.stride(0) % 4:
            buf = torch.empty_strided(masks.shape, masks.stride(), dtype=torch.float32, device=masks.device)
            buf.copy_(dm)
            dm = buf
        padded = S["wh_bf"].shape[0] != masks.shape[1] and not bw._NO_HEAD_PAD   # the products come back wider than the parameters
        dx, dw, db = bw.head_bwd(dm, masks, S["x_enc"], S["wh_bf"],
                                 dw_out=None if padded else _direct_grad(model.linear.weight),
                                 db_out=None if padded else _direct_grad(model.linear.bias))
        put(model.linear.weight, dw), put(model.linear.bias, db)
        red.flush()
        _deliver_group(model, model.linear, G)
        window = S["window"]   # (window, its rows in the table[, window^T]): ob.relpos_tables in the forward
        dwin = None
        # skewed dS buffer shared by all layers: the band positions are the same every layer, the rest stays zero
        # (not needed - and its 82 MB not allocated - when the position pass skews the row-major dS itself)
        from_ds = len(window) > 2 and not bw._NO_POS_FUSED and not bw._NO_POS_FROM_DS
        skew_buf = None if from_ds else _skew_buffer(model, (B, T, H, window[0].shape[0]), masks.device)
        ds = S["drop"]
        nlay = len(enc.encoders)
        layers = list(enc.encoders)
        side = _side_stream(masks.device) if _WGRAD_SIDE and not S["wide"] else None
        ln_done = False   # this block's final LayerNorm was back-propagated by the block above (ffn_bwd's second LayerNorm)
        pending = None    # (ffn_bwd tuple of this block's feed_forward_out, its conv module's gradient buffer) from the block above
        for li, lay, L in zip(range(nlay - 1, -1, -1), reversed(layers), reversed(S["layers"])):
            if not ln_done:
                dx, dg, dbeta = ob.layernorm_bwd(L["x4"], dx, lay.layer_norm.weight.detach(), lay.layer_norm.bias.detach())
                put(lay.layer_norm.weight, dg), put(lay.layer_norm.bias, dbeta)

            def ffn_b(m, tag, dy, ln2=None, conv=None):
                dh, do = ds.cfg(f"{li}.{tag}.hidden"), ds.cfg(f"{li}.{tag}.out")
                if S["wide"]:
                    xin, w1_bf, w2_bf = L[tag]
                    dxo, g = bw.ffn_bwd(xin, dy, m.layer_norm.weight.detach(), m.layer_norm.bias.detach(),
                                        m.net[0].bias.detach(), w1_bf, w2_bf, 0.5, drop_hidden=dh, drop_out=do)
                else:
                    xin, w1p, w2tp, w1tp = L[tag]
                    # the four big gradients of the FeedForward go straight into their (zeroed) .grad views
                    outs = tuple(_direct_grad(q) for q in (m.net[0].weight, m.net[0].bias, m.net[3].weight, m.net[3].bias))
                    dxo, g = bw.ffn_bwd_fused(xin, dy, m.layer_norm.weight.detach(), m.layer_norm.bias.detach(),
                                              m.net[0].bias.detach(), w1p, w2tp, w1tp, enc.d_ffn, 0.5,
                                              drop_hidden=dh, drop_out=do, outs=outs, ln2=ln2, conv=conv)
                for p, gv in zip((m.layer_norm.weight, m.layer_norm.bias, m.net[0].weight, m.net[0].bias,
                                  m.net[3].weight, m.net[3].bias), g):
                    put(p, gv)
                return (dxo, g[6:]) if ln2 is not None else dxo

            x2, W, fws = L["conv"]
            c = lay.conv
            # the statistics pass of the conv module's backward rides in the epilogue of the FeedForward backward in front of it
            cflat = None
            if pending is not None:
                # this block's feed_forward_out already ran as the third stage of the kernel the block above launched: its
                # weight gradients are queued now, with this block's group
                res_below, cflat = pending
                pending = None
                m = lay.feed_forward_out
                outs = tuple(_direct_grad(q) for q in (m.net[0].weight, m.net[0].bias, m.net[3].weight, m.net[3].bias))
                dx, g = bw._ffn_wgrads(res_below, None, 0.5, outs)
                for p, gv in zip((m.layer_norm.weight, m.layer_norm.bias, m.net[0].weight, m.net[0].bias,
                                  m.net[3].weight, m.net[3].bias), g):
                    put(p, gv)
            elif (not S["wide"] and not S["sync_bn"] and not _NO_CONV_NEXT and not _NO_FFN_CONV and D == 256 and M % 64 == 0
                    and enc.d_ffn <= ob.ffn_static_max_ffn() and enc.d_ffn % 256 == 0 and enc.kernel_size == 33):
                cflat = torch.empty(38 * D + 8, dtype=torch.float32, device=dx.device)
                dx = ffn_b(lay.feed_forward_out, "ffn_out", dx, conv=(fws, W["pw2_w"], ds.cfg(f"{li}.conv.out"), cflat, B, T))
            else:
                dx = ffn_b(lay.feed_forward_out, "ffn_out", dx)
            dyb = None
            if cflat is not None:
                dx3, g, dyb = ob.convmod_bwd(x2.view(B, T, D), dx.view(B, T, D), W, fws, enc.kernel_size,
                                             bn_eval=not S["batch_bn"], drop=ds.cfg(f"{li}.conv.out"), want_next=True,
                                             next_drop=ds.cfg(f"{li}.attn.out"), flat=cflat, stats_done=True)
            elif S["sync_bn"] or _NO_CONV_NEXT:
                dx3, g = ob.convmod_bwd(x2.view(B, T, D), dx.view(B, T, D), W, fws, enc.kernel_size,
                                        bn_eval=not S["batch_bn"], drop=ds.cfg(f"{li}.conv.out"), sync=S["sync_bn"])
            else:   # also the gradient through the attention output's dropout, as the bf16 operand of the two products behind it
                dx3, g, dyb = ob.convmod_bwd(x2.view(B, T, D), dx.view(B, T, D), W, fws, enc.kernel_size,
                                             bn_eval=not S["batch_bn"], drop=ds.cfg(f"{li}.conv.out"), want_next=True,
                                             next_drop=ds.cfg(f"{li}.attn.out"))
            dx = dx3.view(M, D)
            put(c.layer_norm.weight, g["ln_g"]), put(c.layer_norm.bias, g["ln_b"])
            put(c.pw_conv_1.weight, g["pw1_w"]), put(c.pw_conv_1.bias, g["pw1_b"])
            put(c.dw_conv_1d.weight, g["dw_w"].t()), put(c.dw_conv_1d.bias, g["dw_b"])
            put(c.BN.weight, g["bn_g"]), put(c.BN.bias, g["bn_b"])
            put(c.pw_conv_2.weight, g["pw2_w"]), put(c.pw_conv_2.bias, g["pw2_b"])
            x1, qkv, ao, wqkv_bf, wo_bf, scale, lse = L["mha"]
            a = lay.self_attn
            # the block's input is the final LayerNorm of the block below: its backward rides in the FeedForward backward's epilogue
            ln_done = li > 0 and not S["wide"] and not _NO_LN2 and M % 64 == 0 and enc.d_ffn <= ob.ffn_static_max_ffn()
            ln2 = (S["layers"][li - 1]["x4"], layers[li - 1].layer_norm.weight.detach()) if ln_done else None
            # ... and the FeedForward backward itself rides behind the q | k | v projection's in one kernel where it can
            ffn_next = None
            if not S["wide"]:
                fm = lay.feed_forward_in
                xin, w1p, w2tp, w1tp = L["ffn_in"]
                ffn_next = dict(x=xin, ln_g=fm.layer_norm.weight.detach(), ln_b=fm.layer_norm.bias.detach(), b1=fm.net[0].bias.detach(),
                                w1p=w1p, w2tp=w2tp, w1tp=w1tp, d_ffn=enc.d_ffn, alpha=0.5, drop_hidden=ds.cfg(f"{li}.ffn_in.hidden"),
                                drop_out=ds.cfg(f"{li}.ffn_in.out"), ln2=ln2)
                if bw.chained_ffn(x1, L.get("wqkv_tp"), ffn_next):
                    ffn_next["outs"] = tuple(_direct_grad(q) for q in (fm.net[0].weight, fm.net[0].bias, fm.net[3].weight, fm.net[3].bias))
                else:
                    ffn_next = None
            # ... and the feed_forward_out of the block BELOW (its output is that LayerNorm's input) as a third stage, with the
            # statistics pass of that block's conv module in its epilogue
            ffn_below = None
            if (ffn_next is not None and ln_done and bw._QKV_FFN2 and not S["sync_bn"] and not _NO_CONV_NEXT and not _NO_FFN_CONV
                    and enc.kernel_size == 33 and ds.cfg(f"{li - 1}.ffn_out.out") is not None):
                bm, Lb = layers[li - 1].feed_forward_out, S["layers"][li - 1]
                xb, w1pb, w2tpb, w1tpb = Lb["ffn_out"]
                cflat_b = torch.empty(38 * D + 8, dtype=torch.float32, device=dx.device)
                ffn_below = (dict(x=xb, ln_g=bm.layer_norm.weight.detach(), ln_b=bm.layer_norm.bias.detach(), b1=bm.net[0].bias.detach(),
                                  w1p=w1pb, w2tp=w2tpb, w1tp=w1tpb, drop_hidden=ds.cfg(f"{li - 1}.ffn_out.hidden"),
                                  drop_out=ds.cfg(f"{li - 1}.ffn_out.out")),
                             (Lb["conv"][2], Lb["conv"][1]["pw2_w"], ds.cfg(f"{li - 1}.conv.out"), cflat_b, B, T))
            res = bw.mha_bwd(x1, dx, qkv.view(B, T, 3 * D), ao, a.layer_norm.weight.detach(),
                             a.layer_norm.bias.detach(), wqkv_bf, wo_bf, S["pe_bf"], H, scale, window=window,
                             probs=(S["pe_pack"], lse), skew_buf=skew_buf, drop_p=ds.cfg(f"{li}.attn.p"),
                             drop_out=ds.cfg(f"{li}.attn.out"), dwin_acc=dwin, dyb_bf16=dyb,
                             dwo_out=_direct_grad(a.linear_out.weight), dbo_out=_direct_grad(a.linear_out.bias),
                             wo_tp=L.get("wo_tp"), wqkv_tp=L.get("wqkv_tp"), ffn_next=ffn_next, ffn_below=ffn_below)
            dx, g = res[:2]
            if ffn_below is not None:
                pending = (res[3], ffn_below[1][3])
            dg, dbeta, dwqkv, dbqkv, dwo, dbo, dwin = g   # dwin: the window gradient summed over the blocks so far
            put(a.layer_norm.weight, dg), put(a.layer_norm.bias, dbeta)
            for i, lin in enumerate((a.linear_q, a.linear_k, a.linear_v)):
                put(lin.weight, dwqkv[i * D:(i + 1) * D]), put(lin.bias, dbqkv[i * D:(i + 1) * D])
            put(a.linear_out.weight, dwo), put(a.linear_out.bias, dbo)
            if ffn_next is not None:   # (dx is already the FeedForward's input gradient)
                fm, gf = lay.feed_forward_in, res[2]
                for p, gv in zip((fm.layer_norm.weight, fm.layer_norm.bias, fm.net[0].weight, fm.net[0].bias,
                                  fm.net[3].weight, fm.net[3].bias), gf):
                    put(p, gv)
                if ln_done:
                    below = layers[li - 1]
                    put(below.layer_norm.weight, gf[6]), put(below.layer_norm.bias, gf[7])
            elif ln_done:
                below = layers[li - 1]
                dx, (dg2, dbeta2) = ffn_b(lay.feed_forward_in, "ffn_in", dx, ln2=ln2)
                put(below.layer_norm.weight, dg2), put(below.layer_norm.bias, dbeta2)
            else:
                dx = ffn_b(lay.feed_forward_in, "ffn_in", dx)
            L.clear()
            if side is not None and getattr(model, "_grad_bucket_hook", None) is None:
                red.flush(side=side)   # the block's weight gradients and parameter reductions beside the next block's chain
            else:
                red.flush()
            _deliver_group(model, lay, G)
        if enc.pos_emb is not None:
            dpe = torch.zeros_like(enc.pos_emb.pe_k.weight)
            dpe.index_add_(0, window[1], dwin[:2 * T - 1])
            put(enc.pos_emb.pe_k.weight, dpe)
        if ds.active:
            ops.dropout_apply(dx, ds.cfg("embed"))
        dwe, dbe, dg, dbeta = bw.embed_bwd(S["f"], dx, S["mean"], S["rinv"], S["we_bf"], enc.embed[0].bias.detach(),
                                           enc.embed[1].weight.detach(), enc.embed[1].bias.detach(), e=S["e"], fn=S["fn"])
        put(enc.embed[0].weight, dwe), put(enc.embed[0].bias, dbe)
        put(enc.embed[1].weight, dg), put(enc.embed[1].bias, dbeta)
        red.flush()
        red.join()
        red.rside.join()
        return (None, None) + _deliver(model, G)


class BlstmTrain(torch.autograd.Function):
    """masks = BLSTM(f) (blstm.py:68-88, dropout off) with the explicit HIP backward: BPTT kernel for the recurrence,
    batched GEMMs / column sums for every weight and input gradient, LayerNorm / ReLU backward kernels."""

    @staticmethod
    def forward(ctx, model, f, *params):
        enc = model.blstm
        h = enc.hidden_dim
        H = h // 2
        B, T, F = f.shape
        M = B * T
        f = f.contiguous()
        mean, rinv = ops.mvn_stats(f)
        Kp = (F + 7) // 8 * 8
        we_bf, _ = ob.pack_weight(enc.embed[0].weight.detach(), want_row_major=True)
        fn = ob.mvn_apply(f, mean, rinv, Kp)
        e = ob.bgemm(fn, we_bf, bias=enc.embed[0].bias.detach())
        _, x = ops.layernorm_relu(e, enc.embed[1].weight.detach(), enc.embed[1].bias.detach())
        ds = model_drop_state(model, getattr(enc, "dropout_rate", 0.0))   # blstm.py:147 (embedding) and :158 (every layer)
        if ds.active:
            ds = DropState(ds.seed.clone(), ds.p, ds.offset)
            ops.dropout_apply(x, ds.cfg("embed"))
        S = {"f": f, "mean": mean, "rinv": rinv, "we_bf": we_bf, "e": e, "layers": [], "drop": ds}
        Hp = (H + 31) // 32 * 32
        for li, lay in enumerate(enc.encoders):
            l = lay.lstm
            w_ih_bf = torch.empty((8 * H, h), dtype=BF, device=f.device)
            whh_p = torch.empty(2 * 4 * H * Hp, dtype=BF, device=f.device)
            whhT_p = torch.empty(2 * H * 4 * H, dtype=BF, device=f.device)
            for d, sfx in enumerate(("", "_reverse")):
                ob.pack_weight(getattr(l, "weight_ih_l0" + sfx).detach(), row_major=w_ih_bf, row0=d * 4 * H)
                whh = getattr(l, "weight_hh_l0" + sfx).detach()
                ob.pack_weight(whh, frag=whh_p, row0=d * 4 * H)
                ob.pack_weight(whh.t().contiguous(), frag=whhT_p, row0=d * H)
            bias = torch.cat([l.bias_ih_l0.detach() + l.bias_hh_l0.detach(),
                              l.bias_ih_l0_reverse.detach() + l.bias_hh_l0_reverse.detach()], 0)
            gx = ob.bgemm(x, w_ih_bf, bias=bias)                       # [M, 8H] = (dir, 4H)
            y, gates, c = ob.lstm_bidir_train(gx, whh_p, B, T, H)
            S["layers"].append({"x": x, "y": y, "gates": gates, "c": c, "w_ih_bf": w_ih_bf, "whhT_p": whhT_p})
            _, x = ops.layernorm_relu(y.view(M, h), lay.layer_norm.weight.detach(), lay.layer_norm.bias.detach(),
                                      relu=False)
            if ds.active:
                ops.dropout_apply(x, ds.cfg(f"{li}.blstm.layer"))
        n_out = model.linear.out_features
        wh_bf, _ = ob.pack_weight(model.linear.weight.detach(), want_row_major=True)
        ld = (n_out + 3) // 4 * 4
        z = torch.empty((M, ld), dtype=torch.float32, device=f.device)[:, :n_out]
        ob.bgemm(x, wh_bf, bias=model.linear.bias.detach(), out=z)
        masks = torch.relu_(z)  # elementwise clamp on the GEMM output; values == the eval path's fused epilogue
        S.update(x_last=x, masks=masks, wh_bf=wh_bf)
        ctx.model, ctx.S = model, S
        return masks.view(B, T, n_out)

    @staticmethod
    def backward(ctx, grad_masks):
        model, S = ctx.model, ctx.S
        ctx.S = None
        enc = model.blstm
        h = enc.hidden_dim
        H = h // 2
        B, T, F = S["f"].shape
        M = B * T
        G = {}

        def put(p, g):
            G[p] = g.reshape(p.shape)

        masks = S["masks"]
        dm = grad_masks.reshape(M, -1)
        if dm.stride(1) != 1 or dm.stride(0) % 4:
            buf = torch.empty_strided(masks.shape, masks.stride(), dtype=torch.float32, device=masks.device)
            buf.copy_(dm)
            dm = buf
        dx, dw, db = bw.head_bwd(dm, masks, S["x_last"], S["wh_bf"])
        put(model.linear.weight, dw), put(model.linear.bias, db)
        ds = S["drop"]
        nlay = len(enc.encoders)
        for li, lay, L in zip(range(nlay - 1, -1, -1), reversed(list(enc.encoders)), reversed(S["layers"])):
            y = L["y"]
            if ds.active:
                dx = ops.dropout_apply(dx.contiguous(), ds.cfg(f"{li}.blstm.layer"))
            dy, dg, dbeta = ob.layernorm_bwd(y.view(M, h), dx, lay.layer_norm.weight.detach(), lay.layer_norm.bias.detach())
            put(lay.layer_norm.weight, dg), put(lay.layer_norm.bias, dbeta)
            dgates = ob.lstm_bidir_bwd(L["gates"], L["c"], dy.view(B, T, h), L["whhT_p"])
            dg2 = dgates.view(M, 8 * H)
            dwih, dbias = ob.linear_wgrad(dg2, L["x"])                        # [8H, h], [8H]
            dx = ob.bgemm(dg2, L["w_ih_bf"], b_kmajor=True)                   # [M, h]
            yp = torch.zeros_like(y)                                          # h_{t-1} of each direction
            yp[:, 1:, :H] = y[:, :-1, :H]
            yp[:, :-1, H:] = y[:, 1:, H:]
            yp2 = yp.view(M, 2 * H)
            l = lay.lstm
            for d, sfx in enumerate(("", "_reverse")):
                dwhh = ob.bgemm(dg2[:, d * 4 * H:(d + 1) * 4 * H], yp2[:, d * H:(d + 1) * H], a_kmajor=True, b_kmajor=True)
                put(getattr(l, "weight_ih_l0" + sfx), dwih[d * 4 * H:(d + 1) * 4 * H])
                put(getattr(l, "weight_hh_l0" + sfx), dwhh)
                put(getattr(l, "bias_ih_l0" + sfx), dbias[d * 4 * H:(d + 1) * 4 * H])
                put(getattr(l, "bias_hh_l0" + sfx), dbias[d * 4 * H:(d + 1) * 4 * H])
            L.clear()
        if ds.active:
            dx = ops.dropout_apply(dx.contiguous(), ds.cfg("embed"))
        dwe, dbe, dg, dbeta = bw.embed_bwd(S["f"], dx, S["mean"], S["rinv"], S["we_bf"], enc.embed[0].bias.detach(),
                                           enc.embed[1].weight.detach(), enc.embed[1].bias.detach(), e=S["e"])
        put(enc.embed[0].weight, dwe), put(enc.embed[0].bias, dbe)
        put(enc.embed[1].weight, dg), put(enc.embed[1].bias, dbeta)
        return (None, None) + _deliver(model, G)


def masks_for_training(model, f):
    params = tuple(model.parameters())
    enc = getattr(model, "conformer", None)
    if enc is not None:
        if enc.d_model in (256, 512) and enc.d_model // enc.n_head == 64 and enc.kernel_size == 33 and enc.d_ffn % 64 == 0:
            return ConformerTrain.apply(model, f, *params)
        raise NotImplementedError("css_amd: the Conformer backward kernels cover d_model 256 / 512 with 64-wide heads, "
                                  "kernel 33 and d_ffn % 64 == 0; got "
                                  f"d_model {enc.d_model}, {enc.n_head} heads, kernel {enc.kernel_size}, d_ffn {enc.d_ffn}")
    if hasattr(model, "blstm"):
        H = model.blstm.hidden_dim // 2
        if H % 16 == 0 and (H <= 128 or H % 128 == 0) and H <= 512 and model.blstm.hidden_dim <= 1024:
            return BlstmTrain.apply(model, f, *params)
        raise NotImplementedError("css_amd: the BLSTM back-propagation-through-time kernel covers hidden_dim = 32k up to "
                                  f"256 and 512, 768, 1024; got {model.blstm.hidden_dim}")
    raise NotImplementedError(f"css_amd: no backward for {type(model).__name__}")
