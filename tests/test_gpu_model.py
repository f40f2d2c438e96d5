"""GPU parity of the whole hot path: css_amd.models.Conformer + PIT-MSE vs the reference's
golden masks/loss (tests/golden/conformer.npz) and vs the oracle at other shapes.

north_star tolerance: masks within 1e-4 MSE of the reference CPU path (bf16 MFMA GEMMs with
fp32 accumulation, fp32 LayerNorm / softmax / BN statistics); PIT loss within 1e-3 relative.
"""
import numpy as np
import pytest
import torch

import oracle
from css_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def build(num_blocks, seed, idim=257, num_spk=2):
    from css_amd.models import MODELS
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=num_blocks, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    m = MODELS["Conformer"](idim, 257, num_spk, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(idim=idim, num_spk=num_spk, num_blocks=num_blocks), seed)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def mask_mse(a, b):
    return float(((a.detach().double().cpu() - b.detach().double().cpu()) ** 2).mean())


@pytest.mark.parametrize("nb", [2, 16])
def test_conformer_golden(golden, nb):
    g = golden("conformer.npz")
    m, _ = build(nb, 30 + nb)
    f_in = rnd((2, 100, 257), 12).abs()
    tg = rnd((2, 2, 100, 257), 13).abs() * 0.5
    y_pred, masks = m(f_in.to(DEV))
    got = torch.stack(masks, 0)
    ref = torch.from_numpy(g[f"masks_b{nb}"])
    mse = mask_mse(got, ref)
    print(f"[parity] Conformer {nb} blocks: mask MSE vs reference {mse:.3e}, max abs {float((got.cpu()-ref).abs().max()):.3e}")
    assert mse < 1e-4
    assert y_pred.shape == (2, 2, 100, 257)
    assert torch.equal(y_pred[:, 0], masks[0] * f_in.to(DEV))
    # fused objective == golden loss of the reference (tuple-shim objective)
    from css_amd.objectives import OBJECTIVES
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": f_in, "source1": tg[:, 0], "source2": tg[:, 1]}, device=DEV)
    ref_loss = float(g[f"loss_b{nb}"])
    print(f"[parity] PIT loss {float(loss):.6f} vs reference {ref_loss:.6f}")
    assert abs(float(loss) - ref_loss) < 1e-3 * abs(ref_loss)


def test_encoder_layer_by_layer():
    """Encoder output (before the head) after 1 block vs the oracle: localises errors."""
    m, sd = build(1, 41)
    f = rnd((2, 70, 257), 14).abs()
    masks, enc = m.forward_masks(f.to(DEV), want_encoder_out=True)
    ref = oracle.conformer.encoder(oracle.conformer.mvn_time(f), sd, num_blocks=1)
    e = float((enc.cpu() - ref).abs().max())
    print(f"[parity] 1-block encoder output max abs err {e:.3e} (values are LayerNorm outputs, O(1))")
    assert e < 6e-2 and mask_mse(enc, ref) < 1e-4


def test_conformer_long_clamp(golden):
    g = golden("conformer.npz")
    m, _ = build(1, 41)
    f_long = rnd((1, 1100, 257), 14).abs()
    _, masks = m(f_long.to(DEV))
    got = torch.stack(masks, 0).cpu()
    assert mask_mse(got[:, :, :40], torch.from_numpy(g["masks_long_head"])) < 1e-4
    assert mask_mse(got[:, :, -40:], torch.from_numpy(g["masks_long_tail"])) < 1e-4


def test_conformer_batch_independent_and_4d():
    m, sd = build(2, 32)
    f = rnd((3, 50, 257), 15).abs().to(DEV)
    _, m_all = m(f)
    _, m_one = m(f[1:2].unsqueeze(0))  # 1 x N x T x F input is squeezed (conformer.py:90-91)
    # not bit-identical: a row's panel index decides the order in which the hidden chunks are summed (fp32),
    # and a flipped bf16 rounding downstream amplifies that; measured 2e-6, budget for parity is 1e-4
    assert mask_mse(torch.stack(m_all)[:, 1], torch.stack(m_one)[:, 0]) < 1e-5


def test_conformer_three_speakers_7ch_dims():
    """config 3/5 shapes: idim 1799 (7ch IPD features), 3 speakers + noise head."""
    m, sd = build(1, 42, idim=1799, num_spk=3)
    f = rnd((1, 50, 1799), 15)
    f[..., :257] = f[..., :257].abs()
    y_pred, masks = m(f.to(DEV))
    assert len(masks) == 4 and y_pred.shape == (1, 3, 50, 257)
    h = oracle.conformer.encoder(oracle.conformer.mvn_time(f), sd, num_blocks=1)
    ref = torch.relu(torch.nn.functional.linear(h, sd["linear.weight"], sd["linear.bias"]))
    assert mask_mse(torch.cat(masks, 2), ref) < 1e-4


def test_conformer_large_dims_generic_path():
    """config 5 shape class: Conformer-large widths (d_model 512, 8 heads, ffn 2048), 3 speakers + noise head,
    through the any-width path of csship_conformer_forward (2 blocks), vs the oracle."""
    from css_amd.models import MODELS
    conf = dict(attention_dim=512, attention_heads=8, linear_units=2048, num_blocks=2, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 3, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_spk=3, attention_dim=512, attention_heads=8, linear_units=2048,
                                                  num_blocks=2), 77)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    f = rnd((2, 90, 257), 31).abs()
    y_pred, masks = m(f.to(DEV))
    assert len(masks) == 4 and y_pred.shape == (2, 3, 90, 257)
    _, o = oracle.conformer.forward(f, sd, num_spk=3, n_head=8, num_blocks=2)
    mse = mask_mse(torch.stack(masks), torch.stack(o))
    print(f"[parity] Conformer-large dims (2 blocks, 4 masks): mask MSE vs oracle {mse:.3e}")
    assert mse < 1e-4
    # the same generic path at base width agrees with the tuned path
    mb, sdb = build(2, 32)
    fb = rnd((2, 70, 257), 33).abs().to(DEV)
    ref = mb.forward_masks(fb).clone()
    mb.force_generic = True
    got = mb.forward_masks(fb)
    mb.force_generic = False
    assert mask_mse(got, ref) < 2e-5


def test_fused_conv_path_matches():
    """Optional eval path: conv module split over csship_outproj_glu + csship_encoder_chain (panels straddle
    utterance boundaries: T = 70, 64-row panels)."""
    m, sd = build(2, 32)
    f = rnd((3, 70, 257), 21).abs().to(DEV)
    ref = m.forward_masks(f).clone()
    o = torch.cat(oracle.conformer.forward(f.cpu(), sd, num_blocks=2)[1], 2)
    # 1: conv inside the chain kernel; 2: out-proj + GLU, then conv from g; 3: out-proj + conv module in one kernel
    # (50-frame tiles: T = 70 has a partial second tile); 4: separate out-proj and conv kernels
    for mode in (1, 2, 3, 4):
        m.fuse_conv = mode
        got = m.forward_masks(f)
        assert mask_mse(got, ref) < 1e-6 and mask_mse(got, o) < 1e-4, mode
    m.fuse_conv = 0
    # a length that is an exact multiple of the tile, and one frame more
    for T in (100, 101):
        f2 = rnd((2, T, 257), 22).abs().to(DEV)
        m.fuse_conv = 4
        r2 = m.forward_masks(f2).clone()
        m.fuse_conv = 3
        assert mask_mse(m.forward_masks(f2), r2) < 1e-6
    m.fuse_conv = 0


def test_train_mode_batch_stats():
    m, sd = build(1, 41)
    m.train()
    f = rnd((2, 70, 257), 14).abs()
    masks, enc = m.forward_masks(f.to(DEV), want_encoder_out=True)
    ref = oracle.conformer.encoder(oracle.conformer.mvn_time(f), sd, num_blocks=1, training_bn=True)
    assert mask_mse(enc, ref) < 1e-4


def test_no_cpu_path():
    from css_amd.models import MODELS
    m = MODELS["Conformer"](257, 257, 2, 1, dict(attention_dim=256, attention_heads=4, linear_units=1024,
                                                   num_blocks=1, kernel_size=33, dropout_rate=0.1,
                                                   relative_pos_emb=True))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 10, 257))


def test_blstm_golden(golden):
    """config[0] shape class: BLSTM 1ch 2-spk (tiny hidden size) vs the reference golden masks / y_pred."""
    from css_amd.models import MODELS
    g = golden("blstm.npz")
    m = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=64, num_layers=2, dropout_rate=0.1))
    m.load_state_dict(syn.fill_state_dict(syn.blstm_shapes(hidden_dim=64, num_layers=2), 50))
    m = m.to(DEV).eval()
    fb = rnd((2, 30, 257), 16).abs()
    y_pred, masks = m(fb.to(DEV))
    mse = mask_mse(torch.stack(masks, 0), torch.from_numpy(g["masks"]))
    print(f"[parity] BLSTM tiny: mask MSE vs reference {mse:.3e}")
    assert mse < 1e-4 and mask_mse(y_pred, torch.from_numpy(g["y_pred"])) < 1e-4


@pytest.mark.parametrize("H,B,T", [(256, 1, 7), (256, 20, 33), (512, 16, 50), (512, 32, 400), (512, 35, 21), (512, 70, 9),
                                   (256, 130, 5)])
def test_lstm_distributed_recurrence_equals_single_cu_kernel(H, B, T):
    """csship_lstm_bidir_dist / csship_lstm_bidir_bwd_dist (W_hh in registers across H/32 CUs, state exchanged per step
    through write-through global panels + epoch tags / flags salted per launch) vs the single-CU kernels that stream
    W_hh: same arithmetic order, so bitwise equal outputs, gates, cell states and gate gradients; no hand-off timeout.
    Every repetition uses DIFFERENT inputs and weights and alternates forward and backward on the one shared
    workspace, so a granule or flag left behind by an earlier launch would be visible as a mismatch."""
    from css_amd import ops, ops_bwd as ob
    from css_amd._lib import check, lib
    assert ops.lstm_dist_enabled(H, B) and ops.lstm_dist_enabled(H, B, backward=True)
    for it in range(3):
        g = torch.Generator().manual_seed(1000 * it + H + B + T)
        gx = (torch.randn((B * T, 8 * H), generator=g) * 0.5).to(DEV)
        w_raw = [(torch.randn((4 * H, H), generator=g) / H ** 0.5).to(DEV) for _ in range(2)]
        whh = torch.cat([ops.pack_frag(w) for w in w_raw]).contiguous()
        whhT = torch.cat([ops.pack_frag(w.t().contiguous()) for w in w_raw]).contiguous()
        y0 = torch.empty((B, T, 2 * H), device=DEV)
        g0 = torch.empty((B, T, 2, 4 * H), device=DEV)
        c0 = torch.empty((B, T, 2, H), device=DEV)
        check(lib().csship_lstm_bidir_train(gx.data_ptr(), whh.data_ptr(), y0.data_ptr(), g0.data_ptr(), c0.data_ptr(), B, T, H, None))
        y1, g1, c1 = ops.lstm_bidir(gx, whh, B, T, H, keep=True)
        ops.lstm_dist_check()
        assert torch.equal(y1, y0) and torch.equal(g1, g0) and torch.equal(c1, c0), it
        # back-propagation through time: distributed vs single-CU kernel on the saved gates / cell states
        dy = (torch.randn((B, T, 2 * H), generator=g) * 0.1).to(DEV)
        d0 = torch.empty_like(g0)
        check(lib().csship_lstm_bidir_bwd(g0.data_ptr(), c0.data_ptr(), dy.data_ptr(), whhT.data_ptr(), d0.data_ptr(), B, T, H, None))
        d1 = ob.lstm_bidir_bwd(g0, c0, dy, whhT)
        ops.lstm_dist_check()
        assert torch.equal(d1, d0), it
        y2 = ops.lstm_bidir(gx, whh, B, T, H)   # inference form straight after a backward on the same workspace
        ops.lstm_dist_check()
        assert torch.equal(y2, y0), it


def test_lstm_dist_error_word_is_sticky():
    """The hand-off error word survives later launches (they memset only their flags / panels) until
    ops.lstm_dist_check() reads and clears it."""
    from css_amd import ops
    from css_amd._lib import lib
    H, B, T = 256, 4, 6
    g = torch.Generator().manual_seed(5)
    gx = (torch.randn((B * T, 8 * H), generator=g) * 0.5).to(DEV)
    whh = torch.cat([ops.pack_frag((torch.randn((4 * H, H), generator=g) / 16).to(DEV)) for _ in range(2)]).contiguous()
    ops.lstm_bidir(gx, whh, B, T, H)
    ops.lstm_dist_check()
    (key, ws), = [(k, v) for k, v in ops._LSTM_WS.items() if k[2] == B and k[3] == H]
    off = lib().csship_lstm_dist_workspace_error_offset(B, H)
    assert off == ws.numel() - 64
    ws[off:off + 4].view(torch.int32).fill_(1)      # what a timed-out wave does
    ops.lstm_bidir(gx, whh, B, T, H)                # a later, healthy launch must not erase it
    with pytest.raises(RuntimeError, match="timed out"):
        ops.lstm_dist_check()
    ops.lstm_dist_check()                           # cleared by the failed check


def test_blstm_default_width_vs_oracle():
    """CLI-default width (hdim 1024 -> H = 512 per direction: streamed W_hh, 4 unit tiles per wave), 1 layer."""
    from css_amd.models import MODELS
    m = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=1024, num_layers=1, dropout_rate=0.1))
    sd = syn.fill_state_dict(syn.blstm_shapes(hidden_dim=1024, num_layers=1), 52)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    fb = rnd((3, 40, 257), 17).abs()
    _, masks = m(fb.to(DEV))
    _, o_masks = oracle.blstm.forward(fb, sd, num_layers=1)
    mse = mask_mse(torch.stack(masks, 0), torch.stack(o_masks, 0))
    print(f"[parity] BLSTM hdim 1024: mask MSE vs oracle {mse:.3e}")
    assert mse < 1e-4


def test_train_one_epoch_matches_reference_trace():
    """6-step loss trace of the reference's train_one_epoch on a tiny BLSTM (tests/golden/trainer.json): HIP
    forward + hand-written PIT backward + stop-gap encoder backward, Adam, warm-up schedule."""
    import json, os, types
    from conftest import GOLDEN
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    from css_amd.trainer import LRScheduler, train_one_epoch
    tr = json.load(open(os.path.join(GOLDEN, "trainer.json")))
    m = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=32, num_layers=1, dropout_rate=0.0))
    m.load_state_dict(syn.fill_state_dict(syn.blstm_shapes(hidden_dim=32, num_layers=1), 51))
    m = m.to(DEV).train()

    def gen():
        i = 0
        while True:
            yield {"mix": rnd((2, 20, 257), 200 + i).abs(), "source1": rnd((2, 20, 257), 300 + i).abs() * 0.5,
                   "source2": rnd((2, 20, 257), 400 + i).abs() * 0.5, "ovl": 0.5}
            i += 1

    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = LRScheduler(opt, types.SimpleNamespace(warmup=3, decay=0.0, fixed=0, min_lr=1e-9))
    targs = types.SimpleNamespace(gpu=True, fp16=False, batches_per_epoch=1, grad_thresh=5.0)
    g = gen()
    losses = [train_one_epoch(targs, g, m, OBJECTIVES["MSE"].build_objective({}), opt, sch, DEV) for _ in range(6)]
    print("[parity] train losses", [round(v, 5) for v in losses], "reference", [round(v, 5) for v in tr["train_losses"]])
    assert np.allclose(losses, tr["train_losses"], rtol=5e-3)


def test_train_one_epoch_replays_a_graph_with_fused_adam():
    """train_one_epoch with css_amd.optim.FusedAdam captures the step the second time a batch shape is met (the first
    occurrence runs eagerly; at most args.graph_cache captured shapes are kept) and replays it: same losses and
    parameters, bit for bit, as the same loop launched kernel by kernel (args.graph = False), with the warm-up learning-rate
    schedule changing the rate every step and dropout on (the seed advances inside the graph)."""
    import types
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.trainer import LRScheduler, train_one_epoch

    def gen():
        i = 0
        while True:
            yield {"mix": rnd((2, 40, 257), 200 + i).abs().to(DEV), "source1": (rnd((2, 40, 257), 300 + i).abs() * 0.5).to(DEV),
                   "source2": (rnd((2, 40, 257), 400 + i).abs() * 0.5).to(DEV), "ovl": 0.5}
            i += 1

    def run(graph):
        m, _ = build(2, 61)
        m.train()
        opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=5.0)
        sch = LRScheduler(opt, types.SimpleNamespace(warmup=3, decay=0.0, fixed=0, min_lr=1e-9))
        targs = types.SimpleNamespace(gpu=True, fp16=False, batches_per_epoch=2, grad_thresh=5.0, graph=graph)
        g = gen()
        obj = OBJECTIVES["MSE"].build_objective({})
        losses = [train_one_epoch(targs, g, m, obj, opt, sch, DEV) for _ in range(3)]
        return losses, opt.flat_param.clone(), len(getattr(opt, "_graphed_steps", {"cache": {}})["cache"])
    l_e, p_e, n_e = run(False)
    l_g, p_g, n_g = run(True)
    assert n_e == 0 and n_g == 1
    assert l_e == l_g and torch.equal(p_e, p_g)


def test_graphed_step_cache_is_bounded_and_first_sight_runs_eager():
    """The reference draws the window length per batch (separation.py:94-95,175): shapes seen once run the eager step,
    a repeated shape is captured, and the cache evicts the least recently used captured step beyond args.graph_cache."""
    import types
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.trainer import LRScheduler, train_one_epoch
    m, _ = build(1, 61)
    m.train()
    opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=5.0)
    sch = LRScheduler(opt, types.SimpleNamespace(warmup=3, decay=0.0, fixed=0, min_lr=1e-9))
    frames = [24, 24, 32, 32, 40, 40, 24, 56]   # 24, 32, 40 captured in turn (cache of 2: 24 evicted, re-seen -> eager count restarts)

    def gen():
        for i, t in enumerate(frames):
            yield {"mix": rnd((2, t, 257), 200 + i).abs().to(DEV), "source1": (rnd((2, t, 257), 300 + i).abs() * 0.5).to(DEV),
                   "source2": (rnd((2, t, 257), 400 + i).abs() * 0.5).to(DEV), "ovl": 0.5}
    targs = types.SimpleNamespace(gpu=True, fp16=False, batches_per_epoch=len(frames), grad_thresh=5.0, graph=True, graph_cache=2)
    loss = train_one_epoch(targs, gen(), m, OBJECTIVES["MSE"].build_objective({}), opt, sch, DEV)
    st = opt._graphed_steps
    assert np.isfinite(loss) and len(st["cache"]) == 2
    assert [k[3][1][1] for k in st["cache"]] == [32, 40]   # key = (model, objective, clip, ("mix", shape), ...)


def test_mixed_eager_and_replayed_steps_equal_the_eager_loop():
    """Shapes alternate the way the reference's random 2-5 s windows do (separation.py:94-95,175): a shape seen for the
    first time runs eagerly BETWEEN replays of captured shapes, and an evicted shape runs eagerly again.  Every eager
    step after a replay must (a) advance the step count / upload lr and the Adam bias corrections itself and (b) start
    from a zeroed gradient buffer (the replay leaves its gradients in it and the explicit backward accumulates): losses,
    parameters, Adam moments and the step count equal the graph=False loop bit for bit, with the warm-up schedule
    changing the rate every step and dropout on (both high findings)."""
    import types
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.trainer import LRScheduler, train_one_epoch
    frames = [24, 24, 32, 24, 32, 32, 40, 24, 40, 40, 24]

    def gen():
        for i, t in enumerate(frames):
            yield {"mix": rnd((2, t, 257), 200 + i).abs().to(DEV), "source1": (rnd((2, t, 257), 300 + i).abs() * 0.5).to(DEV),
                   "source2": (rnd((2, t, 257), 400 + i).abs() * 0.5).to(DEV), "ovl": 0.5}

    def run(graph):
        m, _ = build(1, 61)
        m.train()
        opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=5.0)
        sch = LRScheduler(opt, types.SimpleNamespace(warmup=5, decay=1e-2, fixed=0, min_lr=1e-9))
        targs = types.SimpleNamespace(gpu=True, fp16=False, batches_per_epoch=1, grad_thresh=5.0, graph=graph, graph_cache=2)
        g, obj = gen(), OBJECTIVES["MSE"].build_objective({})
        losses = [train_one_epoch(targs, g, m, obj, opt, sch, DEV) for _ in frames]
        replayed = sum(v is not None for v in getattr(opt, "_graphed_steps", {"cache": {}})["cache"].values())
        return losses, opt.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt._step, replayed
    l_e, p_e, m_e, v_e, s_e, n_e = run(False)
    l_g, p_g, m_g, v_g, s_g, n_g = run(True)
    assert n_e == 0 and n_g == 2            # something WAS replayed in the second run
    assert s_e == s_g == len(frames)        # every step, eager or replayed, counted once
    assert l_e == l_g, (l_e, l_g)
    assert torch.equal(p_e, p_g) and torch.equal(m_e, m_g) and torch.equal(v_e, v_g)


def test_conformer_gradients_flow_and_match_oracle_autograd():
    """Gradient check of the training path on a 1-block Conformer with BatchNorm frozen (model.eval() +
    eval_gradients): loss.backward() fills every parameter's grad, and the gradients agree with torch.autograd through
    the ORACLE (fp32, CPU) evaluated at the permutation the fused objective chose."""
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    m, sd = build(1, 41)
    f = rnd((2, 40, 257), 14).abs().to(DEV)
    tg = [(rnd((2, 40, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    assert not obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV).requires_grad  # eval: inference path
    m.eval_gradients = True   # differentiate with BatchNorm frozen at its running statistics
    loss = obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV)
    assert loss.requires_grad
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    ref = grad_ref.oracle_param_grads(m, lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm), f, training_bn=False)
    err, worst = grad_ref.compare(grads, ref)
    print(f"[parity] frozen-BN gradients vs oracle autograd: global rel err {err:.3e}, worst {worst[0]} {worst[1]:.3e}")
    assert err < 3e-2 and worst[1] < 0.15  # bf16 operands in forward and backward against an fp32 graph
    # ... and against the oracle evaluated WITH those operand roundings (oracle/emulate.py; tools/grad_floor.py prints the
    # per-parameter table, profiles/r03_grad_table.txt: measured 7.5e-3 global, 2.2e-2 worst live parameter)
    ref16 = grad_ref.oracle_param_grads(m, lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm), f, training_bn=False,
                                        emulate_bf16=True)
    g16, w16 = grad_ref.check_against_emulation(grads, ref16, global_tol=1.5e-2, param_tol=5e-2)
    print(f"[parity] ... vs the bf16-operand oracle: global {g16:.3e}, worst live parameter {w16[0]} {w16[1]:.3e}")


@pytest.mark.parametrize("B,T", [(2, 7), (5, 17), (1, 33)])
def test_conformer_hip_backward_on_tiny_chunks_within_bf16_noise(B, T):
    """A handful of frames (shorter than the conv kernel, than one attention tile, than one row panel): the relative
    error of a gradient grows like 1 / sqrt(rows) with bf16 operands - the oracle with bf16 operands shows the same
    distance from fp32 autograd as the HIP path does (measured: 5.9e-2 against 7.9e-2 at 2 x 7, equal at 1 x 33 and
    5 x 17), which is what separates operand rounding from a mishandled edge."""
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    m, sd = build(2, 43)
    m.train()
    f = rnd((B, T, 257), 14).abs().to(DEV)
    tg = [(rnd((B, T, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV).backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert all(torch.isfinite(g).all() for g in grads.values())
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    lf = lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm)   # noqa: E731
    ref = grad_ref.oracle_param_grads(m, lf, f, training_bn=True)
    ref16 = grad_ref.oracle_param_grads(m, lf, f, training_bn=True, emulate_bf16=True)
    e32, _ = grad_ref.compare(grads, ref)
    e16, _ = grad_ref.compare(grads, ref16)
    noise, _ = grad_ref.compare(ref16, ref)
    print(f"[parity] {B} x {T}: HIP vs fp32 autograd {e32:.3e}, vs the bf16-operand oracle {e16:.3e}; that oracle vs fp32 {noise:.3e}")
    assert e32 < 1.6 * noise + 5e-3 and e16 < 1.3 * noise + 5e-3


@pytest.mark.parametrize("B,T", [(3, 60), (2, 96), (4, 64), (1, 460)])
def test_conformer_hip_backward_matches_oracle_autograd(B, T):
    """Training path with the explicit HIP backward (css_amd/autograd.py ConformerTrain, 2 blocks, batch-statistics
    BatchNorm): every parameter gradient vs torch.autograd through the oracle (fp32, training_bn=True) at the same
    permutation.  bf16 GEMM operands in forward and backward: global relative error bound 3e-2.
    (3, 60): 180 rows, a ragged last panel - the bounds-checked kernels; (2, 96): 192 rows = whole 64-row panels - the
    static forms the full-size step runs (chain_s_kernel<TRAIN>, ffn_bwd_kernel<FULL, LN2>, qkv_in_bwd_kernel,
    panel_dx_kernel, the grouped weight gradients); (1, 460): a chunk longer than the fused position pass takes (449
    frames) - the skewed dS in memory and the two tiled position products."""
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    m, sd = build(2, 43)
    m.train()
    f = rnd((B, T, 257), 14).abs().to(DEV)
    tg = [(rnd((B, T, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    rm0 = m.conformer.encoders[0].conv.BN.running_mean.clone()
    cnt0 = [int(l.conv.BN.num_batches_tracked) for l in m.conformer.encoders]
    ref_state = {k: v.clone() for k, v in m.state_dict().items()}  # the oracle must see the PRE-step running statistics
    loss = obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV)
    assert isinstance(loss.grad_fn, torch.autograd.graph.Node) and loss.requires_grad
    loss.backward()
    assert not torch.equal(rm0, m.conformer.encoders[0].conv.BN.running_mean)   # BatchNorm bookkeeping happened
    assert [int(l.conv.BN.num_batches_tracked) for l in m.conformer.encoders] == [c + 1 for c in cnt0]
    sdn = m.state_dict()   # the counters are views of one tensor now: keys, dtypes and values as before
    assert all(sdn[f"conformer.encoders.{i}.conv.BN.num_batches_tracked"].dtype == torch.int64 and
               int(sdn[f"conformer.encoders.{i}.conv.BN.num_batches_tracked"]) == cnt0[i] + 1 for i in range(2))
    m.load_state_dict({k: v.clone() for k, v in sdn.items()})
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    ref = grad_ref.oracle_param_grads(m, lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm), f, training_bn=True)
    # (rounds 4-5 held (3, 60) to one bound for every parameter: its scalar sums happened to be well conditioned under the
    # masks of the old dropout function; under round 6's masks pw_conv_1.weight reads 0.39 there.  The scalar parameters
    # are pinned to 3e-4 at kernel level - test_gpu_bwd.py, both forward forms - and bounded by their conditioning here.)
    strict = False
    err, worst = grad_ref.compare(grads, ref, worst_min_numel=0 if strict else 3)
    print(f"[parity] HIP backward vs oracle autograd: global rel err {err:.3e}, worst {worst[0]} {worst[1]:.3e}")
    assert err < 3e-2
    assert worst[1] < 0.15
    # against the oracle with bf16 operands in forward and backward (measured: 1.6e-2 global, 7.4e-2 on the conv module's
    # scalar bias - a sum with heavy cancellation over every position; profiles/r03_grad_table.txt)
    ref16 = grad_ref.oracle_param_grads(m, lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm), f, training_bn=True,
                                        emulate_bf16=True)
    # the conv module's 1- / 2-element parameters: bounded by how ill-conditioned their sums are at THIS shape
    ws = grad_ref.scalar_sums_within_conditioning(grads, ref, ref16)
    print(f"[parity] ... scalar conv parameters: worst {ws[0]} {ws[1]:.3e} (allowed {ws[2]:.3e})")
    g16, w16 = grad_ref.check_against_emulation(grads, ref16, global_tol=2.5e-2, param_tol=0.12, min_numel=0 if strict else 3)
    print(f"[parity] ... vs the bf16-operand oracle: global {g16:.3e}, worst live parameter {w16[0]} {w16[1]:.3e}")
    del ref_state


def test_conformer_without_relative_positions():
    """`relative_pos_emb=False` (conformer.py:131-137, 236: scores = q . k / sqrt(d_k)): no pe_k parameter, eval forward
    and train-mode gradients vs the oracle run the same way."""
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=2, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=False)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=2, relative_pos_emb=False), 47)
    assert not any("pe_k" in k for k in m.state_dict()) and set(m.state_dict()) == set(sd)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    f = rnd((2, 90, 257), 15).abs()
    want = torch.cat(oracle.conformer.forward(f, sd, num_blocks=2, relative_pos_emb=False)[1], -1)
    with_pos = torch.cat(oracle.conformer.forward(f, dict(sd, **{"conformer.pos_emb.pe_k.weight": 0.3 * rnd((2000, 64), 5)}),
                                                  num_blocks=2)[1], -1)
    got = m.forward_masks(f.to(DEV))
    print(f"[parity] no relative positions: mask MSE {mask_mse(got, want):.3e} (vs the relative model {mask_mse(with_pos, want):.3e})")
    assert mask_mse(got, want) < 1e-4 and mask_mse(with_pos, want) > 10 * mask_mse(got, want)
    m.train()
    tg = [(rnd((2, 90, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": f.to(DEV), "source1": tg[0], "source2": tg[1]}, device=DEV)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    ref = grad_ref.oracle_param_grads(m, lambda mk: grad_ref.pit_loss_with_perm(mk, f, tc, perm), f.to(DEV), training_bn=True)
    err, worst = grad_ref.compare(grads, ref)
    print(f"[parity] no relative positions: gradients vs oracle autograd {err:.3e}, worst {worst[0]} {worst[1]:.3e}")
    assert err < 6e-2


def test_causal_conformer_forward_and_gradients(golden):
    """ConformerEncoder(causal=True) (conformer.py:131, 265-273): eval forward of a 2-block model against the REFERENCE's
    own causal encoder (tests/golden/causal.npz, through the head-free oracle for the masks), and train-mode gradients
    through the explicit backward (causal halo in the conv module's forward, tap-gradient and input-gradient passes)
    against autograd through the oracle."""
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=2, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True, causal=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    shapes = {k[len("conformer."):]: v for k, v in syn.conformer_shapes(num_blocks=2).items() if k.startswith("conformer.")}
    sd = {"conformer." + k: v for k, v in syn.fill_state_dict(shapes, 31).items()}
    sd.update({k: v for k, v in syn.fill_state_dict(syn.conformer_shapes(num_blocks=2), 32).items() if k.startswith("linear.")})
    assert set(m.state_dict()) == set(sd)      # same keys and shapes as the non-causal model (conformer.py:265-282)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    f = rnd((2, 90, 257), 12).abs()
    with torch.no_grad():
        masks, enc = m._forward_masks_hip(f.to(DEV), want_encoder_out=True)
    g = golden("causal.npz")
    e_enc = float(((enc.cpu() - torch.from_numpy(g["encoder2"])) ** 2).mean())
    want = torch.cat(oracle.conformer.forward(f, sd, num_blocks=2, causal=True)[1], -1)
    noncausal = torch.cat(oracle.conformer.forward(f, sd, num_blocks=2)[1], -1)
    print(f"[parity] causal encoder vs the reference's causal ConformerEncoder: MSE {e_enc:.3e}; masks vs oracle "
          f"{mask_mse(masks, want):.3e} (the non-causal model is {mask_mse(noncausal, want):.3e} away)")
    assert e_enc < 1e-4 and mask_mse(masks, want) < 1e-4 and mask_mse(noncausal, want) > 10 * mask_mse(masks, want)
    m.train()
    tg = [(rnd((2, 90, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": f.to(DEV), "source1": tg[0], "source2": tg[1]}, device=DEV)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    of = lambda mk: grad_ref.pit_loss_with_perm(mk, f, tc, perm)   # noqa: E731
    ref = grad_ref.oracle_param_grads(m, of, f.to(DEV), training_bn=True)
    err_g, worst = grad_ref.compare(grads, ref)
    print(f"[parity] causal model: gradients vs oracle autograd {err_g:.3e}, worst {worst[0]} {worst[1]:.3e}")
    for k, e, d in grad_ref.per_parameter(grads, ref)[:5]:
        print(f"    {k:50s} rel {e:.3e} |ref| {d:.3e}  got {float(grads[k].flatten()[0]):+.4e} ref {float(ref[k].flatten()[0]):+.4e}")
    # bf16 operand noise of a 2-block model (the conv module's own backward, all f32, is held to 2e-4 against float64 in
    # tests/test_gpu_bwd.py::test_convmod_backward[True]); per parameter against the oracle evaluated with the same
    # operand roundings, live parameters only (grad_ref.check_against_emulation)
    assert err_g < 6e-2
    ref16 = grad_ref.oracle_param_grads(m, of, f.to(DEV), training_bn=True, emulate_bf16=True)
    g16, w16 = grad_ref.check_against_emulation(grads, ref16, global_tol=4e-2, param_tol=0.25)
    print(f"[parity] ... vs the bf16-operand oracle: global {g16:.3e}, worst live parameter {w16[0]} {w16[1]:.3e}")


def test_unsupported_widths_raise_instead_of_falling_back():
    """There is no torch-op arithmetic path in the package: a training forward the backward kernels do not cover raises."""
    from css_amd.models import MODELS
    conf = dict(attention_dim=128, attention_heads=2, linear_units=256, num_blocks=1, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf).to(DEV).train()
    with pytest.raises(NotImplementedError):
        m.forward_masks(rnd((1, 20, 257), 3).abs().to(DEV))
    mb = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=48, num_layers=1, dropout_rate=0.0)).to(DEV).train()
    with pytest.raises(NotImplementedError):
        mb.forward_masks(rnd((1, 20, 257), 3).abs().to(DEV))


@pytest.mark.parametrize("hdim,layers,B,T", [(64, 2, 3, 25), (256, 1, 18, 30), (1024, 1, 2, 12)])
def test_blstm_hip_backward_matches_oracle_autograd(hdim, layers, B, T):
    """BLSTM training path with the explicit HIP backward (BPTT kernel + GEMMs) vs torch.autograd through the oracle,
    same permutation.  Covers 1 / 2 / 4 unit tiles per wave and a ragged batch tile (18 = 16 + 2)."""
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    m = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=hdim, num_layers=layers, dropout_rate=0.0))
    m.load_state_dict(syn.fill_state_dict(syn.blstm_shapes(hidden_dim=hdim, num_layers=layers), 53))
    m = m.to(DEV).train()
    f = rnd((B, T, 257), 14).abs().to(DEV)
    tg = [(rnd((B, T, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    ref = grad_ref.oracle_param_grads(m, lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm), f, training_bn=True)
    err, worst = grad_ref.compare(grads, ref)
    print(f"[parity] BLSTM hdim {hdim}: HIP backward vs oracle autograd: global rel err {err:.3e}, worst {worst[0]} {worst[1]:.3e}")
    assert err < 3e-2 and worst[1] < 0.15


def test_graphed_train_step_matches_eager():
    """GraphedTrainStep (whole step captured in one HIP graph, lr and Adam bias corrections read from device memory)
    reproduces eager FusedAdam training bit for bit over 4 steps with a changing learning rate, and its constructor
    leaves model / optimiser state untouched."""
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.trainer import GraphedTrainStep

    def batches(i):
        return {"mix": rnd((2, 40, 257), 500 + i).abs().to(DEV), "source1": (rnd((2, 40, 257), 600 + i).abs() * 0.5).to(DEV),
                "source2": (rnd((2, 40, 257), 700 + i).abs() * 0.5).to(DEV)}

    lrs = [1e-3, 2e-3, 5e-4, 1e-3]
    obj = OBJECTIVES["MSE"].build_objective({})
    runs = []
    for graphed in (False, True):
        m, _ = build(1, 47)
        m.train()
        opt = FusedAdam(m.parameters(), lr=lrs[0], max_norm=5.0)
        losses = []
        if graphed:
            p0 = opt.flat_param.clone()
            stepper = GraphedTrainStep(m, obj, opt, batches(0), max_norm=5.0)
            assert torch.equal(p0, opt.flat_param) and opt._step == 0
        for i, lr in enumerate(lrs):
            opt.param_groups[0]["lr"] = lr
            if graphed:
                loss, _ = stepper(batches(i))
            else:
                opt.zero_grad()
                loss = obj(m, batches(i), device=DEV)
                loss.backward()
                opt.step(max_norm=5.0)
            losses.append(float(loss))
        runs.append((losses, opt.flat_param.clone(), m.conformer.encoders[0].conv.BN.running_mean.clone()))
    print("[parity] eager losses", runs[0][0], "graphed", runs[1][0])
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


def test_direct_gradient_stores_respect_foreign_writes():
    """the explicit backward STORES a weight gradient into FusedAdam's zeroed .grad view instead of
    accumulating.  Any torch-side write between zero_grad() and that store must switch it back to accumulation:
    (a) a second loss term on the same parameters whose AccumulateGrad runs in the same backward pass (an L2 penalty),
    (b) a manual p.grad.add_ before the backward.  Both against the always-additive path (no direct stores)."""
    from css_amd import autograd as ag
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    obj = OBJECTIVES["MSE"].build_objective({})
    batch = {"mix": rnd((2, 64, 257), 510).abs().to(DEV), "source1": (rnd((2, 64, 257), 511).abs() * 0.5).to(DEV),
             "source2": (rnd((2, 64, 257), 512).abs() * 0.5).to(DEV)}

    def run(direct, case):
        from css_amd import dropout
        m, _ = build(2, 49)
        m.train()
        dropout.set_seed(m, 5)
        opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=5.0)
        old = ag._NO_DIRECT
        ag._NO_DIRECT = not direct
        try:
            opt.zero_grad()
            if case == "manual":
                for p in m.parameters():
                    p.grad.add_(0.25)
                loss = obj(m, batch, device=DEV)
            elif case == "l2":
                loss = obj(m, batch, device=DEV) + 1e-1 * sum(p.pow(2).sum() for p in m.parameters())
            else:
                loss = obj(m, batch, device=DEV)
            loss.backward()
        finally:
            ag._NO_DIRECT = old
        torch.cuda.synchronize()
        return opt.flat_grad.clone(), opt

    base, _ = run(True, "plain")
    for case in ("l2", "manual"):
        g_dir, opt = run(True, case)
        g_add, _ = run(False, case)
        err = float((g_dir - g_add).abs().max()) / float(g_add.abs().max())
        moved = float((g_add - base).abs().max())
        print(f"[parity] direct-grad contract, {case}: max |direct - additive| / max |g| = {err:.2e}; the extra term moves g by {moved:.2e}")
        assert moved > 1e-3          # the foreign contribution is really there ...
        assert err < 1e-6            # ... and the store path did not overwrite it
    # and the fast path is still taken when nothing foreign happened: the views are claimed by stores
    _, opt = run(True, "plain")
    assert opt.grads_untouched() and len(opt._dirty) > 0


def test_conformer_backward_through_the_fused_attention_pass():
    """The opt-in key-stationary attention backward (csship_mha_relpos_bwd_fused + csship_attn_pos_bwd_ds_parts, CSSHIP_ATTN_FUSED_BWD=1)
    through a whole 2-block training step with dropout: every parameter gradient against the default path's (the
    query-stationary pass + csship_attn_dvdk) - same arithmetic up to summation order and one bf16 rounding of dS."""
    from css_amd import backward as bw, dropout
    from css_amd.objectives import OBJECTIVES
    obj = OBJECTIVES["MSE"].build_objective({})
    f = rnd((3, 96, 257), 14).abs().to(DEV)
    tg = [(rnd((3, 96, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    grads = []
    for fused in (False, True):
        m, _ = build(2, 43)
        m.train()
        dropout.set_seed(m, 9)
        old = bw._ATTN_FUSED_BWD
        bw._ATTN_FUSED_BWD = fused
        try:
            obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV).backward()
        finally:
            bw._ATTN_FUSED_BWD = old
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())
    err = float((grads[1] - grads[0]).norm() / grads[0].norm())
    print(f"[parity] fused attention backward vs the default path, whole-model gradient: relative L2 difference {err:.3e}")
    assert bool(torch.isfinite(grads[1]).all()) and err < 5e-3


def test_training_does_not_depend_on_host_synchronisation():
    """12 replayed steps with a different learning rate every step, queued without ever synchronising (the host runs
    many steps ahead of the GPU), equal the same steps with a synchronise after each one, bit for bit: the per-step
    hyper-parameters reach the device through per-step pinned staging slots."""
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.trainer import GraphedTrainStep
    obj = OBJECTIVES["MSE"].build_objective({})
    batch = {"mix": rnd((16, 200, 257), 900).abs().to(DEV), "source1": (rnd((16, 200, 257), 901).abs() * 0.5).to(DEV),
             "source2": (rnd((16, 200, 257), 902).abs() * 0.5).to(DEV)}
    finals = []
    for sync_each in (True, False):
        m, _ = build(4, 48)
        m.train()
        opt = FusedAdam(m.parameters(), lr=1e-3, max_norm=5.0)
        stepper = GraphedTrainStep(m, obj, opt, batch, max_norm=5.0)
        for i in range(12):
            opt.param_groups[0]["lr"] = 1e-3 * (1 + i % 5)
            stepper(batch)
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        finals.append(opt.flat_param.clone())
    assert torch.equal(finals[0], finals[1])


def test_forward_bitwise_reproducible():
    """No atomics, fixed reduction orders, barriers where workgroups hand data to themselves: repeated forwards of
    the headline shape (and of a shape with partial tiles everywhere) are bit-identical."""
    m, _ = build(4, 61)
    for B, T in ((32, 400), (3, 217)):
        f = rnd((B, T, 257), 71).abs().to(DEV)
        a = m.forward_masks(f).clone()
        for _ in range(3):
            assert torch.equal(m.forward_masks(f), a)
        tg = [(rnd((B, T, 257), 72 + k).abs() * 0.5).to(DEV) for k in range(2)]
        from css_amd import ops
        l0 = ops.pit_mse(a, f, tg)[0].clone()
        assert torch.equal(ops.pit_mse(a, f, tg)[0], l0)


@pytest.mark.parametrize("B,T", [(1, 2), (1, 9), (2, 49), (2, 50), (1, 51), (5, 129)])
def test_conformer_small_and_ragged_shapes(B, T):
    """Edge shapes through the whole eval path (tiles of 50 / 64 frames, 64-row panels, 64-key tiles all partial) vs
    the oracle, both conv modes."""
    m, sd = build(1, 35)
    f = rnd((B, T, 257), 90 + T).abs()
    o = torch.cat(oracle.conformer.forward(f, sd, num_blocks=1)[1], 2)
    for mode in (0, 4):
        m.fuse_conv = mode
        got = m.forward_masks(f.to(DEV))
        assert bool(torch.isfinite(got).all())
        assert mask_mse(got, o) < 1e-4, (mode, B, T)
    m.fuse_conv = 0


def test_stft_short_signals():
    """fewer frames than one wave's run of 8, and frame counts that are not multiples of it"""
    from css_amd import ops
    for n_frames in (1, 2, 7, 8, 9, 33):
        n = 512 + 256 * (n_frames - 1)
        x = rnd((3, n), 95 + n_frames, 0.1)
        got = ops.stft(x.to(DEV), ("mag", "re", "im"))
        ref = oracle.feature.stft(x)
        assert got["mag"].shape == (3, n_frames, 257)
        for k, r in zip(("mag", "re", "im"), (ref[0], ref[2], ref[3])):
            assert float((got[k].cpu() - r.transpose(1, 2)).abs().max()) < 2e-5 * max(1.0, float(r.abs().max()))


def test_conformer_large_width_hip_backward():
    """d_model 512 / 8 heads / ffn 2048 (config 5 widths), 3 speakers + noise head: the explicit HIP backward through
    the size-generic kernels vs torch.autograd through the oracle (fp32, train-mode BatchNorm)."""
    from css_amd.models import MODELS
    conf = dict(attention_dim=512, attention_heads=8, linear_units=2048, num_blocks=1, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 3, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_spk=3, attention_dim=512, attention_heads=8,
                                                               linear_units=2048, num_blocks=1), 78))
    m = m.to(DEV).train()
    B, T = 2, 50
    f = rnd((B, T, 257), 15).abs().to(DEV)
    w = rnd((B, T, 4 * 257), 16, 0.1).abs().to(DEV)
    # a weighted energy of the masks: its gradient vanishes with the mask, so ReLU units whose sign differs between
    # the bf16 and the fp32 forward do not dominate the comparison (a plain linear functional reads 7e-2 on BOTH the
    # tuned 256-wide and this path)
    masks = m.forward_masks(f)
    assert type(masks.grad_fn).__name__ == "ConformerTrainBackward"
    (masks ** 2 * w).sum().backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    import grad_ref
    wc = w.cpu()
    ref = grad_ref.oracle_param_grads(m, lambda mk: (mk ** 2 * wc).sum(), f, training_bn=True)
    err, worst = grad_ref.compare(grads, ref)
    print(f"[parity] Conformer-large widths: HIP backward vs oracle autograd, global rel err {err:.3e}, worst {worst}")
    assert err < 3e-2


def test_eval_after_fused_adam_steps_sees_the_new_weights():
    """eval -> N FusedAdam steps -> eval: FusedAdam rewrites the parameter storage through a raw kernel (neither
    data_ptr nor Tensor._version changes), so the eval path's bf16 weight shadows must be re-packed; checked against a
    fresh model loaded from the trained state_dict.  Also through GraphedTrainStep (graph replay)."""
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.trainer import GraphedTrainStep
    obj = OBJECTIVES["MSE"].build_objective({})
    f = rnd((2, 40, 257), 14).abs().to(DEV)
    batch = {"mix": f, "source1": (rnd((2, 40, 257), 61).abs() * 0.5).to(DEV), "source2": (rnd((2, 40, 257), 62).abs() * 0.5).to(DEV)}
    for graphed in (False, True):
        m, _ = build(1, 47)
        m.eval()
        with torch.no_grad():
            before = m.forward_masks(f).clone()
        m.train()
        opt = FusedAdam(m.parameters(), lr=5e-3, max_norm=5.0)
        if graphed:
            stepper = GraphedTrainStep(m, obj, opt, batch, max_norm=5.0)
            for _ in range(3):
                stepper(batch)
        else:
            for _ in range(3):
                opt.zero_grad()
                obj(m, batch, device=DEV).backward()
                opt.step()
        m.eval()
        with torch.no_grad():
            after = m.forward_masks(f).clone()
        fresh, _ = build(1, 47)
        fresh.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
        fresh.eval()
        with torch.no_grad():
            want = fresh.forward_masks(f)
        assert float((after - before).abs().max()) > 1e-3, "training did not move the masks: the test is vacuous"
        assert torch.equal(after, want), f"stale weight shadows after training (graphed={graphed})"


def test_fused_adam_state_dict_is_torch_adam_compatible():
    """A checkpoint's ``optimizer`` entry written by either side loads in the other (train.py:209-225 resumes
    torch.optim.Adam; combine_models.py reads the per-parameter moments)."""
    from css_amd.optim import FusedAdam
    m, _ = build(1, 51)
    m.train()
    opt = FusedAdam(m.parameters(), lr=1e-3)
    for p in m.parameters():
        p.grad.copy_(torch.randn_like(p) * 0.01)
    opt.step()
    opt.step()
    sd = opt.state_dict()
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m.parameters()], lr=1e-3)
    ref.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})   # the reference's resume
    for p, q in zip(ref.param_groups[0]["params"], m.parameters()):
        assert int(ref.state[p]["step"]) == 2 and ref.state[p]["exp_avg"].shape == q.shape
    # and back: a torch.optim.Adam state (no "fused" entry) scattered into the flat moment buffers
    m2, _ = build(1, 51)
    opt2 = FusedAdam(m2.parameters(), lr=1e-3)
    opt2.load_state_dict(ref.state_dict())
    assert opt2._step == 2 and torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    with pytest.raises(ValueError):
        FusedAdam([{"params": list(m2.parameters())[:2]}, {"params": list(m2.parameters())[2:]}])


@pytest.mark.parametrize("B,T", [(3, 100), (2, 64), (5, 77), (1, 400), (32, 400)])
def test_forward_loss_fused_head_pit_matches_unfused(B, T):
    """model.forward_loss (csship_head_pit: mask head + mask application + PIT-MSE, masks never written) ==
    ops.pit_mse(model.forward_masks(f), f, targets): pair matrix, permutations and loss; panels that straddle two
    samples (T not a multiple of 64), a ragged last panel, swapped targets.  T < 64 takes the unfused route."""
    from css_amd import ops
    m, sd = build(1, 91)
    f = rnd((B, T, 257), 92).abs().to(DEV)
    tg = [(rnd((B, T, 257), 93 + k).abs() * 0.5).to(DEV) for k in range(2)]
    with torch.no_grad():
        masks = m.forward_masks(f)
        l0, p0, _, d0 = ops.pit_mse(masks, f, tg, want_pairs=True)
        # make half of the samples prefer the swapped assignment
        est = ops.apply_masks(masks, f, 2, 257)                      # [B,2,T,257]
        tg2 = [torch.where((torch.arange(B, device=DEV) % 2 == 0)[:, None, None], est[:, 1], est[:, 0]) * 0.9 + 0.01,
               torch.where((torch.arange(B, device=DEV) % 2 == 0)[:, None, None], est[:, 0], est[:, 1]) * 0.9 + 0.01]
        l1, p1, d1 = m.forward_loss(f, tg, want_pairs=True)
        l2, p2, _, d2 = ops.pit_mse(masks, f, tg2, want_pairs=True)
        l3, p3, d3 = m.forward_loss(f, tg2, want_pairs=True)
    assert torch.equal(p0, p1) and torch.equal(p2, p3)
    assert p3[:, 0].tolist() == [1 if b % 2 == 0 else 0 for b in range(B)]
    for a_, b_ in ((d0, d1), (d2, d3)):
        assert float((a_ - b_).abs().max()) < 2e-5 * float(a_.abs().max())
    assert abs(float(l0) - float(l1)) < 1e-5 * abs(float(l0)) and abs(float(l2) - float(l3)) < 1e-5 * abs(float(l2)) + 1e-9
    with torch.no_grad():
        l4, p4, _ = m.forward_loss(f, tg)
    assert torch.equal(l4, l1) and torch.equal(p4, p1)                # bit-reproducible (no atomics)


def test_forward_loss_short_windows_take_the_unfused_route():
    from css_amd import ops
    m, sd = build(1, 91)
    f = rnd((2, 40, 257), 95).abs().to(DEV)
    tg = [(rnd((2, 40, 257), 96 + k).abs() * 0.5).to(DEV) for k in range(2)]
    with torch.no_grad():
        l0 = ops.pit_mse(m.forward_masks(f), f, tg)[0]
        l1 = m.forward_loss(f, tg)[0]
    assert torch.equal(l0, l1)
