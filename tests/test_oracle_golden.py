"""Pin the CPU oracle against golden vectors captured from the reference (CPU only).

The reference has no tests/fixtures of its own; tests/golden/*.npz were
produced by importing it in the build container (tests/golden/make_golden.py).
"""
import json
import os
import types

import numpy as np
import pytest
import torch

import oracle
from css_amd.utils import synthetic as syn

from conftest import GOLDEN


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def close(a, b, atol, rtol=0.0):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} (tol {atol}+{rtol}*|b|)"


def test_keys_match_reference():
    keys = json.load(open(os.path.join(GOLDEN, "keys.json")))
    assert [[k, list(s)] for k, s in syn.conformer_shapes().items()] == keys["conformer_base"]
    assert [[k, list(s)] for k, s in syn.blstm_shapes().items()] == keys["blstm_default"]


def test_stft_features(golden):
    g = golden("stft.npz")
    n_short = 512 + 256 * 19
    mix, src = syn.synthetic_sources(2, n_short, seed=3)
    x7 = syn.synthetic_multichannel(src, n_ch=7, seed=3)
    assert abs(float(mix.double().sum()) - float(g["mix_sum"])) < 1e-9
    assert abs(float(x7.double().sum()) - float(g["x7_sum"])) < 1e-9
    mag, f, re, im = oracle.feature.feature_extractor(mix)
    close(re, g["re1"], 2e-5)
    close(im, g["im1"], 2e-5)
    close(mag, g["mag1"], 2e-5)
    close(f, g["f1"], 2e-4)
    mag7, f7, re7, im7 = oracle.feature.feature_extractor(x7, ipd_index="1,0;2,0;3,0;4,0;5,0;6,0")
    close(re7, g["re7"], 2e-5)
    close(im7, g["im7"], 2e-5)
    close(mag7, g["mag7"], 2e-5)
    close(f7[:, :257], g["f7"][:, :257], 2e-4)
    # IPD is an angle: compare on the circle (atan2 branch cut) and away from ill-conditioned points
    d = torch.as_tensor(g["f7"][:, 257:]).double() - f7[:, 257:].double()
    d = torch.atan2(torch.sin(d), torch.cos(d)).abs()
    assert float(d.mean()) < 1e-4 and float((d > 1e-2).double().mean()) < 1e-3
    mixw, _ = syn.synthetic_sources(1, 38656, seed=4)
    magw, fw, _, _ = oracle.feature.feature_extractor(mixw)
    assert magw.shape == (1, 257, 150)
    close(magw, g["magw"], 2e-5)
    close(fw, g["fw"], 2e-4)


def test_stft_matches_torch_stft():
    x = rnd((2, 4096), 5, 0.1)
    mag, pha, re, im = oracle.feature.stft(x)
    ref = torch.stft(x, 512, 256, window=torch.hann_window(512), center=False, return_complex=True)
    close(re, ref.real, 1e-5)
    close(im, ref.imag, 1e-5)


def test_istft_roundtrip():
    x = rnd((2, 256 * 20), 6, 0.1)
    spec = torch.stft(x, 512, 256, window=torch.hann_window(512), center=True, return_complex=True)
    y = oracle.feature.istft_center(spec.real, spec.imag, 512, 256, length=x.shape[-1])
    ref = torch.istft(spec, 512, 256, window=torch.hann_window(512), center=True, length=x.shape[-1])
    close(y, ref, 1e-5)
    close(y[:, 256:-256], x[:, 256:-256], 1e-5)


D, H, FFN, KS = 256, 4, 1024, 33


def _sd(shapes, seed, prefix=""):
    return {prefix + k: v for k, v in syn.fill_state_dict(shapes, seed).items()}


def test_conformer_submodules(golden):
    g = golden("conformer.npz")
    x = rnd((2, 64, D), 11)
    assert abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-9
    full = syn.conformer_shapes(num_blocks=1)
    pre = "conformer.encoders.0."
    ffn_shapes = {k[len(pre + "feed_forward_in."):]: s for k, s in full.items() if k.startswith(pre + "feed_forward_in.")}
    sd = _sd(ffn_shapes, 21, "ffn.")
    close(oracle.conformer.feed_forward(x, sd, "ffn"), g["ffn"], 2e-5)

    pe = syn.synth_tensor("pe_k.weight", (2000, D // H), 22)
    pos_k = oracle.conformer.rel_pos_table(pe, 64)
    mha_shapes = {k[len(pre + "self_attn."):]: s for k, s in full.items() if k.startswith(pre + "self_attn.")}
    sd = _sd(mha_shapes, 23, "mha.")
    close(oracle.conformer.attention(x, sd, "mha", H, pos_k), g["mha"], 2e-5)

    conv_shapes = {k[len(pre + "conv."):]: s for k, s in full.items() if k.startswith(pre + "conv.")}
    sd = _sd(conv_shapes, 24, "cv.")
    close(oracle.conformer.conv_module(x, sd, "cv", KS), g["conv_eval"], 2e-5)
    close(oracle.conformer.conv_module(x, sd, "cv", KS, training_bn=True), g["conv_trainbn"], 5e-5)

    lay_shapes = {k[len(pre):]: s for k, s in full.items() if k.startswith(pre)}
    sd = _sd(lay_shapes, 25, "l.")
    close(oracle.conformer.encoder_layer(x, sd, "l", H, KS, pos_k), g["layer"], 5e-5)


def _causal_encoder_sd():
    """state_dict of the reference's ConformerEncoder(257, 256, 4, 1024, 2, 33, causal=True) as make_golden_causal.py fills
    it (keys without the model prefix), re-keyed for oracle.conformer.encoder"""
    shapes = {k[len("conformer."):]: v for k, v in syn.conformer_shapes(num_blocks=2).items() if k.startswith("conformer.")}
    return {"conformer." + k: v for k, v in syn.fill_state_dict(shapes, 31).items()}


def test_causal_conv_module_and_encoder(golden):
    """The causal ConvModule (conformer.py:265-273, 297-298) - no recipe of the reference builds it, the class does -
    against outputs of the reference's own ConvModule(causal=True) / ConformerEncoder(causal=True)
    (tests/golden/make_golden_causal.py)."""
    g = golden("causal.npz")
    x = rnd((2, 64, D), 11)
    full = syn.conformer_shapes(num_blocks=1)
    pre = "conformer.encoders.0."
    conv_shapes = {k[len(pre + "conv."):]: s for k, s in full.items() if k.startswith(pre + "conv.")}
    sd = _sd(conv_shapes, 24, "cv.")
    close(oracle.conformer.conv_module(x, sd, "cv", KS, causal=True), g["conv_eval"], 2e-5)
    close(oracle.conformer.conv_module(x, sd, "cv", KS, training_bn=True, causal=True), g["conv_trainbn"], 5e-5)
    assert float((oracle.conformer.conv_module(x, sd, "cv", KS) - torch.from_numpy(g["conv_eval"])).abs().max()) > 1e-2
    f = rnd((2, 90, 257), 12).abs()
    esd = _causal_encoder_sd()
    assert sorted(k[len("conformer."):] for k in esd) == list(g["encoder2_keys"])
    close(oracle.conformer.encoder(oracle.conformer.mvn_time(f), esd, num_blocks=2, causal=True), g["encoder2"], 1e-4)


@pytest.mark.parametrize("nb", [2, 16])
def test_conformer_full(golden, nb):
    g = golden("conformer.npz")
    f_in = rnd((2, 100, 257), 12).abs()
    assert abs(float(f_in.double().sum()) - float(g["f_sum"])) < 1e-9
    tg = rnd((2, 2, 100, 257), 13).abs() * 0.5
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=nb), 30 + nb)
    y_pred, masks = oracle.conformer.forward(f_in, sd, num_blocks=nb)
    close(torch.stack(masks, 0), g[f"masks_b{nb}"], 2e-4)
    loss = oracle.pit.mse_objective(y_pred, tg[:, 0], tg[:, 1])
    assert abs(float(loss) - float(g[f"loss_b{nb}"])) < 1e-5 * max(1.0, float(g[f"loss_b{nb}"]))


def test_conformer_long_clamp(golden):
    g = golden("conformer.npz")
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=1), 41)
    f_long = rnd((1, 1100, 257), 14).abs()
    _, masks = oracle.conformer.forward(f_long, sd, num_blocks=1)
    m = torch.stack(masks, 0)
    close(m[:, :, :40], g["masks_long_head"], 2e-4)
    close(m[:, :, -40:], g["masks_long_tail"], 2e-4)
    assert int(g["seven_ch_runs"]) == 0  # reference forward raises for idim != num_bins (documented quirk)


def test_blstm(golden):
    g = golden("blstm.npz")
    sd = syn.fill_state_dict(syn.blstm_shapes(hidden_dim=64, num_layers=2), 50)
    fb = rnd((2, 30, 257), 16).abs()
    y_pred, masks = oracle.blstm.forward(fb, sd, num_layers=2)
    close(torch.stack(masks, 0), g["masks"], 5e-5)
    close(y_pred, g["y_pred"], 5e-5)


def test_pit(golden):
    g = golden("pit.npz")
    for K in (2, 3):
        est = rnd((K, 40, 257), 60 + K).abs()
        tgt = rnd((K, 40, 257), 70 + K).abs()
        l, p = oracle.pit.pit_mse(est, tgt, return_permutation=True)
        assert abs(float(l) - float(g[f"loss_k{K}"])) < 1e-6
        assert list(p) == list(g[f"perm_k{K}"])
        rot = list(range(1, K)) + [0]
        est2 = tgt[rot] + 0.01 * rnd((K, 40, 257), 80 + K)
        l2, p2 = oracle.pit.pit_mse(est2, tgt, return_permutation=True)
        assert abs(float(l2) - float(g[f"loss_rot_k{K}"])) < 1e-7
        assert list(p2) == list(g[f"perm_rot_k{K}"])
        # pairwise-SSE formulation (what the HIP kernel computes) gives the same minimum
        Dm = oracle.pit.pairwise_sse(est, tgt)
        import itertools
        cands = [sum(Dm[pp[i], i] for i in range(K)) / (K * 40 * 257) for pp in itertools.permutations(range(K))]
        assert abs(float(min(cands)) - float(l)) < 1e-6


def _stitch_inputs():
    W, Fb, Tw, hop = 6, 257, 150, 50
    total = hop * (W - 1) + Tw
    base = rnd((Fb, total, 3), 90).abs()
    base = base / base.sum(-1, keepdim=True)
    magl = rnd((Fb, total), 91).abs()
    flips = [0, 1, 1, 0, 1, 0]
    masks, mags = [], []
    for w in range(W):
        m = base[:, w * hop:w * hop + Tw].clone() + 0.02 * rnd((Fb, Tw, 3), 100 + w).abs()
        if flips[w]:
            m = m[:, :, [1, 0, 2]]
        masks.append(m)
        mags.append(magl[:, w * hop:w * hop + Tw])
    return torch.stack(masks), torch.stack(mags), hop


def test_stitcher(golden):
    g = golden("stitcher.npz")
    masks, mags, hop = _stitch_inputs()
    margin = oracle.stitcher.stitch_margin(2.4, 0.8, 256)
    assert margin == int(g["margin"]) == 100
    perm = oracle.stitcher.get_stitch(mags, masks, margin)
    assert np.array_equal(np.array(perm), g["perm"])
    r1, r2, rn = oracle.stitcher.get_connect(perm, masks, hop)
    close(r1, g["res1"], 1e-6)
    close(r2, g["res2"], 1e-6)
    close(rn, g["noise"], 1e-6)


def test_lr_trace():
    tr = json.load(open(os.path.join(GOLDEN, "trainer.json")))
    got = oracle.schedule.lr_trace(1e-3, 10, warmup=4, fixed=2, decay=0.1, min_lr=1e-9)
    assert np.allclose(got, tr["lr_trace"], rtol=1e-12, atol=0)


def test_mvdr_oracle_matches_reference_numpy_twin(golden):
    """oracle/mvdr.py (mask_scm, souden_coeffs, apply_bf) vs css/utils/mvdr_util.py outputs (tests/golden/mvdr.npz),
    and the asteroid-style restatement (normalised SCMs, eps on the trace; parity unpinned) agrees with it."""
    import numpy as np
    from oracle import mvdr
    g = golden("mvdr.npz")
    mix, masks = syn.mvdr_inputs()
    tgt, ms = mvdr.mask_scm(mix, masks[0])
    itf, _ = mvdr.mask_scm(mix, masks[1])
    noi, _ = mvdr.mask_scm(mix, masks[2])
    coef = mvdr.souden_coeffs(noi + itf, tgt)
    res = mvdr.apply_bf(mix, coef)
    assert np.abs(tgt - g["tgt_scm"]).max() < 1e-4 * np.abs(g["tgt_scm"]).max()
    assert abs(ms - float(g["mask_sum"])) < 1e-3
    assert np.abs(coef - g["coef"]).max() < 2e-5 and np.abs(res - g["res"]).max() < 1e-4
    # live-path formulation: scalar SCM normalisation cancels in w
    S = torch.from_numpy(mix)[None].to(torch.complex128)
    m = torch.from_numpy(masks).double()
    t2 = mvdr.compute_scm(S, m[0:1])
    n2 = mvdr.compute_scm(S, m[1:2] + m[2:3])   # same noise statistics as noi + itf up to the per-bin normaliser
    n_un = torch.einsum("bmft,bnft->bfmn", S * (m[1:2] + m[2:3])[:, None], S.conj())
    y = mvdr.souden_mvdr(S, t2, n_un / (m[1:2] + m[2:3]).sum(-1)[..., None, None], eps=0.0)[0].numpy()
    assert np.abs(y - g["res"]).max() < 2e-4
    assert np.isfinite(n2.real.numpy()).all()


# ------------------------------------------------------------------ properties of the oracle (hypothesis, small sizes)
def test_oracle_properties_pit_permutation_and_stft_linearity():
    """The checker itself obeys the domain's invariants: PIT loss is invariant under a joint relabelling of the estimates
    (the returned permutation maps through it) and under a relabelling of the targets; the STFT is linear."""
    import itertools
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=15, deadline=None)
    @given(seed=st.integers(0, 10_000), K=st.integers(2, 3), T=st.integers(1, 9), F=st.integers(1, 7))
    def pit(seed, K, T, F):
        r = np.random.RandomState(seed)
        est = torch.from_numpy(r.standard_normal((K, T, F)).astype(np.float32))
        tgt = torch.from_numpy(r.standard_normal((K, T, F)).astype(np.float32))
        l0, p0 = oracle.pit.pit_mse(est, tgt, return_permutation=True)
        for sigma in itertools.permutations(range(K)):
            ls, ps = oracle.pit.pit_mse(est[list(sigma)], tgt, return_permutation=True)
            assert abs(float(ls) - float(l0)) <= 1e-6 * max(1.0, abs(float(l0)))
            # est'[i] = est[sigma[i]]; est'[ps[i]] matched to tgt[i]  =>  est[sigma[ps[i]]] matched to tgt[i]
            assert abs(float(((est[[sigma[j] for j in ps]] - tgt) ** 2).mean()) - float(l0)) <= 1e-6 * max(1.0, abs(float(l0)))
            lt, _ = oracle.pit.pit_mse(est, tgt[list(sigma)], return_permutation=True)
            assert abs(float(lt) - float(l0)) <= 1e-6 * max(1.0, abs(float(l0)))

    @settings(max_examples=8, deadline=None)
    @given(seed=st.integers(0, 10_000), frames=st.integers(1, 5), a=st.floats(-2, 2), b=st.floats(-2, 2))
    def lin(seed, frames, a, b):
        r = np.random.RandomState(seed)
        n = 512 + 256 * (frames - 1)
        x = torch.from_numpy(r.standard_normal((2, n)).astype(np.float32))
        y = torch.from_numpy(r.standard_normal((2, n)).astype(np.float32))
        sx, sy, sz = (oracle.feature.stft(v) for v in (x, y, a * x + b * y))
        for i in (2, 3):  # re, im
            assert float((sz[i] - (a * sx[i] + b * sy[i])).abs().max()) < 1e-3

    pit()
    lin()


def test_oracle_training_forward_with_dropout_matches_reference(golden):
    """Train-mode forwards (batch-statistics BatchNorm + dropout) of the oracle vs the reference run with its nn.Dropout
    layers applying the build's counter-based masks (tests/golden/make_golden_dropout.py): pins WHERE and on WHAT the
    dropout acts (embed before the ReLU, hidden / output of both FeedForwards, attention probabilities and output,
    ConvModule output; BLSTM: embed and every layer)."""
    from css_amd.dropout import stream_of
    g = golden("dropout.npz")
    drop = oracle.dropout.Masks(int(g["seed"]), float(g["p"]), stream_of)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=2), 61)
    f = rnd((3, 44, 257), 62).abs()
    _, masks = oracle.conformer.forward(f, sd, num_blocks=2, training_bn=True, drop=drop)
    ref = torch.from_numpy(g["conformer_masks"])
    assert float((torch.stack(masks) - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))
    # and dropout matters: the eval-style forward differs visibly
    _, masks0 = oracle.conformer.forward(f, sd, num_blocks=2, training_bn=True)
    assert float((torch.stack(masks0) - ref).abs().max()) > 1e-2
    sdb = syn.fill_state_dict(syn.blstm_shapes(hidden_dim=64, num_layers=2), 63)
    fb = rnd((2, 30, 257), 64).abs()
    _, bm = oracle.blstm.forward(fb, sdb, num_layers=2, drop=drop)
    refb = torch.from_numpy(g["blstm_masks"])
    assert float((torch.stack(bm) - refb).abs().max()) < 2e-4 * max(1.0, float(refb.abs().max()))


def test_dropout_mask_statistics():
    """keep rate of the counter-based mask = 1 - round(p * 65536) / 65536; different sites / seeds are uncorrelated."""
    k1 = oracle.dropout.keep_mask(7, 17, 200000, 0.1)
    k2 = oracle.dropout.keep_mask(7, 18, 200000, 0.1)
    k3 = oracle.dropout.keep_mask(8, 17, 200000, 0.1)
    assert abs(k1.mean() - (1 - 6554 / 65536)) < 3e-3
    for a, b in ((k1, k2), (k1, k3)):
        assert abs(np.corrcoef(a, b)[0, 1]) < 1e-2
    assert oracle.dropout.keep_mask(7, 17, 100, 0.0).all()


def test_doa_merge_golden(golden):
    """oracle/doa.py vs the reference's own Separator.steervec_7ch / doa_likelihood / angle_merge
    (tests/golden/make_golden_separator.py: separator.py:113-250 run with a stub onnxruntime module)."""
    g = golden("separator.npz")
    sv, ang = oracle.doa.steervec_7ch(257, 30)
    ref_sv = torch.complex(torch.from_numpy(g["steervec_re"]), torch.from_numpy(g["steervec_im"]))
    assert float((sv - ref_sv).abs().max()) < 1e-6 and torch.equal(ang, torch.from_numpy(g["angles"]))
    lo, hi = (int(v) for v in g["lo_hi"])
    assert (lo, hi) == (2, 64)   # 80 Hz / 31.25 Hz floor, 2000 Hz / 31.25 Hz ceil (separator.py:59-61)
    X, mask = syn.doa_inputs(ref_sv)
    like = oracle.doa.doa_likelihood(X, sv, (mask > 0.5).float(), compression=0.5, lowcut=lo, highcut=hi, softmax=False)
    close(like, g["like"], 0.0, 1e-5)
    sm = oracle.doa.doa_likelihood(X, sv, mask, compression=1 / 3, lowcut=lo, highcut=hi, softmax=True)
    close(sm, g["like_softmax"], 1e-6, 1e-4)
    merged, _ = oracle.doa.angle_merge(X, mask, sv, ang, lo, hi, thresh=16)
    assert torch.equal(merged, torch.from_numpy(g["merged"]))
    # quirk 6 is in the golden: window 0 (same direction) has OUTPUT 0 silenced, window 1 is untouched
    assert float(merged[0, ..., 0].max()) == pytest.approx(1e-12) and torch.equal(merged[1], mask[1])


def test_separator_windowing_golden(golden):
    """oracle/separator.py vs the reference's Separator.separate (separator.py:63-111) on a 7-channel recording with the
    DOA merge and a 1-channel one, the ONNX session replaced on both sides by synthetic.FakeMaskSession."""
    g = golden("separator.npz")
    n = 12800 * 2 + 38656 + 777
    mix, src = syn.synthetic_sources(1, n, seed=5)
    rec7 = syn.synthetic_multichannel(src, n_ch=7, seed=5)[0]
    m7, g7 = oracle.separator.separate(rec7, syn.FakeMaskSession(), ipd="1,0;2,0;3,0;4,0;5,0;6,0", merge=True)
    assert m7.shape == (3, 257, 150, 3) and float(m7.max()) <= 1.0
    close(g7, g["sep7_mags"], 2e-5 * float(np.abs(g["sep7_mags"]).max()))
    # sin(k * f) of the MVN'd feature: 1e-4 feature noise times k <= 3 - except in the top bins, where the synthetic
    # sources' 3-tap shaping filter has its zero (|X| ~ 3e-4 of the peak): there the 2e-5 absolute difference between
    # an FFT and the reference's DFT matrix is several per cent of the bin's own standard deviation
    close(m7[:, :250], g["sep7_masks"][:, :250], 2e-3)
    close(m7, g["sep7_masks"], 6e-2)
    m1, g1 = oracle.separator.separate(mix, syn.FakeMaskSession())
    close(g1, g["sep1_mags"], 2e-5 * float(np.abs(g["sep1_mags"]).max()))
    close(m1[:, :250], g["sep1_masks"][:, :250], 2e-3)
    close(m1, g["sep1_masks"], 6e-2)


def test_combine_models_golden(golden, tmp_path):
    """css_amd.utils.checkpoint.combine_models vs the output of the reference's scripts/python/combine_models.py on three
    job checkpoints (tests/golden/combine.npz): f32 accumulation order of :73-76 and the 0-dim buffer rule of :77-78
    (last job's counter x 1/n, truncated: 500, 500, 501 -> 167)."""
    from css_amd.models import MODELS
    from css_amd.utils import checkpoint as ck
    g = golden("combine.npz")
    conf = dict(attention_dim=64, attention_heads=1, linear_units=64, num_blocks=1, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    paths = []
    for j, seed in enumerate(g["seeds"]):
        torch.manual_seed(int(seed))
        m = MODELS["Conformer"](257, 257, 2, 1, conf)   # same parameter creation order as the reference class
        sd = m.state_dict()
        for k in sd:
            if sd[k].dim() == 0:
                sd[k].fill_(int(g["tracked"][j]))
        p = str(tmp_path / f"1.{j + 1}.mdl")
        torch.save({"model": sd, "objective": {}, "optimizer": None, "lr_sched": None, "epoch": 1}, p)
        paths.append(p)
    res = torch.load(ck.combine_models(str(tmp_path / "1.mdl"), paths))["model"]
    i = 0
    while f"key{i}" in g:
        k, ref = str(g[f"key{i}"]), torch.from_numpy(np.asarray(g[f"val{i}"]))
        assert res[k].dtype == ref.dtype and torch.equal(res[k], ref), k
        i += 1
    assert i >= 2 and int(res["conformer.encoders.0.conv.BN.num_batches_tracked"]) == 167
