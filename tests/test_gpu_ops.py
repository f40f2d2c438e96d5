"""GPU parity tests: every HIP op (called through the C ABI of libcsship.so) against the CPU
oracle on the same seeded inputs, plus the committed golden vectors of the reference.

Tolerances are stated per test.  GEMM-class ops compute in bf16 with fp32 accumulation, so
they are checked (a) tightly against a float64 evaluation of the SAME bf16-rounded operands
and (b) against the fp32 oracle within a bf16-sized tolerance.
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


def bf(x):
    """round-trip through bf16 (what the kernels see)"""
    return x.to(torch.bfloat16).float()


def err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from css_amd import ops as o
    o.lib()  # raises if libcsship.so is missing: no fallback
    return o


# ----------------------------------------------------------------------------- STFT / features
@pytest.mark.parametrize("shape", [(3, 512 + 256 * 70), (2, 3, 512 + 256 * 33), (1, 38656), (2, 512)])
def test_stft_vs_oracle(ops, shape):
    n = shape[-1]
    x = rnd(shape, 5, 0.1)
    o = ops.stft(x.to(DEV), ("mag", "re", "im", "pha"))
    mag, pha, re, im = oracle.feature.stft(x)  # [..., F, T]
    scale = float(mag.max())
    # fp32 FFT vs fp32 dense DFT: both ~1e-6 relative to the spectrum scale
    assert err(o["re"].transpose(-1, -2), re) < 2e-5 * scale
    assert err(o["im"].transpose(-1, -2), im) < 2e-5 * scale
    assert err(o["mag"].transpose(-1, -2), mag) < 2e-5 * scale
    # phase only where the bin carries energy (atan2 is ill-conditioned near 0) and modulo 2 pi
    d = o["pha"].transpose(-1, -2).cpu().double() - pha.double()
    d = torch.atan2(torch.sin(d), torch.cos(d)).abs()
    strong = mag > 1e-2 * scale
    assert float(d[strong].max()) < 1e-3
    assert o["mag"].shape[-2] == (n - 512) // 256 + 1


def test_stft_strided_windows(ops):
    """unfold()-style overlapping windows (separator.py:67) are read in place."""
    rec = rnd((2, 12800 * 4 + 38656), 7, 0.1).to(DEV)
    win = rec.unfold(-1, 38656, 12800)  # [2, W, 38656] view, stride (n, 12800, 1)
    a = ops.stft(win, ("mag",))["mag"]
    b = ops.stft(win.contiguous(), ("mag",))["mag"]
    assert a.shape == (2, 5, 150, 257) and torch.equal(a, b)


def test_stft_padded_rows_and_fused_loss_on_them(ops):
    """ops.stft(pad_rows=True): [..., :257] views of 260-float (16-byte aligned) rows - the same values bit for bit - and the
    fused mask head + PIT loss reading the mixture AND its targets through that row stride (csship_head_pit's ld_f / ld_t)
    gives the same loss, permutations and pair matrix as on dense rows."""
    from css_amd.models import MODELS
    B, T = 3, 100
    mix, src = syn.synthetic_sources(B, 512 + 256 * (T - 1), seed=9)
    audio = torch.stack([mix, src[0], src[1]], 0).to(DEV)
    dense = ops.stft(audio, ("mag", "re"))
    padded = ops.stft(audio, ("mag", "re"), pad_rows=True)
    assert padded["mag"].shape == (3, B, T, 257) and padded["mag"].stride(-2) == 260 and padded["mag"].stride(-1) == 1
    assert torch.equal(padded["mag"], dense["mag"]) and torch.equal(padded["re"], dense["re"])
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=1, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=1), 5))
    m = m.to(DEV).eval()
    with torch.no_grad():
        a = m.forward_loss(dense["mag"][0], [dense["mag"][1], dense["mag"][2]], want_pairs=True)
        b = m.forward_loss(padded["mag"][0], [padded["mag"][1], padded["mag"][2]], want_pairs=True)
        c = m.forward_loss(padded["mag"][0], [padded["mag"][1], dense["mag"][2]], want_pairs=True)   # mixed strides: dense copies
    for x in (b, c):
        assert torch.equal(a[0], x[0]) and torch.equal(a[1], x[1]) and torch.equal(a[2], x[2])


def test_feature_extractor_golden(ops, golden):
    from css_amd.executor.feature import FeatureExtractor
    g = golden("stft.npz")
    mix, _ = syn.synthetic_sources(2, 512 + 256 * 19, seed=3)
    fe = FeatureExtractor(512, 256)
    mag, f, re, im = fe.forward(mix.to(DEV))
    assert err(mag, torch.from_numpy(g["mag1"])) < 5e-5
    assert err(re, torch.from_numpy(g["re1"])) < 5e-5
    assert err(im, torch.from_numpy(g["im1"])) < 5e-5
    assert err(f, torch.from_numpy(g["f1"])) < 5e-4
    mixw, _ = syn.synthetic_sources(1, 38656, seed=4)
    magw, fw, _, _ = fe.forward(mixw.to(DEV))
    assert magw.shape == (1, 257, 150)
    assert err(magw, torch.from_numpy(g["magw"])) < 5e-5
    assert err(fw, torch.from_numpy(g["fw"])) < 5e-4


def test_feature_extractor_7ch_ipd_golden(ops, golden):
    from css_amd.executor.feature import FeatureExtractor
    g = golden("stft.npz")
    _, src = syn.synthetic_sources(2, 512 + 256 * 19, seed=3)
    x7 = syn.synthetic_multichannel(src, n_ch=7, seed=3)
    fe = FeatureExtractor(512, 256, ipd_index="1,0;2,0;3,0;4,0;5,0;6,0")
    mag, f, re, im = fe.forward(x7.to(DEV))
    assert f.shape == (2, 1799, 20) and re.shape == (2, 7, 257, 20)
    assert err(mag, torch.from_numpy(g["mag7"])) < 5e-5
    assert err(re, torch.from_numpy(g["re7"])) < 5e-5 and err(im, torch.from_numpy(g["im7"])) < 5e-5
    assert err(f[:, :257], torch.from_numpy(g["f7"][:, :257])) < 5e-4
    # IPD is an angle: compare on the circle.  atan2(yi - mean, yr - mean) is ill-conditioned where the centred
    # vector is short (angle error ~ rounding / radius), so the bound is on error x radius, with the radius
    # taken from the oracle; plus a plain bound on all well-conditioned points.
    d = f[:, 257:].cpu().double() - torch.from_numpy(g["f7"][:, 257:]).double()
    d = torch.atan2(torch.sin(d), torch.cos(d)).abs()
    _, pha, _, _ = oracle.feature.stft(x7)
    li, ri = oracle.feature.parse_ipd_index("1,0;2,0;3,0;4,0;5,0;6,0")
    dl = (pha[:, li] - pha[:, ri]).double()
    c, sn = torch.cos(dl), torch.sin(dl)
    radius = torch.sqrt((c - c.mean(-1, keepdim=True)) ** 2 + (sn - sn.mean(-1, keepdim=True)) ** 2).reshape(2, -1, 20)
    # measured: median 3e-7, max 9e-5 (low-magnitude bins: phase error = STFT abs error / |X|)
    e = d * radius
    assert float(e.flatten().quantile(0.99)) < 2e-5 and float(e.max()) < 3e-4
    assert float(d[radius > 1e-2].max()) < 2e-2


@pytest.mark.parametrize("T,length", [(20, None), (152, 38656), (33, 5000)])
def test_istft(ops, T, length):
    n = 256 * (T - 1)
    x = rnd((3, n), 77, 0.1)
    spec = torch.stft(x, 512, 256, window=torch.hann_window(512), center=True, return_complex=True)  # [3,257,T]
    assert spec.shape[-1] == T
    y = ops.istft(spec.real.contiguous().to(DEV), spec.imag.contiguous().to(DEV), length)
    ref = torch.istft(spec, 512, 256, window=torch.hann_window(512), center=True, length=length)
    assert y.shape == ref.shape
    assert err(y, ref) < 2e-6 * 512  # f32 FFT round-off on signals of amplitude ~0.1
    assert err(y, oracle.feature.istft_center(spec.real, spec.imag, 512, 256, length=y.shape[-1])) < 1e-5


def test_mvn_stats(ops):
    x = rnd((3, 77, 257), 8).abs()
    mean, rinv = ops.mvn_stats(x.to(DEV))
    assert err(mean, x.mean(1)) < 1e-5
    assert err(rinv, 1.0 / (x.std(1) + oracle.feature.EPS)) < 1e-4


@pytest.mark.parametrize("T", [2, 16, 400, 512, 513, 700])
def test_mvn_stats_lengths(ops, T):
    """rows kept in registers up to 512 frames, the rest re-read; T = 2 leaves most waves without a row"""
    x = rnd((2, T, 257), 80 + T).abs() + 0.01
    mean, rinv = ops.mvn_stats(x.to(DEV))
    assert err(mean, x.mean(1)) < 1e-5
    assert err(1.0 / rinv, x.std(1) + oracle.feature.EPS) < 1e-5   # (1 / std itself is unbounded for two near-equal frames)
    again = ops.mvn_stats(x.to(DEV))
    assert torch.equal(again[0], mean) and torch.equal(again[1], rinv)


# ----------------------------------------------------------------------------- LayerNorm / GEMM
def test_layernorm(ops):
    x = rnd((301, 256), 9) * 2 + 0.5
    g, b = 1 + 0.1 * rnd((256,), 10), 0.1 * rnd((256,), 11)
    ob, of = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), out_bf16=True, out_f32=True)
    ref = torch.nn.functional.layer_norm(x, (256,), g, b, 1e-5)
    assert err(of, ref) < 1e-5
    assert err(ob.float(), bf(ref)) < 2e-2  # one bf16 ulp at |y| <= 4


def _ref_gemm(a, w, bias):
    return (bf(a).double() @ bf(w).double().t() + bias.double()).float()


@pytest.mark.parametrize("M", [64, 237, 1000])
def test_gemm_bf16_relu_bf16out(ops, M):
    a, w, bias = rnd((M, 256), 20), rnd((1024, 256), 21, 1 / 16), rnd((1024,), 22, 0.1)
    out = ops.gemm(a.to(DEV).to(torch.bfloat16), ops.pad_weight_bf16(w.to(DEV)), bias.to(DEV), relu=True, out_bf16=True)
    ref = torch.relu(_ref_gemm(a, w, bias))
    assert out.dtype == torch.bfloat16 and out.shape == (M, 1024)
    assert err(out.float(), ref) < 2e-2 * max(1.0, float(ref.abs().max()))  # bf16 output rounding
    assert err(out.float(), bf(ref)) < 4e-2


@pytest.mark.parametrize("M", [128, 333])
def test_gemm_k1024_residual(ops, M):
    a, w, bias, res = rnd((M, 1024), 23), rnd((256, 1024), 24, 1 / 32), rnd((256,), 25, 0.1), rnd((M, 256), 26)
    out = ops.gemm(a.to(DEV).to(torch.bfloat16), ops.pad_weight_bf16(w.to(DEV)), bias.to(DEV), residual=res.to(DEV), alpha=0.5)
    ref = res + 0.5 * _ref_gemm(a, w, bias)
    assert err(out, ref) < 2e-4  # fp32 accumulate of identical bf16 operands


def test_gemm_residual_layernorm_inplace(ops):
    M = 200
    a, w, bias, res = rnd((M, 1024), 27), rnd((256, 1024), 28, 1 / 32), rnd((256,), 29, 0.1), rnd((M, 256), 30)
    g, b = 1 + 0.1 * rnd((256,), 31), 0.1 * rnd((256,), 32)
    x = res.to(DEV).clone()
    out = ops.gemm(a.to(DEV).to(torch.bfloat16), ops.pad_weight_bf16(w.to(DEV)), bias.to(DEV), residual=x, alpha=0.5,
                   ln=(g.to(DEV), b.to(DEV)), out=x)
    ref = torch.nn.functional.layer_norm(res + 0.5 * _ref_gemm(a, w, bias), (256,), g, b, 1e-5)
    assert out.data_ptr() == x.data_ptr()
    assert err(out, ref) < 5e-4


def test_gemm_f32_a_head(ops):
    M = 150
    a, w, bias = rnd((M, 256), 33), rnd((771, 256), 34, 1 / 16), rnd((771,), 35, 0.1)
    out = ops.gemm(a.to(DEV), ops.pad_weight_bf16(w.to(DEV)), bias.to(DEV), relu=True)
    ref = torch.relu(_ref_gemm(a, w, bias))
    assert out.shape == (M, 771)
    assert err(out, ref) < 2e-4


def test_gemm_embed_mvn_ln_relu(ops):
    B, T, Fdim = 3, 50, 257
    f = rnd((B, T, Fdim), 36).abs()
    w, bias = rnd((256, Fdim), 37, 1 / 16), rnd((256,), 38, 0.1)
    g, b = 1 + 0.1 * rnd((256,), 39), 0.1 * rnd((256,), 40)
    fd = f.to(DEV)
    mean, rinv = ops.mvn_stats(fd)
    out = ops.gemm(fd.view(B * T, Fdim), ops.pad_weight_bf16(w.to(DEV)), bias.to(DEV), K=Fdim, relu=True,
                   ln=(g.to(DEV), b.to(DEV)), mvn=(mean, rinv), rows_per_batch=T)
    fn = oracle.conformer.mvn_time(f).reshape(B * T, Fdim)
    ref = torch.relu(torch.nn.functional.layer_norm(_ref_gemm(fn, w, bias), (256,), g, b, 1e-5))
    assert err(out, ref) < 3e-2  # bf16 rounding of the normalised input happens after a slightly different fp32 MVN
    fp32 = torch.relu(torch.nn.functional.layer_norm(fn @ w.t() + bias, (256,), g, b, 1e-5))
    assert float(((out.cpu() - fp32) ** 2).mean()) < 1e-4


# ----------------------------------------------------------------------------- row-panel kernels
def _ln_ref(x, g, b):
    return torch.nn.functional.layer_norm(x, (256,), g, b, 1e-5)


@pytest.mark.parametrize("M", [64, 200, 1000])
@pytest.mark.parametrize("final_ln", [False, True])
def test_ffn_fused(ops, M, final_ln):
    x = rnd((M, 256), 90) * 1.5
    g, b = 1 + 0.1 * rnd((256,), 91), 0.1 * rnd((256,), 92)
    w1, b1 = rnd((1024, 256), 93, 1 / 16), rnd((1024,), 94, 0.1)
    w2, b2 = rnd((256, 1024), 95, 1 / 32), rnd((256,), 96, 0.1)
    fg, fb = 1 + 0.1 * rnd((256,), 97), 0.1 * rnd((256,), 98)
    d = lambda t: t.to(DEV)  # noqa: E731
    xd = d(x).clone()
    out = ops.ffn_fused(xd, (d(g), d(b)), ops.pack_frag(d(w1)), d(b1), ops.pack_frag(d(w2)), d(b2), 1024, 0.5,
                        final_ln=(d(fg), d(fb)) if final_ln else None, out=xd)  # in place, like the encoder
    h = bf(torch.relu(_ref_gemm(_ln_ref(x, g, b), w1, b1)))
    ref = x + 0.5 * _ref_gemm(h, w2, b2)
    if final_ln:
        ref = _ln_ref(ref, fg, fb)
    assert out.data_ptr() == xd.data_ptr()
    assert err(out, ref) < 2e-3  # same bf16 roundings as the kernel; fp32 accumulation order differs
    fp32 = x + 0.5 * (torch.relu(_ln_ref(x, g, b) @ w1.t() + b1) @ w2.t() + b2)
    if final_ln:
        fp32 = _ln_ref(fp32, fg, fb)
    assert float(((out.cpu() - fp32) ** 2).mean()) < 1e-4


@pytest.mark.parametrize("M,N", [(130, 768), (64, 256), (333, 771)])
def test_panel_linear(ops, M, N):
    x = rnd((M, 256), 100) * 1.5
    g, b = 1 + 0.1 * rnd((256,), 101), 0.1 * rnd((256,), 102)
    w, bias = rnd((N, 256), 103, 1 / 16), rnd((N,), 104, 0.1)
    d = lambda t: t.to(DEV)  # noqa: E731
    wp = ops.pack_frag(d(w))
    # LN prologue, bf16 out (QKV projection)
    out = ops.panel_linear(d(x), wp, d(bias), N, ln=(d(g), d(b)), out_bf16=True)
    ref = _ref_gemm(_ln_ref(x, g, b), w, bias)
    assert out.shape == (M, N) and err(out.float(), bf(ref)) < 4e-2
    # f32 prologue + ReLU, f32 out (mask head; odd N exercises the scalar-store tail)
    out = ops.panel_linear(d(x), wp, d(bias), N, relu=True)
    assert err(out, torch.relu(_ref_gemm(x, w, bias))) < 5e-4
    if N == 256:  # bf16 prologue + residual in place (attention output projection)
        a = rnd((M, 256), 105)
        res = d(x).clone()
        out = ops.panel_linear(d(a).to(torch.bfloat16), wp, d(bias), N, residual=res, alpha=1.0, out=res)
        assert err(out, x + _ref_gemm(a, w, bias)) < 5e-4


def test_encoder_chain(ops):
    """FFN_out + LN -> FFN_in -> LN + QKV projection in one kernel == the three separate kernels."""
    M = 300
    x = rnd((M, 256), 110) * 1.5
    d = lambda t: t.to(DEV)  # noqa: E731
    def ffn_w(seed):
        return ((d(1 + 0.1 * rnd((256,), seed)), d(0.1 * rnd((256,), seed + 1))), ops.pack_frag(d(rnd((1024, 256), seed + 2, 1 / 16))),
                d(rnd((1024,), seed + 3, 0.1)), ops.pack_frag(d(rnd((256, 1024), seed + 4, 1 / 32))), d(rnd((256,), seed + 5, 0.1)))
    fa, fb = ffn_w(120), ffn_w(130)
    fin = (d(1 + 0.1 * rnd((256,), 140)), d(0.1 * rnd((256,), 141)))
    qln = (d(1 + 0.1 * rnd((256,), 142)), d(0.1 * rnd((256,), 143)))
    wq, bq = ops.pack_frag(d(rnd((768, 256), 144, 1 / 16))), d(rnd((768,), 145, 0.1))
    xo, qkv = ops.encoder_chain(d(x), 1024, ffn_a=fa, final_ln=fin, ffn_b=fb, qkv=(qln, wq, bq, 768))
    r1 = ops.ffn_fused(d(x), fa[0], fa[1], fa[2], fa[3], fa[4], 1024, 0.5, final_ln=fin)
    r2 = ops.ffn_fused(r1, fb[0], fb[1], fb[2], fb[3], fb[4], 1024, 0.5)
    rq = ops.panel_linear(r2, wq, bq, 768, ln=qln, out_bf16=True)
    # the two paths round differently (hidden-chunk rotation, LayerNorm statistics combined across waves): they agree to
    # bf16-operand noise, and against a float64 evaluation of the same chain the fused kernel is no further off
    assert err(xo, r2) < 5e-3
    assert err(qkv.float(), rq.float()) < 6e-2  # bf16 outputs of O(4) values: one ulp = 3e-2
    assert float(((qkv.float() - rq.float()) ** 2).mean()) < 1e-5

    # the form the model runs: LayerNorm affines folded into W1 / Wqkv (the prologues read no parameters) and the
    # q | k | v rows in the 8-feature interleave (16-byte stores); same function up to bf16 rounding points
    from css_amd import _lib

    def folded(f, seed):
        w1, b1 = ops.fold_layernorm(d(rnd((1024, 256), seed + 2, 1 / 16)), f[2], f[0][0], f[0][1])
        return (None, ops.pack_frag(w1), b1.contiguous(), f[3], f[4])
    wqf, bqf = ops.fold_layernorm(d(rnd((768, 256), 144, 1 / 16)), bq, qln[0], qln[1])
    order = ops.il8_row_order(768, DEV)
    xo2, qkv2 = ops.encoder_chain(d(x), 1024, ffn_a=folded(fa, 120), final_ln=fin, ffn_b=folded(fb, 130),
                                  qkv=(None, ops.pack_frag(wqf[order]), bqf[order].contiguous(), 768), flags=_lib.CHAIN_QKV_IL8)
    assert err(xo2, xo) < 1.5e-2   # max abs on O(1) values: two chains of bf16 products with different rounding points
    assert float(((qkv2.float() - qkv.float()) ** 2).mean()) < 4e-5 and err(qkv2.float(), qkv.float()) < 6e-2

    def ffn64(v, seed, fin_ln=None):
        g, b = (1 + 0.1 * rnd((256,), seed)).double(), (0.1 * rnd((256,), seed + 1)).double()
        w1, b1 = bf(rnd((1024, 256), seed + 2, 1 / 16)).double(), rnd((1024,), seed + 3, 0.1).double()
        w2, b2 = bf(rnd((256, 1024), seed + 4, 1 / 32)).double(), rnd((256,), seed + 5, 0.1).double()
        a = torch.nn.functional.layer_norm(v, (256,), g, b)
        v = v + 0.5 * (torch.relu(a @ w1.t() + b1) @ w2.t() + b2)
        return torch.nn.functional.layer_norm(v, (256,), fin_ln[0].double().cpu(), fin_ln[1].double().cpu()) if fin_ln else v
    x64 = ffn64(ffn64(x.double(), 120, fin), 130)
    e_fused, e_sep = err(xo, x64), err(r2, x64)
    assert e_fused < 1.5 * e_sep + 1e-3, (e_fused, e_sep)
    assert err(xo2, x64) < 1.5 * e_sep + 1e-3
    # q | k | v against float64 on the float64 residual stream: the folded / interleaved form is as close as the plain one
    # (UNROUNDED projection weights: both forms round their own weight matrix to bf16 - W or W diag(gamma))
    q64 = torch.nn.functional.layer_norm(x64, (256,), qln[0].double().cpu(), qln[1].double().cpu()) @ \
        rnd((768, 256), 144, 1 / 16).double().t() + rnd((768,), 145, 0.1).double()
    mse_plain = float(((qkv.float().cpu().double() - q64) ** 2).mean())
    mse_fold = float(((qkv2.float().cpu().double() - q64) ** 2).mean())
    assert mse_fold < 1.5 * mse_plain + 1e-6, (mse_fold, mse_plain)


# ----------------------------------------------------------------------------- attention
def _ref_attention(q, k, v, pe, maxlen=1000):
    """q (pre-scaled), k, v: [B,H,T,64] float; pe [2*maxlen,64] -> [B,T,H*64] (float64)."""
    B, H, T, dk = q.shape
    pos_k = oracle.conformer.rel_pos_table(pe, T, maxlen).double()
    s = torch.matmul(q.double(), k.double().transpose(-1, -2)) + torch.einsum("bhid,ijd->bhij", q.double(), pos_k)
    att = torch.softmax(s, dim=-1)
    return torch.matmul(att, v.double()).transpose(1, 2).reshape(B, T, H * dk)


@pytest.mark.parametrize("B,T", [(2, 100), (1, 64), (3, 17), (2, 400), (1, 1100), (1, 2300)])
def test_mha_relpos(ops, B, T):
    H = 4
    qkv = rnd((B, T, 3 * H * 64), 41)
    qkv[..., : H * 64] *= 0.125
    pe = rnd((2000, 64), 42, 0.3)
    out = ops.mha_relpos(qkv.to(DEV).to(torch.bfloat16), ops.pack_relpos(pe.to(DEV)), H)
    qb = bf(qkv).view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = _ref_attention(qb[0], qb[1], qb[2], bf(pe))
    # P is rounded to bf16 before P.V and the output is stored in bf16: ~2^-8 relative
    assert err(out.float(), ref) < 3e-2


@pytest.mark.parametrize("B,T", [(2, 64), (3, 100), (1, 7), (2, 400), (1, 1100)])
def test_mha_relpos_log2_domain(ops, B, T):
    """CSSHIP_MHA_Q_LOG2E (the inference forward's form): q carries log2(e) / sqrt(d_k), the band accumulators start at
    minus the running reference and the exponentials take the accumulators as they are.  Same softmax: checked against
    the float64 attention of the UNSCALED problem, and against the explicit-scaling kernel on the same values.  The
    second case plants scores that grow from tile to tile (every tile moves the reference: the rescale branch) and one
    very large score at the last key."""
    H = 4
    LOG2E = 1.4426950408889634
    for grow in (False, True):
        qkv = rnd((B, T, 3 * H * 64), 141)
        qkv[..., : H * 64] *= 0.125
        if grow:   # every query shares a direction u with every key, and key j's share grows with j: + 30 j / T per score
            u = rnd((64,), 143)
            u = (u / u.norm()).repeat(H)
            qkv[..., : H * 64] += 0.5 * u
            qkv[..., H * 64: 2 * H * 64] += 60.0 * torch.linspace(0, 1, T)[None, :, None] * u
            qkv[:, T - 1, H * 64: 2 * H * 64] *= 3.0
        pe = rnd((2000, 64), 142, 0.3)
        q2 = qkv.clone()
        q2[..., : H * 64] *= LOG2E
        pk = ops.pack_relpos(pe.to(DEV))
        out2 = ops.mha_relpos(q2.to(DEV).to(torch.bfloat16), pk, H, q_log2e=True)
        # reference on exactly the operands the kernel sees: bf16(q log2e) / log2e
        qb2 = bf(q2).view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref = _ref_attention(qb2[0] / LOG2E, qb2[1], qb2[2], bf(pe))
        assert bool(torch.isfinite(out2.float()).all())
        assert err(out2.float(), ref) < 3e-2, (grow, err(out2.float(), ref))
        out1 = ops.mha_relpos(qkv.to(DEV).to(torch.bfloat16), pk, H)
        assert err(out2.float(), out1.float()) < 4e-2   # (q is rounded to bf16 after a different factor)


# ----------------------------------------------------------------------------- conv module
def _conv_weights(seed):
    full = syn.conformer_shapes(num_blocks=1)
    pre = "conformer.encoders.0.conv."
    shapes = {k[len(pre):]: s for k, s in full.items() if k.startswith(pre)}
    sd = {"cv." + k: v for k, v in syn.fill_state_dict(shapes, seed).items()}
    dev = {
        "ln_g": sd["cv.layer_norm.weight"], "ln_b": sd["cv.layer_norm.bias"],
        "pw1_w": sd["cv.pw_conv_1.weight"].reshape(2), "pw1_b": sd["cv.pw_conv_1.bias"],
        "dw_w": sd["cv.dw_conv_1d.weight"].reshape(256, 33).t(), "dw_b": sd["cv.dw_conv_1d.bias"],
        "bn_g": sd["cv.BN.weight"], "bn_b": sd["cv.BN.bias"], "bn_mean": sd["cv.BN.running_mean"],
        "bn_var": sd["cv.BN.running_var"], "pw2_w": sd["cv.pw_conv_2.weight"].reshape(1), "pw2_b": sd["cv.pw_conv_2.bias"],
    }
    return sd, {k: v.contiguous().to(DEV) for k, v in dev.items()}


@pytest.mark.parametrize("B,T", [(2, 64), (3, 100), (1, 7)])
@pytest.mark.parametrize("batch_stats", [False, True])
def test_convmod(ops, B, T, batch_stats):
    sd, dw = _conv_weights(24)
    x = rnd((B, T, 256), 43)
    y = ops.convmod(x.to(DEV), dw, 33, batch_stats=batch_stats)
    ref = x + oracle.conformer.conv_module(x, sd, "cv", 33, training_bn=batch_stats)
    assert err(y, ref) < (2e-4 if batch_stats else 5e-5)  # all-fp32 kernel


@pytest.mark.parametrize("B,T,drop", [(2, 64, False), (3, 130, True), (2, 400, True), (1, 7, False)])
def test_outproj_conv_train_equals_the_two_operators(ops, B, T, drop):
    """csship_outproj_conv_train (output projection + dropout + residual, LayerNorm, GLU and depthwise convolution in one
    kernel, then the BatchNorm reduce / apply launches) == csship_panel_linear_drop followed by csship_convmod_drop with
    batch statistics: same x2, same module output, and the same workspace contents the backward reads (pre-BatchNorm
    activations and the statistics block); against the oracle (conformer.py:254,356 + 289-303,357, training BatchNorm)
    without dropout."""
    from css_amd.dropout import DropState
    sd, dw = _conv_weights(31)
    D = 256
    M = B * T
    x1 = rnd((M, D), 52).to(DEV)
    ao = rnd((M, D), 53).to(DEV).to(torch.bfloat16)
    wo, bo = rnd((D, D), 54, 1 / 16).to(DEV), rnd((D,), 55, 0.1).to(DEV)
    wp = ops.pack_frag(wo)
    st = DropState(torch.tensor([11], dtype=torch.int32, device=DEV), 0.1, 0)
    dp, dc = (st.cfg("2.attn.out"), st.cfg("2.conv.out")) if drop else (None, None)
    x2, y, ws = ops.outproj_conv_train(ao, wp, bo, x1, dw, B, T, 33, drop_proj=dp, drop_conv=dc)
    x2_ref = ops.panel_linear(ao, wp, bo, D, residual=x1, drop=dp) if drop else ops.panel_linear(ao, wp, bo, D, residual=x1)
    y_ref, ws_ref = ops.convmod(x2_ref.view(B, T, D), dw, 33, batch_stats=True, return_workspace=True, drop=dc)
    assert err(x2, x2_ref) < 1e-6
    assert err(y, y_ref.view(M, D)) < 2e-5
    n = M * D
    z, z_ref = ws.view(torch.float32)[:n], ws_ref.view(torch.float32)[:n]
    assert err(z, z_ref) < 2e-5
    assert err(ws.view(torch.float32)[-4 * D:], ws_ref.view(torch.float32)[-4 * D:]) < 2e-5   # scale | shift | mean | var
    if not drop:
        x2_o = x1.cpu() + bf(ao.float().cpu()) @ bf(wo.cpu()).t() + bo.cpu()
        ref = x2_o.view(B, T, D) + oracle.conformer.conv_module(x2_o.view(B, T, D), sd, "cv", 33, training_bn=True)
        assert err(y.view(B, T, D), ref) < 3e-4


@pytest.mark.parametrize("B,T,F,drop", [(2, 100, 257, False), (3, 64, 257, True), (2, 70, 1799, True)])
def test_embed_train_equals_the_four_launches(ops, B, T, F, drop):
    """csship_embed_train (MVN + Linear + LayerNorm + ReLU + dropout in one kernel, keeping the backward's operands) against
    csship_mvn_apply + csship_bgemm + csship_layernorm_relu + csship_dropout_apply: the bf16 normalised features bit for
    bit, the pre-LayerNorm rows and the output within the two GEMMs' summation orders, the same dropout mask."""
    from css_amd import ops_bwd as ob
    from css_amd.dropout import DropState
    D = 256
    f = rnd((B, T, F), 90).abs().to(DEV)
    w = rnd((D, F), 91, F ** -0.5).to(DEV)
    b, g, be = (0.1 * rnd((D,), 92)).to(DEV), (1 + 0.1 * rnd((D,), 93)).to(DEV), (0.1 * rnd((D,), 94)).to(DEV)
    mean, rinv = ops.mvn_stats(f)
    kp = (F + 255) // 256 * 256
    wp = torch.zeros(D, kp, device=DEV)
    wp[:, :F] = w
    st = DropState(torch.tensor([7], dtype=torch.int32, device=DEV), 0.1, 0)   # (owns the seed word and the cfg structs)
    cfg = st.cfg("embed") if drop else None
    x, fn, e = ops.embed_train(f, mean, rinv, ops.pack_frag(wp), b, g, be, drop=cfg)
    Kp8 = (F + 7) // 8 * 8
    fn0 = ob.mvn_apply(f, mean, rinv, Kp8)
    w_bf, _ = ob.pack_weight(w, want_row_major=True)
    e0 = ob.bgemm(fn0, w_bf, bias=b)
    _, x0 = ops.layernorm_relu(e0, g, be)
    if drop:
        ops.dropout_apply(x0, cfg)
    assert torch.equal(fn, fn0)
    assert err(e, e0) < 2e-5 and err(x, x0) < 1e-4
    assert torch.equal(x == 0, x0 == 0) or float(((x == 0) != (x0 == 0)).float().mean()) < 1e-4   # same mask (ReLU zeros may flip at 0)


@pytest.mark.parametrize("B,T,drop", [(2, 64, False), (3, 128, True), (32, 400, True)])
def test_chain_train_forms_its_own_input_rows_behind_the_conv_module(ops, B, T, drop):
    """Round 6: csship_encoder_chain_train with the bn_* fields behind csship_outproj_conv_train(..., y = NULL) - the conv
    module's tail (BatchNorm apply, ReLU, affine, dropout, residual; conformer.py:298-303) in the chain kernel's prologue
    instead of a launch of its own - against the two-launch form: the rows x3 and every output of the chain BIT FOR BIT (the
    same arithmetic, element for element)."""
    from css_amd.dropout import DropState
    sd, dw = _conv_weights(31)
    D, FF = 256, 1024
    M = B * T
    x1 = rnd((M, D), 52).to(DEV)
    ao = rnd((M, D), 53).to(DEV).to(torch.bfloat16)
    wo, bo = rnd((D, D), 54, 1 / 16).to(DEV), rnd((D,), 55, 0.1).to(DEV)
    wp = ops.pack_frag(wo)
    st = DropState(torch.tensor([11], dtype=torch.int32, device=DEV), 0.1 if drop else 0.0, 0)
    g1, b0 = (1 + 0.1 * rnd((D,), 56)).to(DEV), (0.1 * rnd((D,), 57)).to(DEV)

    def ffn_w(seed):
        return ((g1, b0), ops.pack_frag(rnd((FF, D), seed, 1 / 16).to(DEV)), (0.1 * rnd((FF,), seed + 1)).to(DEV),
                ops.pack_frag(rnd((D, FF), seed + 2, 1 / 32).to(DEV)), (0.1 * rnd((D,), seed + 3)).to(DEV))
    fa, fb = ffn_w(60), ffn_w(70)
    wq, bq = ops.pack_frag(rnd((768, D), 80, 1 / 16).to(DEV)), (0.1 * rnd((768,), 81)).to(DEV)
    drops = tuple(st.cfg(f"2.{t}") for t in ("ffn_out.hidden", "ffn_out.out", "ffn_in.hidden", "ffn_in.out"))
    dp, dc = st.cfg("2.attn.out"), st.cfg("2.conv.out")
    x2a, x3a, wsa = ops.outproj_conv_train(ao, wp, bo, x1, dw, B, T, 33, drop_proj=dp, drop_conv=dc)
    ref = ops.encoder_chain_train(x3a, FF, fb, ((g1, b0), wq, bq, 768), ffn_a=fa, final_ln=(g1, b0), drops=drops)
    x2b, none, wsb = ops.outproj_conv_train(ao, wp, bo, x1, dw, B, T, 33, drop_proj=dp, drop_conv=dc, apply=False)
    assert none is None and torch.equal(x2a, x2b)
    got = ops.encoder_chain_train(None, FF, fb, ((g1, b0), wq, bq, 768), ffn_a=fa, final_ln=(g1, b0), drops=drops,
                                  bn=(x2b, wsb, dw, dc))
    assert torch.equal(got[4], x3a)
    for a, b in zip(got[:4], ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("batch_stats", [False, True])
def test_convmod_causal_golden(ops, golden, batch_stats):
    """csship_conv_weights.causal: the causal ConvModule (conformer.py:265-273, 297-298) against the reference's own
    ConvModule(causal=True) output (tests/golden/causal.npz) and the oracle at a ragged length."""
    g = golden("causal.npz")
    sd, dw = _conv_weights(24)
    dw = dict(dw, causal=True)
    x = rnd((2, 64, 256), 11)
    want = torch.from_numpy(g["conv_trainbn" if batch_stats else "conv_eval"])
    assert err(ops.convmod(x.to(DEV), dw, 33, batch_stats=batch_stats) - x.to(DEV), want) < (2e-4 if batch_stats else 5e-5)
    x = rnd((3, 101, 256), 44)
    ref = x + oracle.conformer.conv_module(x, sd, "cv", 33, training_bn=batch_stats, causal=True)
    assert err(ops.convmod(x.to(DEV), dw, 33, batch_stats=batch_stats), ref) < (2e-4 if batch_stats else 5e-5)


def test_convmod_golden(ops, golden):
    g = golden("conformer.npz")
    sd, dw = _conv_weights(24)
    x = rnd((2, 64, 256), 11)
    assert err(ops.convmod(x.to(DEV), dw, 33) - x.to(DEV), torch.from_numpy(g["conv_eval"])) < 5e-5
    assert err(ops.convmod(x.to(DEV), dw, 33, batch_stats=True) - x.to(DEV), torch.from_numpy(g["conv_trainbn"])) < 2e-4


def test_feature_extractor_frame_major_equals_reference_layout(ops):
    """FeatureExtractor.forward_tf (frame-major STFT -> MVN -> 16-byte-row IPD kernels, no transposes) returns the
    same features as forward (the reference's [B, F, T] layout, itself checked against the goldens): magnitudes and
    spectra bitwise, normalised magnitude to 1 ulp-level, IPD to 2e-6 rad (different summation order of the time
    means); single-channel input without IPD as well; odd frame counts."""
    from css_amd.executor.feature import FeatureExtractor
    for n, B in ((512 + 256 * 36, 3), (512 + 256 * 149, 2)):
        x = (rnd((B, 7, n), 170) * 0.1).to(DEV)
        fe = FeatureExtractor(ipd_index="1,0;2,0;3,0;4,0;5,0;6,0")
        mag, f, re, im = fe.forward(x)
        mag_tf, f_tf, re_tf, im_tf = fe.forward_tf(x)
        assert torch.equal(mag_tf.transpose(1, 2), mag) and torch.equal(re_tf.transpose(2, 3), re) and torch.equal(im_tf.transpose(2, 3), im)
        ft = f_tf.transpose(1, 2)
        assert err(ft[:, :257], f[:, :257]) < 1e-5 * float(f[:, :257].abs().max())
        d = (ft[:, 257:] - f[:, 257:]).abs()
        d = torch.minimum(d, (2 * np.pi - d).abs())  # +-pi wrap
        assert float(d.max()) < 2e-5, float(d.max())
        fe1 = FeatureExtractor()
        mag1, f1, _, _ = fe1.forward(x[:, 0])
        mag1_tf, f1_tf, re1, _ = fe1.forward_tf(x[:, 0], want_spectra=False)
        assert re1 is None and torch.equal(mag1_tf.transpose(1, 2), mag1)
        assert err(f1_tf.transpose(1, 2), f1) < 1e-5 * float(f1.abs().max())


# ----------------------------------------------------------------------------- PIT
@pytest.mark.parametrize("K", [2, 3])
def test_pit_mse(ops, K):
    B, T, F = 5, 40, 257
    masks = rnd((B, T, (K + 1) * F), 50).abs()
    f = rnd((B, T, F), 51).abs()
    tg = [rnd((B, T, F), 52 + k).abs() * 0.5 for k in range(K)]
    loss, perm, y_pred, pairs = ops.pit_mse(masks.to(DEV), f.to(DEV), [t.to(DEV) for t in tg], True, True)
    yp = torch.stack([masks[..., k * F:(k + 1) * F] * f for k in range(K)], 1)
    assert torch.equal(y_pred.cpu(), yp)  # same single fp32 multiply
    ref_loss, ref_perm = oracle.pit.objective_multi(yp, torch.stack(tg, 1))
    assert abs(float(loss) - float(ref_loss)) < 1e-6 * max(1.0, float(ref_loss))
    assert [tuple(p) for p in perm.cpu().tolist()] == [tuple(p) for p in ref_perm]
    for b in range(B):
        assert err(pairs[b], oracle.pit.pairwise_sse(yp[b], torch.stack(tg, 1)[b])) < 1e-3 * float(pairs[b].max())
    # hand-written backward vs autograd of the oracle
    m = masks.clone().requires_grad_(True)
    yp2 = torch.stack([m[..., k * F:(k + 1) * F] * f for k in range(K)], 1)
    oracle.pit.objective_multi(yp2, torch.stack(tg, 1))[0].backward()
    gm = ops.pit_mse_bwd(masks.to(DEV), f.to(DEV), [t.to(DEV) for t in tg], perm, K + 1)
    assert err(gm, m.grad) < 1e-9 + 1e-5 * float(m.grad.abs().max())


@pytest.mark.parametrize("shape", [(1, 1, 5, 1), (3, 37, 257, 2), (7, 130, 64, 3), (2, 50, 771, 4), (5, 16, 3, 2),
                                   (2, 33, 1030, 2)])
def test_pit_mse_ragged_shapes(ops, shape):
    """Row-wise 16-byte-load PIT kernel on shapes that exercise every path: F < 4 (tail lanes only), F % 4 = 0..3, F
    beyond one 256-bin pass, frame counts that are not a multiple of the 16-row chunk, a single sample; y_pred rows
    written by the unaligned 16-byte stores; padded mask rows."""
    B, T, F, K = shape
    ld = (K + 1) * F + 3
    masks = torch.full((B, T, ld), float("nan"))
    masks[..., :(K + 1) * F] = rnd((B, T, (K + 1) * F), 160).abs()
    f = rnd((B, T, F), 161).abs()
    tg = [rnd((B, T, F), 162 + k).abs() * 0.5 for k in range(K)]
    md = masks.to(DEV)[..., :(K + 1) * F]
    loss, perm, y_pred, pairs = ops.pit_mse(md, f.to(DEV), [t.to(DEV) for t in tg], True, True)
    yp = torch.stack([masks[..., k * F:(k + 1) * F] * f for k in range(K)], 1)
    assert torch.equal(y_pred.cpu(), yp)
    ref_loss, ref_perm = oracle.pit.objective_multi(yp, torch.stack(tg, 1))
    assert abs(float(loss) - float(ref_loss)) < 1e-6 * max(1.0, float(ref_loss))
    assert [tuple(p) for p in perm.cpu().tolist()] == [tuple(p) for p in ref_perm]
    for b in range(B):
        assert err(pairs[b], oracle.pit.pairwise_sse(yp[b], torch.stack(tg, 1)[b])) < 1e-3 * float(pairs[b].max()) + 1e-6


def test_pit_loss_api_golden(ops, golden):
    from css_amd.objectives import pit_loss
    g = golden("pit.npz")
    for K in (2, 3):
        est, tgt = rnd((K, 40, 257), 60 + K).abs(), rnd((K, 40, 257), 70 + K).abs()
        l, p = pit_loss(est.to(DEV), tgt.to(DEV), axis=0, return_permutation=True)
        assert abs(float(l) - float(g[f"loss_k{K}"])) < 1e-6 and list(p) == list(g[f"perm_k{K}"])
        rot = list(range(1, K)) + [0]
        est2 = tgt[rot] + 0.01 * rnd((K, 40, 257), 80 + K)
        l2, p2 = pit_loss(est2.to(DEV), tgt.to(DEV), axis=0, return_permutation=True)
        assert abs(float(l2) - float(g[f"loss_rot_k{K}"])) < 1e-7 and list(p2) == list(g[f"perm_rot_k{K}"])
    with pytest.raises(AssertionError):
        pit_loss(est.to(DEV), tgt[:, :10].to(DEV), axis=0)


# ----------------------------------------------------------------------------- wide / MXFP8 FeedForward (ffnw.hip)
def _ffn_params(D, FF, seed):
    g, b = 1 + 0.1 * rnd((D,), seed), 0.1 * rnd((D,), seed + 1)
    w1, b1 = rnd((FF, D), seed + 2, D ** -0.5), rnd((FF,), seed + 3, 0.1)
    w2, b2 = rnd((D, FF), seed + 4, FF ** -0.5), rnd((D,), seed + 5, 0.1)
    return g, b, w1, b1, w2, b2


def _ffn_ref64(x, g, b, w1, b1, w2, b2, fin=None, qx=lambda t: t, qw=lambda t: t):
    """x + 0.5 * FFN(LN(x)) [-> LN] in float64 with operand roundings qx (activations) / qw (weights)."""
    D = x.shape[1]
    a = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double())
    h = torch.relu(qx(a.float()).double() @ qw(w1).double().t() + b1.double())
    y = x.double() + 0.5 * (qx(h.float()).double() @ qw(w2).double().t() + b2.double())
    return torch.nn.functional.layer_norm(y, (D,), fin[0].double(), fin[1].double()) if fin else y


@pytest.mark.parametrize("D,FF,M", [(256, 1024, 300), (512, 2048, 200), (512, 1024, 64)])
def test_ffn_wide_bf16(ops, D, FF, M):
    """csship_ffn_wide (d_model 256 / 512) vs float64 on the same bf16-rounded operands, with and without the final LN."""
    g, b, w1, b1, w2, b2 = _ffn_params(D, FF, 500 + D)
    x = rnd((M, D), 510 + D) * 1.5
    fin = (1 + 0.1 * rnd((D,), 520), 0.1 * rnd((D,), 521))
    d = lambda t: t.to(DEV)  # noqa: E731
    for f in (None, fin):
        out = ops.ffn_wide(d(x), (d(g), d(b)), ops.pack_frag(d(w1)), d(b1), ops.pack_frag(d(w2)), d(b2), FF, 0.5,
                           final_ln=(d(f[0]), d(f[1])) if f else None)
        ref = _ffn_ref64(x, g, b, w1, b1, w2, b2, f, qx=bf, qw=bf)
        assert err(out, ref) < 3e-3, (D, FF, f is not None)


@pytest.mark.parametrize("D,FF,M", [(256, 1024, 300), (512, 2048, 200), (512, 1024, 64)])
def test_ffn_mx(ops, D, FF, M):
    """csship_ffn_mx vs float64 on MXFP8-quantised operands (oracle/mxfp8.py: the OCP MX rule restated) - tight; and
    the format's own price vs the unquantised float64 FeedForward - loose, reported."""
    g, b, w1, b1, w2, b2 = _ffn_params(D, FF, 600 + D)
    x = rnd((M, D), 610 + D) * 1.5
    fin = (1 + 0.1 * rnd((D,), 620), 0.1 * rnd((D,), 621))
    d = lambda t: t.to(DEV)  # noqa: E731
    q = oracle.mxfp8.quantize
    w1p, w2p = ops.pack_mx(d(w1)), ops.pack_mx(d(w2))
    for f in (None, fin):
        out = ops.ffn_mx(d(x), (d(g), d(b)), w1p, d(b1), w2p, d(b2), FF, 0.5, final_ln=(d(f[0]), d(f[1])) if f else None)
        ref_q = _ffn_ref64(x, g, b, w1, b1, w2, b2, f, qx=q, qw=q)
        ref = _ffn_ref64(x, g, b, w1, b1, w2, b2, f)
        e_q, e_fmt = err(out, ref_q), err(out, ref)
        # a LayerNorm output that differs in the last f32 bit can land on the other side of an e4m3 rounding boundary
        # (1 fp8 ulp = 6 % of one element): the bound is on the mean, the max is reported by the assert message
        mse_q = float(((out.cpu().double() - ref_q) ** 2).mean())
        assert mse_q < 2e-6 and e_q < 3e-2, (D, FF, e_q, mse_q)
        assert e_fmt < 0.3, (D, FF, e_fmt)


@pytest.mark.parametrize("D,M", [(256, 300), (512, 200)])
def test_wide_row_panel_linears(ops, D, M):
    """csship_ln_linear_wide (LayerNorm + q|k|v projection, interleaved rows, bf16 out) and csship_linear_res_wide
    (output projection + residual) vs float64 on the same bf16-rounded operands."""
    d = lambda t: t.to(DEV)  # noqa: E731
    x = rnd((M, D), 700 + D) * 1.5
    g, b = 1 + 0.1 * rnd((D,), 701), 0.1 * rnd((D,), 702)
    wq, bq = rnd((3 * D, D), 703, D ** -0.5), rnd((3 * D,), 704, 0.1)
    out = ops.ln_linear_wide(d(x), (d(g), d(b)), d(wq), d(bq))
    a64 = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double())
    ref = bf(a64.float()).double() @ bf(wq).double().t() + bq.double()
    assert out.shape == (M, 3 * D) and err(out.float(), ref) < 3e-2                      # bf16 output of O(2) values
    assert float(((out.float().cpu().double() - ref) ** 2).mean()) < 3e-5
    a = bf(rnd((M, D), 705))
    wo, bo, res = rnd((D, D), 706, D ** -0.5), rnd((D,), 707, 0.1), rnd((M, D), 708)
    out2 = ops.linear_res_wide(d(a).to(torch.bfloat16), d(wo), d(bo), d(res))
    ref2 = res.double() + a.double() @ bf(wo).double().t() + bo.double()
    assert err(out2, ref2) < 2e-4


@pytest.mark.gpu
def test_panel_linear_dropout_epilogue():
    """csship_panel_linear_drop: x + D(a W^T + b) with the mask applied in the projection's epilogue == the projection
    followed by csship_dropout_add (same element indices, same arithmetic order: bit for bit)."""
    from css_amd import ops
    from css_amd.dropout import DropState
    torch.manual_seed(3)
    M, D = 12800, 256
    a = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    w = torch.randn(D, D, device="cuda") * 0.05
    b = torch.randn(D, device="cuda") * 0.1
    x = torch.randn(M, D, device="cuda")
    wp = ops.pack_frag(w)
    cfg = DropState(torch.tensor([77], dtype=torch.int32, device="cuda"), 0.1, 0).cfg("3.attn.out")
    want = ops.dropout_add(x, ops.panel_linear(a, wp, b, D), cfg)
    got = ops.panel_linear(a, wp, b, D, residual=x, drop=cfg)
    assert torch.equal(got, want)
    assert float((got == x).float().mean()) > 0.05          # about one element in ten is dropped
    assert torch.equal(ops.panel_linear(a, wp, b, D, residual=x), x + ops.panel_linear(a, wp, b, D))


@pytest.mark.gpu
def test_ffn_fused_final_layernorm_keeps_the_rows_in_front_of_it():
    """csship_ffn_fused_train: FeedForward + residual + final LayerNorm in one launch, handing back the rows in front of
    the LayerNorm as well == the FeedForward launch followed by the LayerNorm launch."""
    from css_amd import ops
    torch.manual_seed(5)
    M, D, FF = 3000, 256, 1024
    x = torch.randn(M, D, device="cuda")
    ln = (1 + 0.1 * torch.randn(D, device="cuda"), 0.1 * torch.randn(D, device="cuda"))
    fin = (1 + 0.1 * torch.randn(D, device="cuda"), 0.1 * torch.randn(D, device="cuda"))
    w1, w2 = torch.randn(FF, D, device="cuda") * 0.05, torch.randn(D, FF, device="cuda") * 0.03
    b1, b2 = torch.randn(FF, device="cuda") * 0.1, torch.randn(D, device="cuda") * 0.1
    w1p, w2p = ops.pack_frag(w1), ops.pack_frag(w2)
    want_pre = ops.ffn_fused(x, ln, w1p, b1, w2p, b2, FF, 0.5)
    pre, out = ops.ffn_fused(x, ln, w1p, b1, w2p, b2, FF, 0.5, final_ln=fin, want_pre_ln=True)
    assert torch.equal(pre, want_pre)
    assert torch.equal(out, ops.ffn_fused(x, ln, w1p, b1, w2p, b2, FF, 0.5, final_ln=fin))
    ref = torch.nn.functional.layer_norm(pre.double().cpu(), (D,), fin[0].double().cpu(), fin[1].double().cpu(), 1e-5)
    assert float((out.double().cpu() - ref).abs().max()) < 1e-4


@pytest.mark.gpu
def test_attention_training_forward_is_reproducible_at_full_size():
    """Two launches of csship_mha_relpos_train_drop on the same operands at 32 x 4 x 400 (all 256 CUs busy) give the same
    bits - and so do the probabilities / one-pass backward.  A running maximum read from MFMA accumulators still in the
    pipe (inline-assembly v_max3 without wait states) is the same softmax with other roundings, only on a full chip."""
    from css_amd import ops, ops_bwd as ob
    from css_amd.dropout import DropState
    B, T, H = 32, 400, 4
    D = H * 64
    torch.manual_seed(11)
    qkv = (torch.randn(B, T, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
    pack = ops.pack_relpos(torch.randn(2000, 64, device="cuda") * 0.3, 1000)
    dao = (torch.randn(B * T, D, device="cuda") * 0.2).to(torch.bfloat16)
    for cfg in (None, DropState(torch.tensor([5], dtype=torch.int32, device="cuda"), 0.1, 0).cfg("0.attn.p")):
        runs = [ops.mha_relpos_train(qkv, pack, H, drop=cfg) for _ in range(4)]
        assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
        ao, lse = runs[0]
        W = (2 * T - 1 + 7) // 8 * 8
        outs = [ob.mha_bwd_core(qkv, pack, lse, dao, None, torch.zeros(B, T, H, W, dtype=torch.bfloat16, device="cuda"), H,
                                drop=cfg, out=ao.view(B * T, D)) for _ in range(3)]
        assert all(all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:])
    assert all(torch.equal(ops.mha_relpos(qkv, pack, H), ops.mha_relpos(qkv, pack, H)) for _ in range(3))
