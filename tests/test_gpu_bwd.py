"""GPU parity tests of the backward kernels (through the C ABI).  The reference gets its gradients from
torch.autograd, so the checker here is torch on the CPU in float64: closed-form products for the GEMM-class
kernels (on the SAME bf16-rounded operands), autograd of the fp32 oracle formulas for the operator backwards.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def bf(x):
    return x.to(torch.bfloat16).float()


def relerr(got, ref):
    ref = ref.double()
    return float((got.detach().double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def froerr(got, ref):
    ref = ref.double()
    return float((got.detach().double().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ob():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from css_amd import ops_bwd as o
    o.lib()
    return o


@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 100), (257, 771, 1300), (64, 256, 12800),
                                   (400, 64, 400), (264, 48, 1000), (800, 64, 25600), (130, 40, 72)])  # N <= 64: narrow tiles
def test_bgemm_layouts(ob, a_km, b_km, M, N, K):
    """all four operand layouts, ragged sizes, f32 and bf16 inputs, automatic split of long reductions"""
    def pad8(n):
        return (n + 7) // 8 * 8
    A = rnd((M, K), 1, 0.5)
    B = rnd((N, K), 2, 0.5)
    ref = bf(A).double() @ bf(B).double().T
    # storage with padded row strides (vector loads need 16-byte aligned rows)
    a_st = torch.zeros((K, pad8(M)) if a_km else (M, pad8(K)))
    b_st = torch.zeros((K, pad8(N)) if b_km else (N, pad8(K)))
    if a_km:
        a_st[:, :M] = A.T
    else:
        a_st[:, :K] = A
    if b_km:
        b_st[:, :N] = B.T
    else:
        b_st[:, :K] = B
    a_v = a_st.to(DEV)[:, :M] if a_km else a_st.to(DEV)[:, :K]
    b_v = b_st.to(DEV).to(torch.bfloat16)
    b_v = b_v[:, :N] if b_km else b_v[:, :K]
    got = ob.bgemm(a_v, b_v, a_kmajor=a_km, b_kmajor=b_km)  # A f32 (rounded on load), B bf16
    assert got.shape == (M, N)
    assert relerr(got, ref) < 2e-5
    a_h = a_st.to(DEV).to(torch.bfloat16)
    a_h = a_h[:, :M] if a_km else a_h[:, :K]
    b_f = bf(b_st).to(DEV)
    b_f = b_f[:, :N] if b_km else b_f[:, :K]
    got2 = ob.bgemm(a_h, b_f, a_kmajor=a_km, b_kmajor=b_km, alpha=0.5, out_bf16=True)
    assert relerr(got2.float(), 0.5 * ref) < 6e-3  # bf16 output rounding


def test_bgemm_epilogues_and_splits(ob):
    M, N, K = 300, 256, 4096
    A, B = rnd((M, K), 3, 0.3), rnd((N, K), 4, 0.3)
    ref = bf(A).double() @ bf(B).double().T
    a, b = A.to(DEV), B.to(DEV).to(torch.bfloat16)
    bias = rnd((N,), 5).to(DEV)
    h = rnd((M, N), 6).to(DEV).to(torch.bfloat16)
    got = ob.bgemm(a, b, bias=bias, mask_pos=h, alpha=2.0)
    want = (2.0 * ref + bias.cpu().double()) * (h.float().cpu() > 0)
    assert relerr(got, want) < 2e-5
    acc = rnd((M, N), 7).to(DEV)
    keep = acc.clone()
    for splits in (1, 5):  # in-place accumulation, direct and through the slabs
        acc.copy_(keep)
        ob.bgemm(a, b, out=acc, add=acc, splits=splits, alpha=0.25)
        assert relerr(acc, keep.cpu().double() + 0.25 * ref) < 2e-5
    # split result is bit-reproducible
    r1 = ob.bgemm(a, b, splits=7).clone()
    r2 = ob.bgemm(a, b, splits=7)
    assert torch.equal(r1, r2)


def test_bgemm_batched_strided_views(ob):
    """attention-shaped operands: q/k/v as strided slices of the packed [B*T, 3*H*64] projection"""
    B_, T, H = 2, 100, 4
    qkv = rnd((B_, T, 3, H, 64), 8, 0.5)
    g = qkv.to(DEV).to(torch.bfloat16)
    q, k, v = (g[:, :, i].permute(0, 2, 1, 3) for i in range(3))       # [B, H, T, 64] views
    qf, kf, vf = (bf(qkv[:, :, i]).permute(0, 2, 1, 3).double() for i in range(3))
    s = ob.bgemm(q, k)                                                   # [B,H,T,T] = q k^T
    assert relerr(s, qf @ kf.transpose(-1, -2)) < 2e-5
    p = torch.zeros((B_, H, T, 104), device=DEV, dtype=torch.bfloat16)[..., :T]  # rows padded to 8 elements
    p.copy_(torch.softmax(s, -1))
    pv = ob.bgemm(p, v, b_kmajor=True)                                   # P V: V is [T(j), 64] -> reduction-strided
    assert relerr(pv, p.float().cpu().double() @ vf) < 2e-5
    dv = ob.bgemm(p, q, a_kmajor=True, b_kmajor=True)                    # P^T Q
    assert relerr(dv, p.float().cpu().double().transpose(-1, -2) @ qf) < 2e-5
    # output into a strided slice, shared (stride-0) second operand
    pe = rnd((2 * T, 64), 9, 0.5).to(DEV).to(torch.bfloat16)
    r = ob.bgemm(q, pe)
    assert relerr(r, qf @ bf(pe.float().cpu()).double().T) < 2e-5


@pytest.mark.parametrize("K,M,N,dy_f32,x_f32", [(12800, 256, 1024, True, False), (12800, 1024, 256, False, False),
                                                 (12800, 512, 256, True, True), (700, 264, 256, True, False),
                                                 (3000, 768, 264, False, True)])
def test_linear_wgrad_fused_bias_gradient(ob, K, M, N, dy_f32, x_f32):
    """dW = dy^T x and db = column sums of dy from ONE pass over dy (csship_bgemm_wgrad; split and unsplit reductions,
    both operand types) == the product and the column sums as separate calls; db sums the bf16-rounded dy."""
    dy = rnd((K, M), 71).to(DEV) if dy_f32 else rnd((K, M), 71).to(DEV).to(torch.bfloat16)
    x = rnd((K, N), 72).to(DEV) if x_f32 else rnd((K, N), 72).to(DEV).to(torch.bfloat16)
    dw, db = ob.linear_wgrad(dy, x, 0.5)
    dw_ref = ob.bgemm(dy, x, a_kmajor=True, b_kmajor=True, alpha=0.5)
    if dy_f32 or x_f32 or (M % 128) or (N % 128):
        assert torch.equal(dw, dw_ref)                   # the same kernel arithmetic, the same slab order
    else:  # bf16 operands in whole tiles take the LDS-DMA loop with its own slab count: same products, another summation order
        assert relerr(dw, dw_ref.double().cpu()) < 1e-6
    db64 = 0.5 * dy.to(torch.bfloat16).double().sum(0)
    assert float((db.double() - db64).abs().max()) < 2e-5 * float(dy.float().abs().sum(0).max())
    # a head of output rows with its own factor (fused q | k | v projection)
    dwh, dbh = ob.linear_wgrad(dy, x, 0.5, head=(64, 0.25))
    assert torch.equal(dwh[64:], dw[64:]) and torch.equal(dbh[64:], db[64:])
    assert torch.allclose(dwh[:64], dw[:64] * 0.5, rtol=1e-6, atol=0) and torch.allclose(dbh[:64], db[:64] * 0.5, rtol=1e-6, atol=0)
    # strided views: rows of a wider buffer, outputs into slices
    wide = rnd((K, M + 2 * N + 16), 73).to(DEV)
    dwo, dbo = torch.zeros(M + 8, N, device=DEV), torch.zeros(M + 8, device=DEV)
    ob.linear_wgrad(wide[:, 8:8 + M], x, 1.0, dw_out=dwo[8:], db_out=dbo[8:])
    assert torch.equal(dwo[8:], ob.bgemm(wide[:, 8:8 + M], x, a_kmajor=True, b_kmajor=True)) and float(dwo[:8].abs().max()) == 0
    want = wide[:, 8:8 + M].to(torch.bfloat16).double().sum(0)
    assert float((dbo[8:].double() - want).abs().max()) < 2e-5 * float(wide[:, 8:8 + M].abs().sum(0).max())


def test_colsum(ob):
    for M, N, dt in ((12800, 771, torch.float32), (1000, 1024, torch.bfloat16), (7, 3, torch.float32)):
        ld = (N + 7) // 8 * 8
        x = rnd((M, ld), 10).to(DEV).to(dt)[:, :N]
        got = ob.colsum(x)
        assert relerr(got, x.float().cpu().double().sum(0)) < 1e-5


def test_layernorm_bwd_and_relu_bwd(ob):
    for D, relu in ((256, False), (512, False), (256, True), (1024, False)):
        M = 333
        x, dy = rnd((M, D), 11).double(), rnd((M, D), 12).double()
        g, b = (1 + 0.1 * rnd((D,), 13)).double(), (0.1 * rnd((D,), 14)).double()
        add = rnd((M, D), 15)
        xr = x.clone().requires_grad_(True)
        gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
        if relu:
            y = torch.relu(y)
        y.backward(dy)
        dx, dg, db = ob.layernorm_bwd(x.float().to(DEV), dy.float().to(DEV), g.float().to(DEV), b.float().to(DEV),
                                      dx_add=add.to(DEV), relu=relu)
        assert relerr(dx, xr.grad + add.double()) < 2e-5
        assert relerr(dg, gr.grad) < 2e-5 and relerr(db, br.grad) < 2e-5
        # the pass also hands back LayerNorm(x) (before the ReLU) as bf16: same dx / dgamma / dbeta, y within a bf16 rounding
        dx2, dg2, db2, yb = ob.layernorm_bwd(x.float().to(DEV), dy.float().to(DEV), g.float().to(DEV), b.float().to(DEV),
                                             dx_add=add.to(DEV), relu=relu, want_y=True)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)
        yref = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-5)
        assert float((yb.double().cpu() - yref).abs().max()) <= 2.0 ** -8 * float(yref.abs().max()) + 1e-6
    y = rnd((50, 771), 16)
    dyy = rnd((50, 772), 17).to(DEV)[:, :771]
    got = ob.relu_bwd(dyy, y.to(DEV))
    assert torch.equal(got.cpu(), torch.where(y > 0, dyy.cpu(), torch.zeros(())))


def test_qkv_in_bwd_equals_gemm_plus_layernorm_backward(ob):
    """csship_qkv_in_bwd: da = dqkv Wqkv (fragments of Wqkv^T packed as three k-slices of one table), the LayerNorm
    backward, the residual gradient and y = LayerNorm(x) in one kernel vs float64 autograd on the same bf16 operands;
    also the panel form of a 256 -> 256 input gradient (no bias, bf16 out)"""
    from css_amd import ops
    D, M = 256, 640
    x, dy = rnd((M, D), 21), rnd((M, D), 22)
    g, b = 1 + 0.1 * rnd((D,), 23), 0.1 * rnd((D,), 24)
    ws = [rnd((D, D), 25 + i, 0.06) for i in range(3)]
    dqkv = rnd((M, 3 * D), 29, 0.5).to(torch.bfloat16)
    scale = 0.125
    plan = ob.PackPlan(torch.device(DEV))
    wt = torch.empty(3 * D * D, dtype=torch.bfloat16, device=DEV)
    wd = [w.to(DEV) for w in ws]
    for i, w in enumerate(wd):
        plan.add(w, scale=scale if i == 0 else 1.0, transposed=True, frag=wt, k0=i * D, k_total=3 * D)
    plan.run()
    # the same table from one transposed job over the concatenated (scaled) weight
    wcat = torch.cat([wd[0] * scale, wd[1], wd[2]], 0).contiguous()          # [768, 256]
    _, wt_one = ob.pack_weight(wcat.t().contiguous(), want_frag=True)           # logical [256, 768]
    assert torch.equal(wt, wt_one)
    dx, dg, dbeta, y = ob.qkv_in_bwd(dqkv.to(DEV), wt, x.to(DEV), dy.to(DEV), g.to(DEV), b.to(DEV))
    da = dqkv.double() @ bf(wcat.cpu()).double()
    xr = x.double().clone().requires_grad_(True)
    gr, br = g.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    yr.backward(da)
    assert relerr(dx, xr.grad + dy.double()) < 2e-5
    assert relerr(dg, gr.grad) < 2e-5 and relerr(dbeta, br.grad) < 2e-5
    assert float((y.double().cpu() - yr.detach()).abs().max()) <= 2.0 ** -8 * float(yr.detach().abs().max()) + 1e-6
    # panel input gradient: dao = dyb Wo against the fragment-packed Wo^T
    wo = rnd((D, D), 31, 0.06).to(DEV)
    _, wo_tp = ob.pack_weight(wo.t().contiguous(), want_frag=True)
    dyb = rnd((M, D), 32).to(torch.bfloat16)
    dao = ops.panel_linear(dyb.to(DEV), wo_tp, None, D, out_bf16=True)
    ref = dyb.double() @ bf(wo.cpu()).double()
    assert float((dao.double().cpu() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    with pytest.raises(RuntimeError):   # whole panels only
        ob.qkv_in_bwd(dqkv[:100].to(DEV), wt, x[:100].to(DEV), dy[:100].to(DEV), g.to(DEV), b.to(DEV))


def test_ffn_bwd_carries_dx_through_a_second_layernorm(ob):
    """csship_ffn_bwd_ln: x = LayerNorm2(x2) (a block's input is the final LayerNorm of the block below): dx and that
    LayerNorm's dgamma / dbeta from the FeedForward backward kernel's epilogue == csship_ffn_bwd then
    csship_layernorm_bwd (same formulas on the same f32 rows: equal to rounding of the summation order)"""
    from css_amd import ops
    M, D, FF = 640, 256, 1024
    x2 = rnd((M, D), 71).to(DEV)
    g2, b2 = (1 + 0.1 * rnd((D,), 72)).to(DEV), (0.1 * rnd((D,), 73)).to(DEV)
    x = torch.nn.functional.layer_norm(x2, (D,), g2, b2, 1e-5)
    dy = rnd((M, D), 74, 0.1).to(DEV)
    G = {"g": (1 + 0.1 * rnd((D,), 75)).to(DEV), "b": (0.1 * rnd((D,), 76)).to(DEV), "w1": rnd((FF, D), 77, 0.06).to(DEV),
         "b1": rnd((FF,), 78, 0.1).to(DEV), "w2": rnd((D, FF), 79, 0.03).to(DEV)}
    w1p = ops.pack_frag(G["w1"])
    w2tp, w1tp = ops.pack_frag(G["w2"].t().contiguous()), ops.pack_frag(G["w1"].t().contiguous())
    cfg_h, cfg_o = None, None
    ref = ob.ffn_bwd(x, dy, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5, cfg_h, cfg_o)
    dx_ref, dg2_ref, db2_ref = ob.layernorm_bwd(x2, ref[0], g2, b2)
    got = ob.ffn_bwd(x, dy, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5, cfg_h, cfg_o, ln2=(x2, g2))
    assert len(got) == 9
    for a, b in zip(got[1:6], ref[1:6]):       # the FeedForward's own outputs are untouched
        assert torch.equal(a, b)
    assert relerr(got[0], dx_ref.double().cpu()) < 1e-5
    assert relerr(got[7], dg2_ref.double().cpu()) < 1e-5 and relerr(got[8], db2_ref.double().cpu()) < 1e-5
    with pytest.raises(RuntimeError):           # whole panels only
        ob.ffn_bwd(x[:100], dy[:100], G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5, ln2=(x2[:100].contiguous(), g2))


@pytest.mark.parametrize("with_ln2", [False, True])
def test_qkv_ffn_bwd_equals_the_two_launches(ob, with_ln2):
    """csship_qkv_ffn_bwd: the q | k | v projection's input-side backward and the FeedForward backward behind it as one
    kernel (dx of the first = dy of the second, kept in registers) vs csship_qkv_in_bwd then csship_ffn_bwd(_ln) on the
    same operands with the same dropout sites.  The first stage's own outputs bit for bit; behind the hand-over the two
    builds of the same formulas may contract their float expressions differently: dy agrees to an ulp, which the bf16
    rounding of alpha * dy o mask can turn into a bf16 ulp of a handful of dyd / dh elements (measured: 2 and 9 of 180 224 /
    720 896) and from there into 1e-4-level differences of one row of da - bounds: elementwise for the bf16 operands,
    5e-4 of the largest element for the f32 results"""
    from css_amd import ops
    from css_amd.dropout import DropState
    M, D, FF = 704, 256, 1024
    st = DropState(torch.tensor([77], dtype=torch.int32, device=DEV), 0.1, 0)
    cfg_h, cfg_o = st.cfg("3.ffn_in.hidden"), st.cfg("3.ffn_in.out")
    x_attn, dy = rnd((M, D), 301).to(DEV), rnd((M, D), 302, 0.1).to(DEV)
    ag, ab = (1 + 0.1 * rnd((D,), 303)).to(DEV), (0.1 * rnd((D,), 304)).to(DEV)
    wcat = rnd((3 * D, D), 305, 0.06).to(DEV)
    _, wt = ob.pack_weight(wcat.t().contiguous(), want_frag=True)
    dqkv = rnd((M, 3 * D), 306, 0.5).to(torch.bfloat16).to(DEV)
    x2 = rnd((M, D), 307).to(DEV)
    g2, b2 = (1 + 0.1 * rnd((D,), 308)).to(DEV), (0.1 * rnd((D,), 309)).to(DEV)
    x = torch.nn.functional.layer_norm(x2, (D,), g2, b2, 1e-5) if with_ln2 else x2
    G = {"g": (1 + 0.1 * rnd((D,), 310)).to(DEV), "b": (0.1 * rnd((D,), 311)).to(DEV), "w1": rnd((FF, D), 312, 0.06).to(DEV),
         "b1": rnd((FF,), 313, 0.1).to(DEV), "w2": rnd((D, FF), 314, 0.03).to(DEV)}
    w1p = ops.pack_frag(G["w1"])
    w2tp, w1tp = ops.pack_frag(G["w2"].t().contiguous()), ops.pack_frag(G["w1"].t().contiguous())
    ln2 = (x2, g2) if with_ln2 else None
    dx1, dg1, db1, y1 = ob.qkv_in_bwd(dqkv, wt, x_attn, dy, ag, ab)
    ref = ob.ffn_bwd(x, dx1, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5, cfg_h, cfg_o, ln2=ln2)
    (dg, db, y), got = ob.qkv_ffn_bwd(dqkv, wt, x_attn, dy, ag, ab, x, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5,
                                      drop_hidden=cfg_h, drop_out=cfg_o, ln2=ln2)
    assert torch.equal(dg, dg1) and torch.equal(db, db1) and torch.equal(y, y1)
    assert len(got) == len(ref) == (9 if with_ln2 else 7)
    for i, (a, b) in enumerate(zip(got, ref)):
        if a.dtype == torch.bfloat16:   # a, h, dh, dyd: a few elements at most, by one bf16 ulp
            d = (a.float() - b.float()).abs()
            assert int((d > 0).sum()) <= 64 and float((d / b.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -7, i
        else:
            assert relerr(a, b.double().cpu()) < 5e-4, i   # (measured 1.2e-4: one row of dx)
    assert float(got[0].abs().max()) > 0 and float(got[5].float().abs().max()) > 0
    with pytest.raises(RuntimeError):   # whole panels only
        ob.qkv_ffn_bwd(dqkv[:100], wt, x_attn[:100], dy[:100], ag, ab, x[:100].contiguous(), G["g"], G["b"], w1p, G["b1"], w2tp, w1tp,
                       FF, 0.5, drop_hidden=cfg_h, drop_out=cfg_o)
    with pytest.raises(RuntimeError):   # dyd is the only copy of the FeedForward's output gradient
        ob.qkv_ffn_bwd(dqkv, wt, x_attn, dy, ag, ab, x, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5)


def test_ffn_bwd_carries_the_conv_statistics(ob):
    """csship_ffn_bwd_conv + csship_convmod_bwd_after_stats: the statistics pass of the conv module's backward in the epilogue
    of the FeedForward backward in front of it (on the dx rows it has just formed) vs csship_ffn_bwd_drop then
    csship_convmod_bwd_next: the FeedForward's bf16 operands bit for bit (its f32 rows to the last place), the module's results
    to the rounding of another summation order of the channel sums (64-row partials instead of 16-row ones)"""
    from css_amd import ops
    from css_amd.dropout import DropState
    B, T, D, FF = 4, 176, 256, 1024
    M = B * T
    st = DropState(torch.tensor([31], dtype=torch.int32, device=DEV), 0.1, 0)
    c_out, c_next, cfg_h, cfg_o = st.cfg("2.conv.out"), st.cfg("2.attn.out"), st.cfg("2.ffn_out.hidden"), st.cfg("2.ffn_out.out")
    W = {"ln_g": 1 + 0.1 * rnd((D,), 401), "ln_b": 0.1 * rnd((D,), 402), "pw1_w": torch.tensor([0.9, -0.7]),
         "pw1_b": torch.tensor([0.1, 0.2]), "dw_w": rnd((33, D), 403, 0.2), "dw_b": 0.1 * rnd((D,), 404),
         "bn_g": 1 + 0.1 * rnd((D,), 405), "bn_b": 0.1 * rnd((D,), 406), "bn_mean": torch.zeros(D), "bn_var": torch.ones(D),
         "pw2_w": torch.tensor([1.3]), "pw2_b": torch.tensor([-0.2])}
    W = {k: v.to(DEV) for k, v in W.items()}
    xc = rnd((B, T, D), 407).to(DEV)
    y, fws = ops.convmod(xc, W, 33, batch_stats=True, return_workspace=True, drop=c_out)   # the FeedForward's input
    x = y.view(M, D)
    dy = rnd((M, D), 408, 0.1).to(DEV)
    G = {"g": (1 + 0.1 * rnd((D,), 409)).to(DEV), "b": (0.1 * rnd((D,), 410)).to(DEV), "w1": rnd((FF, D), 411, 0.06).to(DEV),
         "b1": rnd((FF,), 412, 0.1).to(DEV), "w2": rnd((D, FF), 413, 0.03).to(DEV)}
    w1p = ops.pack_frag(G["w1"])
    w2tp, w1tp = ops.pack_frag(G["w2"].t().contiguous()), ops.pack_frag(G["w1"].t().contiguous())
    ref = ob.ffn_bwd(x, dy, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5, cfg_h, cfg_o)
    dx_r, g_r, nx_r = ob.convmod_bwd(xc, ref[0].view(B, T, D), W, fws, 33, drop=c_out, want_next=True, next_drop=c_next)
    flat = torch.empty(38 * D + 8, dtype=torch.float32, device=DEV)
    got = ob.ffn_bwd(x, dy, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5, cfg_h, cfg_o, conv=(fws, W["pw2_w"], c_out, flat, B, T))
    for i, (a, b) in enumerate(zip(got, ref)):   # (two builds of the same formulas: the f32 rows may differ in the last place)
        assert torch.equal(a, b) if a.dtype == torch.bfloat16 else relerr(a, b.double().cpu()) < 1e-6, i
    dx_g, g_g, nx_g = ob.convmod_bwd(xc, got[0].view(B, T, D), W, fws, 33, drop=c_out, want_next=True, next_drop=c_next,
                                     flat=flat, stats_done=True)
    assert relerr(dx_g, dx_r.double().cpu()) < 1e-5
    for k in g_r:   # (d dw_b is zero in exact arithmetic behind a batch-statistics BatchNorm: an absolute bound)
        assert float((g_g[k] - g_r[k]).abs().max()) <= 2e-5 * max(float(g_r[k].abs().max()), 1.0), k
    assert float((nx_g.float() - nx_r.float()).abs().max()) <= 2.0 ** -7 * float(nx_r.float().abs().max())
    with pytest.raises(RuntimeError):   # the sums live in the caller's gradient buffer
        ob.convmod_bwd(xc, got[0].view(B, T, D), W, fws, 33, drop=c_out, stats_done=True)


def test_qkv_ffn2_bwd_equals_the_chain_plus_the_feedforward_below(ob):
    """csship_qkv_ffn2_bwd: three stages of one kernel - the q | k | v projection's input gradient, the FeedForward backward
    behind it (through the second LayerNorm) and the FeedForward backward of the block below with its conv module's statistics
    - vs csship_qkv_ffn_bwd then csship_ffn_bwd_conv on the same operands: first-stage outputs bit for bit, the rest within the
    bounds of the two-stage test (another build of the same formulas)"""
    from css_amd import ops
    from css_amd.dropout import DropState
    B, T, D, FF = 4, 176, 256, 1024
    M = B * T
    st = DropState(torch.tensor([55], dtype=torch.int32, device=DEV), 0.1, 0)
    cfg = {k: st.cfg(k) for k in ("3.ffn_in.hidden", "3.ffn_in.out", "2.ffn_out.hidden", "2.ffn_out.out", "2.conv.out")}
    W = {"ln_g": 1 + 0.1 * rnd((D,), 501), "ln_b": 0.1 * rnd((D,), 502), "pw1_w": torch.tensor([0.9, -0.7]),
         "pw1_b": torch.tensor([0.1, 0.2]), "dw_w": rnd((33, D), 503, 0.2), "dw_b": 0.1 * rnd((D,), 504),
         "bn_g": 1 + 0.1 * rnd((D,), 505), "bn_b": 0.1 * rnd((D,), 506), "bn_mean": torch.zeros(D), "bn_var": torch.ones(D),
         "pw2_w": torch.tensor([1.3]), "pw2_b": torch.tensor([-0.2])}
    W = {k: v.to(DEV) for k, v in W.items()}
    xc = rnd((B, T, D), 507).to(DEV)
    x3, fws = ops.convmod(xc, W, 33, batch_stats=True, return_workspace=True, drop=cfg["2.conv.out"])   # the lower FeedForward's input
    x3 = x3.view(M, D)

    def ffn_params(seed):
        G = {"ln_g": (1 + 0.1 * rnd((D,), seed)).to(DEV), "ln_b": (0.1 * rnd((D,), seed + 1)).to(DEV),
             "w1": rnd((FF, D), seed + 2, 0.06).to(DEV), "b1": rnd((FF,), seed + 3, 0.1).to(DEV), "w2": rnd((D, FF), seed + 4, 0.03).to(DEV)}
        G["w1p"] = ops.pack_frag(G["w1"])
        G["w2tp"], G["w1tp"] = ops.pack_frag(G["w2"].t().contiguous()), ops.pack_frag(G["w1"].t().contiguous())
        return G
    Gi, Go = ffn_params(520), ffn_params(530)
    x4 = rnd((M, D), 540).to(DEV)                       # the block-final LayerNorm's input (below) ...
    g2, b2 = (1 + 0.1 * rnd((D,), 541)).to(DEV), (0.1 * rnd((D,), 542)).to(DEV)
    x = torch.nn.functional.layer_norm(x4, (D,), g2, b2, 1e-5)   # ... and its output: the upper FeedForward's input
    x_attn, dy = rnd((M, D), 543).to(DEV), rnd((M, D), 544, 0.1).to(DEV)
    ag, ab = (1 + 0.1 * rnd((D,), 545)).to(DEV), (0.1 * rnd((D,), 546)).to(DEV)
    _, wt = ob.pack_weight(rnd((3 * D, D), 547, 0.06).to(DEV).t().contiguous(), want_frag=True)
    dqkv = rnd((M, 3 * D), 548, 0.5).to(torch.bfloat16).to(DEV)
    # reference: the two-stage kernel, then the lower FeedForward with the conv statistics
    (dg0, db0, y0), r_in = ob.qkv_ffn_bwd(dqkv, wt, x_attn, dy, ag, ab, x, Gi["ln_g"], Gi["ln_b"], Gi["w1p"], Gi["b1"], Gi["w2tp"], Gi["w1tp"],
                                          FF, 0.5, drop_hidden=cfg["3.ffn_in.hidden"], drop_out=cfg["3.ffn_in.out"], ln2=(x4, g2))
    flat0 = torch.zeros(38 * D + 8, dtype=torch.float32, device=DEV)
    r_out = ob.ffn_bwd(x3, r_in[0], Go["ln_g"], Go["ln_b"], Go["w1p"], Go["b1"], Go["w2tp"], Go["w1tp"], FF, 0.5, cfg["2.ffn_out.hidden"],
                       cfg["2.ffn_out.out"], conv=(fws, W["pw2_w"], cfg["2.conv.out"], flat0, B, T))
    fin = dict(x=x, ln_g=Gi["ln_g"], ln_b=Gi["ln_b"], w1p=Gi["w1p"], b1=Gi["b1"], w2tp=Gi["w2tp"], w1tp=Gi["w1tp"],
               drop_hidden=cfg["3.ffn_in.hidden"], drop_out=cfg["3.ffn_in.out"], ln2=(x4, g2))
    fout = dict(x=x3, ln_g=Go["ln_g"], ln_b=Go["ln_b"], w1p=Go["w1p"], b1=Go["b1"], w2tp=Go["w2tp"], w1tp=Go["w1tp"],
                drop_hidden=cfg["2.ffn_out.hidden"], drop_out=cfg["2.ffn_out.out"])
    flat = torch.zeros(38 * D + 8, dtype=torch.float32, device=DEV)
    (dg, db, y), g_in, g_out = ob.qkv_ffn2_bwd(dqkv, wt, x_attn, dy, ag, ab, fin, fout, (fws, W["pw2_w"], cfg["2.conv.out"], flat, B, T), FF)
    assert torch.equal(dg, dg0) and torch.equal(db, db0) and torch.equal(y, y0)
    assert g_in[0] is None and len(g_in) == 9 and len(g_out) == 7

    def close(a, b, what):
        if a.dtype == torch.bfloat16:   # (a bf16 flip in one stage moves a row of the next by 1e-4: a few elements, a few ulps)
            d = (a.float() - b.float()).abs()
            assert int((d > 0).sum()) <= 512 and float(d.max()) <= 2.0 ** -7 * float(b.float().abs().max()), what
        else:
            assert relerr(a, b.double().cpu()) < 5e-4, what
    for i in range(1, 9):
        close(g_in[i], r_in[i], ("ffn_in", i))
    for i in range(7):
        close(g_out[i], r_out[i], ("ffn_out", i))
    assert float((flat[:2 * D + 2] - flat0[:2 * D + 2]).abs().max()) <= 5e-4 * float(flat0[:2 * D + 2].abs().max())
    with pytest.raises(RuntimeError):   # the chain runs through the LayerNorm between the blocks: whole panels only
        ob.qkv_ffn2_bwd(dqkv[:100], wt, x_attn[:100], dy[:100], ag, ab, fin, fout, (fws, W["pw2_w"], cfg["2.conv.out"], flat, B, T), FF)


def test_grouped_weight_gradients_equal_the_single_products(ob):
    """csship_wgrad_hold / csship_wgrad_flush: the weight-gradient products of a block queued and run as ONE launch (the
    reduction cut 2-3 times instead of 16-64) vs float64 on the same bf16 operands and vs the products one by one; a
    product that does not fit the grouped loop (f32 operand, ragged width) runs at once inside the scope"""
    shapes = [(1024, 256), (256, 1024), (768, 256), (256, 256), (128, 384)]
    for K in (6400, 6432, 2008):   # whole 64-row tiles of the reduction; a last tile of 32 rows; of 24 rows
        _grouped_case(ob, K, shapes)


@pytest.mark.parametrize("B,T,H", [(3, 100, 4), (2, 400, 4), (1, 64, 2), (2, 449, 1), (5, 37, 3)])
def test_attn_pos_bwd_equals_the_two_products(ob, B, T, H):
    """csship_attn_pos_bwd: dq = dq1 + dS_skew pe_window and d pe_window += dS_skew^T q in one pass over the skewed dS,
    vs float64 on the same bf16 operands (the band structure as the one-pass backward kernel leaves it)"""
    W = (2 * T - 1 + 7) // 8 * 8
    g = torch.Generator().manual_seed(1000 * B + T)
    ds = (0.3 * torch.randn(B, H, T, T, generator=g)).to(torch.bfloat16)
    dsk = torch.zeros(B, T, H, W, dtype=torch.bfloat16)
    ti = torch.arange(T)
    w_idx = ti[:, None] - ti[None, :] + T - 1                    # [t, j] -> w
    for t in range(T):
        dsk[:, t, :, w_idx[t]] = ds[:, :, t, :]
    win = torch.zeros(W, 64, dtype=torch.bfloat16)
    win[:2 * T - 1] = (0.5 * torch.randn(2 * T - 1, 64, generator=g)).to(torch.bfloat16)
    q = (0.5 * torch.randn(B * T * H, 64, generator=g)).to(torch.bfloat16)
    dq1 = 0.2 * torch.randn(B, H, T, 64, generator=g)
    acc0 = 0.1 * torch.randn(W, 64, generator=g)
    dqkv = torch.zeros(B, T, 3 * H * 64, dtype=torch.bfloat16, device=DEV)
    acc = acc0.clone().to(DEV)
    ob.attn_pos_bwd(dsk.to(DEV), ob.pe_window_t(win.to(DEV), T), q.to(DEV), dq1.to(DEV), dqkv, acc, H)
    ref_dq = dq1.double() + torch.einsum("bthw,wd->bhtd", dsk.double(), win.double())
    ref_dw = acc0.double() + torch.einsum("bthw,bthd->wd", dsk.double(), q.double().view(B, T, H, 64))
    got_dq = dqkv.view(B, T, 3, H, 64)[:, :, 0].permute(0, 2, 1, 3).double().cpu()
    assert float((got_dq - ref_dq).abs().max()) <= 2.0 ** -8 * float(ref_dq.abs().max()) + 1e-6
    assert froerr(acc, ref_dw) < 2e-6
    assert torch.count_nonzero(dqkv.view(B, T, 3, H, 64)[:, :, 1:]) == 0     # the k and v thirds are not touched
    # the same from the row-major dS (the kernel skews it while staging; 16 bytes of slack around the buffer, here NaN)
    ldp = (T + 7) // 8 * 8
    flat = torch.full((B * H * T * ldp + 16,), float("nan"), dtype=torch.bfloat16, device=DEV)
    dsr = flat[8:8 + B * H * T * ldp].view(B, H, T, ldp)
    dsr.zero_()
    dsr[..., :T] = ds.to(DEV)
    dqkv2 = torch.zeros_like(dqkv)
    acc2 = acc0.clone().to(DEV)
    ob.attn_pos_bwd_ds(dsr, ob.pe_window_t(win.to(DEV), T), q.to(DEV), dq1.to(DEV), dqkv2, acc2, H)
    assert torch.equal(dqkv2, dqkv) and torch.equal(acc2, acc)
    if T == 449:
        with pytest.raises(RuntimeError):   # 450 frames: the tile would need a 17th k-step
            ob.attn_pos_bwd(torch.zeros(1, 450, 1, 904, dtype=torch.bfloat16, device=DEV), torch.zeros(64, 1000, dtype=torch.bfloat16, device=DEV),
                            torch.zeros(450, 64, dtype=torch.bfloat16, device=DEV), torch.zeros(1, 1, 450, 64, device=DEV),
                            torch.zeros(1, 450, 192, dtype=torch.bfloat16, device=DEV), torch.zeros(904, 64, device=DEV), 1)


def test_grad_accumulate_many_small_tensors(ob):
    """csship_grad_accumulate: dst += src for a table of tensors in one launch per 96 (dense, odd sizes, unaligned views,
    a transposed 2-d view, a view with a padded row stride), the rest handed back"""
    g = torch.Generator().manual_seed(5)
    sizes = [1, 2, 3, 7, 256, 257, 1024, 65536, 33 * 256, 256 * 257] + [256] * 120
    flat = torch.randn(sum((n + 3) // 4 * 4 for n in sizes) + 8, generator=g).to(DEV)
    ref = flat.clone()
    pairs, off = [], 0
    for i, n in enumerate(sizes):
        dst = flat[off:off + n]
        src = torch.randn(n + 1, generator=g).to(DEV)[1:] if i % 5 == 0 else torch.randn(n, generator=g).to(DEV)   # (some 4-byte aligned only)
        if n == 33 * 256:   # the depthwise weight: kept [33, 256], the parameter is [256, 1, 33]
            m = torch.randn(33, 256, generator=g).to(DEV)
            src = m.t()
            ref[off:off + n] += m.t().contiguous().view(-1)
            dst = dst.view(256, 1, 33)
        elif n == 256 * 257:   # the embed weight gradient: rows 264 floats apart
            m = torch.randn(256, 264, generator=g).to(DEV)
            src = m[:, :257]
            ref[off:off + n] += src.contiguous().view(-1)
            dst = dst.view(256, 257)
        else:
            ref[off:off + n] += src
        pairs.append((dst, src))
        off += (n + 3) // 4 * 4
    odd = (torch.zeros(4, 6, device=DEV), torch.randn(6, 4, 2, generator=g).to(DEV)[:, :, 0].t())   # stride(0) != 1: not taken
    rest = ob.grad_accumulate(pairs + [odd])
    assert len(rest) == 1 and rest[0][0] is odd[0]
    assert torch.equal(flat, ref)


def test_grouped_weight_gradients_large_units(ob):
    """products of 64 tiles each (whole-XCD units of the workgroup table) and one too large to be queued (128 tiles)"""
    K = 2048
    shapes = [(1024, 1024), (1024, 1024), (1024, 1024), (256, 256)]
    ops_ = [((rnd((K, M), 80 + i, 0.5)).to(torch.bfloat16).to(DEV), (rnd((K, N), 90 + i, 0.5)).to(torch.bfloat16).to(DEV))
            for i, (M, N) in enumerate(shapes)]
    big = (rnd((K, 2048), 97, 0.5).to(torch.bfloat16).to(DEV), rnd((K, 1024), 98, 0.5).to(torch.bfloat16).to(DEV))
    with ob.deferred_reductions(group_wgrads=True):
        got = [ob.linear_wgrad(dy, x) for dy, x in ops_]
        assert ob.lib().csship_wgrad_pending() == len(shapes)
        gbig = ob.linear_wgrad(*big)                      # 128 tiles: runs at once
        assert ob.lib().csship_wgrad_pending() == len(shapes)
    for (dy, x), (dw, db) in zip(ops_ + [big], got + [gbig]):
        assert froerr(dw, dy.double().cpu().T @ x.double().cpu()) < 2e-6
        assert froerr(db, dy.double().cpu().sum(0)) < 2e-6


def _grouped_case(ob, K, shapes):
    ops_ = []
    for i, (M, N) in enumerate(shapes):
        dy = rnd((K, M), 40 + i, 0.5).to(torch.bfloat16).to(DEV)
        x = rnd((K, N), 50 + i, 0.5).to(torch.bfloat16).to(DEV)
        ops_.append((dy, x))
    single = [ob.linear_wgrad(dy, x, 0.5, head=(64, 0.25)) for dy, x in ops_]
    odd_dy, odd_x = rnd((K, 72), 60).to(DEV), rnd((K, 200), 61).to(torch.bfloat16).to(DEV)
    odd_ref = ob.linear_wgrad(odd_dy, odd_x)
    with ob.deferred_reductions(group_wgrads=True) as red:
        grouped = [ob.linear_wgrad(dy, x, 0.5, head=(64, 0.25)) for dy, x in ops_]
        assert ob.lib().csship_wgrad_pending() == len(shapes)
        odd = ob.linear_wgrad(odd_dy, odd_x)          # not queued
        assert ob.lib().csship_wgrad_pending() == len(shapes)
        red.flush()
        assert ob.lib().csship_wgrad_pending() == 0
        again = ob.linear_wgrad(*ops_[0], 0.5, head=(64, 0.25))   # left to the exit's flush
    assert ob.lib().csship_wgrad_pending() == 0
    assert torch.equal(odd[0], odd_ref[0]) and torch.equal(odd[1], odd_ref[1])
    for (dy, x), (dw1, db1), (dw2, db2) in zip(ops_, single, grouped):
        ref = dy.double().cpu().T @ x.double().cpu() * 0.5
        ref[:64] *= 0.5
        rb = dy.double().cpu().sum(0) * 0.5
        rb[:64] *= 0.5
        assert froerr(dw2, ref) < 2e-6 and froerr(db2, rb) < 2e-6
        assert froerr(dw2, dw1.double().cpu()) < 2e-6 and froerr(db2, db1.double().cpu()) < 2e-6
    assert froerr(again[0], single[0][0].double().cpu()) < 2e-6


def test_ffn_and_head_backward(ob):
    """gradients of the FeedForward block and of the mask head vs float64 autograd of the oracle formulas"""
    from css_amd import backward as bw, ops
    M, D, FF = 500, 256, 1024
    x, dy = rnd((M, D), 21), rnd((M, D), 22, 0.1)
    P = {"g": 1 + 0.1 * rnd((D,), 23), "b": 0.1 * rnd((D,), 24), "w1": rnd((FF, D), 25, 0.06), "b1": rnd((FF,), 26, 0.1),
         "w2": rnd((D, FF), 27, 0.03), "b2": rnd((D,), 28, 0.1)}
    R = {k: v.double().requires_grad_(True) for k, v in P.items()}
    xr = x.double().requires_grad_(True)
    a = torch.nn.functional.layer_norm(xr, (D,), R["g"], R["b"], 1e-5)
    out = xr + 0.5 * (torch.relu(a @ R["w1"].T + R["b1"]) @ R["w2"].T + R["b2"])
    out.backward(dy.double())
    G = {k: v.to(DEV) for k, v in P.items()}
    dx, grads = bw.ffn_bwd(x.to(DEV), dy.to(DEV), G["g"], G["b"], G["b1"],
                           G["w1"].to(torch.bfloat16), G["w2"].to(torch.bfloat16), 0.5)
    # (a) vs float64 autograd: bf16 rounding plus ReLU units whose pre-activation sign flips under rounding
    # (a few flipped units move single elements by several %, so this check uses the Frobenius norm)
    assert froerr(dx, xr.grad) < 6e-2
    for got, k in zip(grads, ("g", "b", "w1", "b1", "w2", "b2")):
        assert froerr(got, R[k].grad) < 6e-2, k
    # (b) tight: float64 evaluation of the SAME rounded operands (a, h, dy, dh rounded to bf16 where the kernels do)
    xd = x.double().requires_grad_(True)
    gd, bd = P["g"].double().requires_grad_(True), P["b"].double().requires_grad_(True)
    a_full = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-5)
    a_b = bf(a_full.detach().float()).double()
    w1b, w2b, dyb = bf(P["w1"]).double(), bf(P["w2"]).double(), bf(dy).double()
    h_b = bf(torch.relu(a_b @ w1b.T + P["b1"].double()).float()).double()
    dh = bf((0.5 * (dyb @ w2b) * (h_b > 0)).float()).double()
    want = {"w2": 0.5 * dyb.T @ h_b, "b2": 0.5 * dy.double().sum(0), "w1": dh.T @ a_b, "b1": dh.sum(0)}
    a_full.backward(bf((dh @ w1b).float()).double() * 0 + dh @ w1b)
    want.update({"g": gd.grad, "b": bd.grad})
    tol = 2e-3  # h / a recomputed on the GPU may differ from this emulation by one bf16 ulp in a few elements
    assert relerr(dx, xd.grad + dy.double()) < tol
    for got, k in zip(grads, ("g", "b", "w1", "b1", "w2", "b2")):
        assert relerr(got, want[k]) < tol, k
    tol = 5e-2
    # the fused row-panel kernel (csship_ffn_bwd) gives the same gradients as the composed route
    w1p = ops.pack_frag(G["w1"])
    w2tp, w1tp = ops.pack_frag(G["w2"].t().contiguous()), ops.pack_frag(G["w1"].t().contiguous())
    dx2, grads2 = bw.ffn_bwd_fused(x.to(DEV), dy.to(DEV), G["g"], G["b"], G["b1"], w1p, w2tp, w1tp, FF, 0.5)
    assert relerr(dx2, xd.grad + dy.double()) < 2e-3
    for got, k in zip(grads2, ("g", "b", "w1", "b1", "w2", "b2")):
        assert relerr(got, want[k]) < 2e-3, k
    # mask head
    N = 771
    w, b = rnd((N, D), 31, 0.06), rnd((N,), 32, 0.1)
    xe = rnd((M, D), 33)
    wr, br, xr = w.double().requires_grad_(True), b.double().requires_grad_(True), xe.double().requires_grad_(True)
    masks = torch.relu(xr @ wr.T + br)
    dm = rnd((M, N), 34, 0.1)
    masks.backward(dm.double())
    mg = torch.zeros((M, 772), device=DEV)[:, :N]
    mg.copy_(masks.detach().float())
    dmg = torch.zeros((M, 772), device=DEV)[:, :N]
    dmg.copy_(dm)
    dx, dw, db = bw.head_bwd(dmg, mg, xe.to(DEV), w.to(DEV).to(torch.bfloat16))
    assert froerr(dx, xr.grad) < tol and froerr(dw, wr.grad) < tol and froerr(db, br.grad) < tol
    # the training path's route: weight shadow with zero rows to a multiple of 8, dz in bf16 rows of that width
    wp = torch.zeros((776, D), dtype=torch.bfloat16, device=DEV)
    wp[:N] = w.to(DEV).to(torch.bfloat16)
    dz = ob.relu_bwd_bf16(dmg, mg, 776)
    zr = torch.where(masks.detach() > 0, dm.double(), torch.zeros((), dtype=torch.float64)).float().to(torch.bfloat16)
    assert torch.equal(dz[:, :N].cpu(), zr) and not dz[:, N:].any()
    dx2, dw2, db2 = bw.head_bwd(dmg, mg, xe.to(DEV), wp)
    assert dw2.shape == (N, D) and db2.shape == (N,)
    assert froerr(dx2, xr.grad) < tol and froerr(dw2, wr.grad) < tol and froerr(db2, br.grad) < tol


def _attn_ref(x, P, T, H, maxlen=1000):
    """float64 restatement of conformer.py:213-254 + the residual of :356 (q scaled by 1/8 as in the reference)"""
    B = x.shape[0]
    D = x.shape[-1]
    a = torch.nn.functional.layer_norm(x, (D,), P["g"], P["b"], 1e-5)
    qkv = a @ P["wqkv"].T + P["bqkv"]
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, T, H, 64).transpose(1, 2) for i in range(3))
    pos = (torch.arange(T)[:, None] - torch.arange(T)[None, :]).clamp(-maxlen, maxlen - 1) + maxlen
    pk = P["pe"][pos]                                        # [T,T,64]
    s = (q @ k.transpose(-1, -2) + torch.einsum("bhid,ijd->bhij", q, pk)) / 8.0
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, D)
    return x + o @ P["wo"].T + P["bo"]


def test_attention_backward(ob):
    from css_amd import backward as bw, ops
    B, T, H, D = 2, 90, 4, 256
    x = rnd((B, T, D), 41)
    P = {"g": 1 + 0.1 * rnd((D,), 42), "b": 0.1 * rnd((D,), 43), "wqkv": rnd((3 * D, D), 44, 0.08),
         "bqkv": rnd((3 * D,), 45, 0.1), "wo": rnd((D, D), 46, 0.06), "bo": rnd((D,), 47, 0.1),
         "pe": rnd((2000, 64), 48, 0.3)}
    R = {k: v.double().requires_grad_(True) for k, v in P.items()}
    xr = x.double().requires_grad_(True)
    dy = rnd((B, T, D), 49, 0.1)
    _attn_ref(xr, R, T, H).backward(dy.double())
    # forward on the GPU exactly as the model does it (q rows pre-scaled)
    G = {k: v.to(DEV) for k, v in P.items()}
    ws = G["wqkv"].clone()
    ws[:D] *= 0.125
    bs = G["bqkv"].clone()
    bs[:D] *= 0.125
    x2 = x.to(DEV).view(B * T, D)
    qkv = ops.panel_linear(x2, ops.pack_frag(ws), bs, 3 * D, ln=(G["g"], G["b"]), out_bf16=True).view(B, T, 3 * D)
    ao = ops.mha_relpos(qkv, ops.pack_relpos(G["pe"]), H)
    dx, grads = bw.mha_bwd(x2, dy.to(DEV).view(B * T, D), qkv, ao.view(B * T, D), G["g"], G["b"],
                           ws.to(torch.bfloat16), G["wo"].to(torch.bfloat16), G["pe"].to(torch.bfloat16), H)
    tol = 5e-2  # vs float64 autograd of the unrounded formula
    assert froerr(dx.view(B, T, D), xr.grad) < tol
    for got, k in zip(grads, ("g", "b", "wqkv", "bqkv", "wo", "bo", "pe")):
        assert froerr(got, R[k].grad) < tol, k
    # tight check of the attention core on the SAME bf16 q/k/v/pe/dO: only P and dS are rounded inside
    dao = rnd((B * T, D), 50, 0.1).to(DEV).to(torch.bfloat16)
    dqkv, dpe = bw.mha_core_bwd(qkv, dao, G["pe"].to(torch.bfloat16), H)
    qd = qkv.float().cpu().double().requires_grad_(True)
    ped = bf(P["pe"]).double().requires_grad_(True)
    q, k, v = (qd[..., i * D:(i + 1) * D].reshape(B, T, H, 64).transpose(1, 2) for i in range(3))
    pos = (torch.arange(T)[:, None] - torch.arange(T)[None, :]) + 1000
    sc = q @ k.transpose(-1, -2) + torch.einsum("bhid,ijd->bhij", q, ped[pos])
    o = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * T, D)
    o.backward(dao.float().cpu().double())
    assert relerr(dqkv.float(), qd.grad) < 1e-2
    assert relerr(dpe, ped.grad) < 1e-2


@pytest.mark.parametrize("causal", [False, True])
def test_convmod_backward(ob, causal):
    """ConvModule backward (LN -> scalar GLU -> depthwise conv -> batch-stat BatchNorm -> ReLU -> affine, + residual)
    vs float64 autograd of the same formula (conformer.py:289-303,357; causal: :265-273, 297-298).  All-f32 kernels:
    tight tolerance."""
    import torch.nn.functional as F
    from css_amd import ops
    for B, T, D in ((3, 77, 256), (2, 50, 512)):
        K = 33
        x = rnd((B, T, D), 61)
        dy = rnd((B, T, D), 62, 0.1)
        P = {"ln_g": 1 + 0.1 * rnd((D,), 63), "ln_b": 0.1 * rnd((D,), 64), "pw1_w": torch.tensor([0.9, -0.7]),
             "pw1_b": torch.tensor([0.1, 0.2]), "dw": rnd((D, K), 65, 0.2), "dw_b": 0.1 * rnd((D,), 66),
             "bn_g": 1 + 0.1 * rnd((D,), 67), "bn_b": 0.1 * rnd((D,), 68), "pw2_w": torch.tensor([1.3]),
             "pw2_b": torch.tensor([-0.2])}
        R = {k: v.double().requires_grad_(True) for k, v in P.items()}
        xr = x.double().requires_grad_(True)
        a = F.layer_norm(xr, (D,), R["ln_g"], R["ln_b"], 1e-5)
        g = (a * R["pw1_w"][0] + R["pw1_b"][0]) * torch.sigmoid(a * R["pw1_w"][1] + R["pw1_b"][1])
        if causal:
            c = F.conv1d(g.transpose(1, 2), R["dw"][:, None, :], R["dw_b"], padding=K - 1, groups=D)[:, :, : -(K - 1)]
        else:
            c = F.conv1d(g.transpose(1, 2), R["dw"][:, None, :], R["dw_b"], padding=K // 2, groups=D)
        mu, var = c.mean((0, 2), keepdim=True), c.var((0, 2), unbiased=False, keepdim=True)
        zh = (c - mu) / torch.sqrt(var + 1e-5) * R["bn_g"][None, :, None] + R["bn_b"][None, :, None]
        y = xr + (torch.relu(zh) * R["pw2_w"] + R["pw2_b"]).transpose(1, 2)
        y.backward(dy.double())
        W = {k: v.to(DEV) for k, v in P.items() if k != "dw"}
        W["dw_w"] = P["dw"].T.contiguous().to(DEV)                      # the forward takes the taps as [k, D]
        W["causal"] = causal
        yg, fws = ops.convmod(x.to(DEV), W, K, batch_stats=True, return_workspace=True)
        assert relerr(yg, y.detach()) < 1e-5
        stats = fws.view(torch.float32)[-4 * D:]
        assert relerr(stats[2 * D:3 * D], mu.detach().flatten()) < 1e-5
        assert relerr(stats[3 * D:], var.detach().flatten()) < 1e-5
        dx, grads = ob.convmod_bwd(x.to(DEV), dy.to(DEV), W, fws, K)
        assert relerr(dx, xr.grad) < 2e-4
        ref = {k: R[k].grad for k in P if k != "dw"}
        ref["dw_w"] = R["dw"].grad.T
        for k, v in ref.items():
            if k == "dw_b":  # analytically zero under batch-stat BatchNorm (the mean is subtracted right after)
                assert float(grads[k].abs().max()) < 1e-5
            else:
                assert relerr(grads[k], v) < 2e-4, k


def test_convmod_backward_behind_the_fused_training_forward(ob):
    """the training forward's fused kernel (csship_outproj_conv_train: output projection + residual +
    LayerNorm + GLU + depthwise convolution, then the BatchNorm reduce / apply launches) leaves the workspace the conv
    module's backward reads.  Its gradients - the SCALAR ones included (pw_conv_1 / pw_conv_2: sums with heavy
    cancellation that the model-level bounds cannot pin) - against float64 autograd of conformer.py:254,356 + 289-303,357
    on the same bf16-rounded projection operands and the same upstream gradient."""
    import torch.nn.functional as F
    from css_amd import ops
    B, T, D, K = 3, 130, 256, 33
    M = B * T
    x1 = rnd((M, D), 71)
    ao = rnd((M, D), 72).to(torch.bfloat16)
    wo, bo = rnd((D, D), 73, 1 / 16), rnd((D,), 74, 0.1)
    dy = rnd((B, T, D), 75, 0.1)
    P = {"ln_g": 1 + 0.1 * rnd((D,), 63), "ln_b": 0.1 * rnd((D,), 64), "pw1_w": torch.tensor([0.9, -0.7]),
         "pw1_b": torch.tensor([0.1, 0.2]), "dw": rnd((D, K), 65, 0.2), "dw_b": 0.1 * rnd((D,), 66),
         "bn_g": 1 + 0.1 * rnd((D,), 67), "bn_b": 0.1 * rnd((D,), 68), "pw2_w": torch.tensor([1.3]),
         "pw2_b": torch.tensor([-0.2])}
    R = {k: v.double().requires_grad_(True) for k, v in P.items()}
    x2r = (x1.double() + ao.double() @ wo.to(torch.bfloat16).double().t() + bo.double()).view(B, T, D).requires_grad_(True)
    a = F.layer_norm(x2r, (D,), R["ln_g"], R["ln_b"], 1e-5)
    g = (a * R["pw1_w"][0] + R["pw1_b"][0]) * torch.sigmoid(a * R["pw1_w"][1] + R["pw1_b"][1])
    c = F.conv1d(g.transpose(1, 2), R["dw"][:, None, :], R["dw_b"], padding=K // 2, groups=D)
    mu, var = c.mean((0, 2), keepdim=True), c.var((0, 2), unbiased=False, keepdim=True)
    zh = (c - mu) / torch.sqrt(var + 1e-5) * R["bn_g"][None, :, None] + R["bn_b"][None, :, None]
    y = x2r + (torch.relu(zh) * R["pw2_w"] + R["pw2_b"]).transpose(1, 2)
    y.backward(dy.double())
    W = {k: v.to(DEV) for k, v in P.items() if k != "dw"}
    W["dw_w"] = P["dw"].T.contiguous().to(DEV)
    x2, yg, fws = ops.outproj_conv_train(ao.to(DEV), ops.pack_frag(wo.to(DEV)), bo.to(DEV), x1.to(DEV), W, B, T, K)
    assert relerr(x2.view(B, T, D), x2r.detach()) < 1e-5 and relerr(yg.view(B, T, D), y.detach()) < 1e-4
    dx, grads = ob.convmod_bwd(x2.view(B, T, D), dy.to(DEV), W, fws, K)
    assert relerr(dx, x2r.grad) < 2e-4
    ref = {k: R[k].grad for k in P if k != "dw"}
    ref["dw_w"] = R["dw"].grad.T
    for k, v in ref.items():
        if k == "dw_b":
            assert float(grads[k].abs().max()) < 1e-5
        else:
            assert relerr(grads[k], v) < 3e-4, k


def test_fused_adam_clip_matches_torch():
    """csship_adam_clip (clip_grad_norm_ + Adam on flat buffers) vs torch.optim.Adam + clip_grad_norm_, 5 steps,
    clipping active on some steps; and the non-finite-gradient guard."""
    from css_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(257, 33), (1000,), (3, 5, 7), (1,), (64, 64)]
    ref = [torch.nn.Parameter(rnd(s, 70 + i).to(DEV)) for i, s in enumerate(shapes)]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.98), eps=1e-8)
    o_mine = FusedAdam(mine, lr=3e-3, betas=(0.9, 0.98), eps=1e-8)
    for step in range(5):
        scale = 10.0 if step % 2 == 0 else 0.01   # norm above / below the threshold
        for i, (a, b) in enumerate(zip(ref, mine)):
            g = rnd(a.shape, 100 + 10 * step + i, scale).to(DEV)
            a.grad = g.clone()
            b.grad.copy_(g)
        n_ref = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        o_ref.step()
        n_mine = o_mine.step(max_norm=5.0)
        assert abs(float(n_ref) - float(n_mine)) < 1e-4 * float(n_ref)
        for a, b in zip(ref, mine):
            assert float((a - b).abs().max()) < 2e-6
        o_mine.zero_grad()
        assert all(float(b.grad.abs().max()) == 0.0 for b in mine)
    before = [b.detach().clone() for b in mine]
    mine[0].grad[0, 0] = float("nan")
    o_mine.step(max_norm=5.0)
    assert all(torch.equal(x, b.detach()) for x, b in zip(before, mine))   # skipped


def test_attention_probs_recompute(ob):
    """csship_mha_relpos_train's log-sum-exp and csship_mha_relpos_probs' P vs the materialised route
    (q k^T + window GEMM + row softmax), incl. a T that is not a multiple of the 64-key tile."""
    from css_amd import backward as bw, ops
    for B, T, H in ((2, 90, 4), (1, 200, 8)):
        qkv = rnd((B, T, 3 * H * 64), 81, 0.6).to(DEV).to(torch.bfloat16)
        pe = rnd((2000, 64), 82, 0.3).to(DEV)
        pack = ops.pack_relpos(pe)
        ao0 = ops.mha_relpos(qkv, pack, H)
        ao, lse = ops.mha_relpos_train(qkv, pack, H)
        assert torch.equal(ao0, ao)
        p = ops.mha_relpos_probs(qkv, pack, lse, H)
        g = qkv.view(B, T, 3, H, 64)
        q, k = (g[:, :, i].permute(0, 2, 1, 3) for i in range(2))
        win, _ = bw.relpos_window(pe.to(torch.bfloat16), T, 1000)
        s = ob.bgemm(q, k)
        rel = ob.bgemm(q, win[None, None])
        pref = ob.relpos_softmax(s, rel, T)
        assert float((p[..., :T].float() - pref[..., :T].float()).abs().max()) < 4e-3   # bf16 ulp of values <= 1
        assert float(p[..., T:].float().abs().max()) == 0.0 if p.shape[3] > T else True
        # lse against float64
        sc = s.double().cpu() + torch.stack([rel.double().cpu()[:, :, i, i + T - 1 - torch.arange(T)] for i in range(T)], 2)
        assert float((lse.double().cpu() - torch.logsumexp(sc, -1)).abs().max()) < 2e-3


def test_pack_weights_multi_matches_single_and_host_layout():
    """csship_pack_weights_multi (one launch for a table of weights, optionally reading the source transposed) vs
    csship_pack_weight per weight and vs the host-side fragment layout (ops.pack_frag): bitwise equal, including
    shared destinations written at row offsets and ragged shapes (R % 16 != 0, K % 32 != 0)."""
    from css_amd import ops, ops_bwd as ob
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(s, generator=g).to(DEV) for s in ((256, 1024), (1024, 256), (771, 256), (40, 72), (256, 257))]
    plan = ob.PackPlan(torch.device(DEV, torch.cuda.current_device()))
    outs = []
    for i, w in enumerate(ws):  # row-major and fragment shadows as separate jobs (their K paddings differ when K % 32 != 0)
        outs.append((plan.add(w, scale=0.5 if i == 1 else 1.0, want_row_major=True)[0],
                     plan.add(w, scale=0.5 if i == 1 else 1.0, want_frag=True)[1]))
    tr = plan.add(ws[0], transposed=True, want_frag=True)[1]           # logical [1024, 256] = ws[0]^T
    shared_rm = torch.zeros((768, 256), dtype=torch.bfloat16, device=DEV)
    shared_fr = torch.zeros(768 * 256, dtype=torch.bfloat16, device=DEV)
    qkv = [torch.randn((256, 256), generator=g).to(DEV) for _ in range(3)]
    for i, w in enumerate(qkv):
        plan.add(w, scale=0.125 if i == 0 else 1.0, row_major=shared_rm, frag=shared_fr, row0=256 * i)
    plan.run()
    plan.run()  # the table is reused
    for i, (w, (rm, fr)) in enumerate(zip(ws, outs)):
        rm1, _ = ob.pack_weight(w, scale=0.5 if i == 1 else 1.0, want_row_major=True)
        _, fr1 = ob.pack_weight(w, scale=0.5 if i == 1 else 1.0, want_frag=True)
        assert torch.equal(rm, rm1) and torch.equal(fr, fr1), i
    assert torch.equal(tr, ob.pack_weight(ws[0].t().contiguous(), want_frag=True)[1])
    assert torch.equal(tr, ops.pack_frag(ws[0].t().contiguous()))
    cat = torch.cat([qkv[0] * 0.125, qkv[1], qkv[2]], 0)
    assert torch.equal(shared_rm, cat.to(torch.bfloat16)) and torch.equal(shared_fr, ops.pack_frag(cat))


def test_deferred_reductions_match_immediate(ob):
    """csship_reduce_defer / csship_reduce_flush: the queued slab reductions (LayerNorm backward, fused FeedForward
    backward: dgamma | dbeta partials) give bit for bit what their own launches give, several jobs in one flush; the
    large ones (weight-gradient products) are never queued."""
    K = 12800
    xs = [rnd((K, 256), 5 + i).to(DEV) for i in range(3)]
    dys = [rnd((K, 256), 15 + i).to(DEV) for i in range(3)]
    g, b = rnd((256,), 7).to(DEV) + 1.0, rnd((256,), 8).to(DEV)
    dyb, xb = rnd((K, 256), 80).to(DEV).to(torch.bfloat16), rnd((K, 1024), 90).to(DEV).to(torch.bfloat16)
    want_ln = [ob.layernorm_bwd(x, dy, g, b) for x, dy in zip(xs, dys)]
    want_w = ob.linear_wgrad(dyb, xb, 0.5)
    with ob.deferred_reductions() as red:
        got_ln = [ob.layernorm_bwd(x, dy, g, b) for x, dy in zip(xs, dys)]
        assert _lib_pending() == 3
        got_w = ob.linear_wgrad(dyb, xb, 0.5)
        assert _lib_pending() == 3                       # 16 MB of slabs: reduced at once
        assert torch.equal(got_w[0], want_w[0]) and torch.equal(got_w[1], want_w[1])
        red.flush()
        assert _lib_pending() == 0
        for got, want in zip(got_ln, want_ln):
            for a_, b_ in zip(got, want):
                assert torch.equal(a_, b_)
        again = ob.layernorm_bwd(xs[0], dys[0], g, b)   # left to the exit's flush
    assert _lib_pending() == 0
    for a_, b_ in zip(again, want_ln[0]):
        assert torch.equal(a_, b_)
    out = ob.layernorm_bwd(xs[1], dys[1], g, b)         # outside the scope: immediate again
    assert _lib_pending() == 0 and torch.equal(out[1], want_ln[1][1])


def _lib_pending():
    from css_amd._lib import lib
    return lib().csship_reduce_pending()


@pytest.mark.parametrize("B,T,H,p_drop", [(2, 400, 4, 0.0), (2, 400, 4, 0.1), (3, 77, 2, 0.0), (1, 150, 4, 0.1), (32, 400, 4, 0.1)])
def test_attention_backward_core_one_pass(ob, B, T, H, p_drop):
    """csship_mha_relpos_bwd_core (probabilities + softmax backward + dS k in one pass) against the three launches it
    replaces: the dV operand bit for bit; dS within a bf16 rounding of P (the fused pass multiplies by the unrounded
    probability); the skewed copy is exactly the fused dS moved onto the window axis; dq1 = (fused dS) k."""
    from css_amd import ops
    from css_amd.dropout import DropState
    D = H * 64
    qkv = rnd((B, T, 3 * D), 60, 0.5).to(DEV).to(torch.bfloat16)
    pe = rnd((2000, 64), 61, 0.3).to(DEV)
    pe_pack = ops.pack_relpos(pe, 1000)
    dao = rnd((B * T, D), 62, 0.2).to(DEV).to(torch.bfloat16)
    cfg = None
    if p_drop > 0:
        ds_ = DropState(torch.tensor([1234], dtype=torch.int32, device=DEV), p_drop, 0)
        cfg = ds_.cfg("0.attn.p")
    ao, lse = ops.mha_relpos_train(qkv, pe_pack, H, drop=cfg)
    d = ob.attn_rowdot(dao, ao.view(B * T, D), B, T, H)
    W = (2 * T - 1 + 7) // 8 * 8
    if cfg is not None:
        p, p_v0 = ops.mha_relpos_probs(qkv, pe_pack, lse, H, drop=cfg)
    else:
        p = p_v0 = ops.mha_relpos_probs(qkv, pe_pack, lse, H)
    ds0, sk0 = ob.attn_softmax_bwd(qkv, dao, p, d, torch.zeros((B, T, H, W), dtype=torch.bfloat16, device=DEV), H, drop=cfg)
    p_v, ds, sk, dq1 = ob.mha_bwd_core(qkv, pe_pack, lse, dao, d, torch.zeros((B, T, H, W), dtype=torch.bfloat16, device=DEV), H, drop=cfg)
    assert torch.equal(p_v, p_v0)
    # D computed by the pass itself from the forward's output, and the q rows in the skewed buffer's row order
    p_v2, ds2, _, dq2, q_rows = ob.mha_bwd_core(qkv, pe_pack, lse, dao, None, torch.zeros((B, T, H, W), dtype=torch.bfloat16, device=DEV),
                                                H, drop=cfg, out=ao.view(B * T, D), want_q_rows=True)
    assert torch.equal(p_v2, p_v) and torch.equal(q_rows, qkv.view(B, T, 3, H, 64)[:, :, 0].reshape(B * T * H, 64))
    assert float((ds2.float() - ds.float()).abs().max()) <= 2.0 ** -7 * float(ds.float().abs().max()) and froerr(ds2.float(), ds.float().cpu()) < 2e-3
    assert relerr(dq2, dq1.double().cpu()) < 5e-3
    a, b_ = ds.float(), ds0.float()
    assert float((a - b_).abs().max()) <= 2.0 ** -7 * float(b_.abs().max()) and froerr(a, b_.cpu()) < 4e-3
    # the skew: ds_skew[b, i, h, i - j + T - 1] = dS[b, h, i, j]
    i = torch.arange(T, device=DEV)
    r = (i[:, None] - i[None, :] + T - 1)                                # [i, j]
    want = torch.zeros_like(sk)
    want[:, i[:, None].expand(T, T), :, r] = ds[..., :T].permute(2, 3, 0, 1)     # [i, j, B, H]
    assert torch.equal(sk, want)
    g = qkv.view(B, T, 3, H, 64)
    k = g[:, :, 1].permute(0, 2, 1, 3)
    ref = ds[..., :T].float().cpu().double() @ k.float().cpu().double()
    assert relerr(dq1, ref) < 2e-5


@pytest.mark.parametrize("B,T,H,p_drop", [(2, 400, 4, 0.0), (2, 400, 4, 0.1), (3, 77, 2, 0.0), (1, 150, 4, 0.1), (2, 64, 4, 0.1),
                                          (1, 449, 4, 0.1), (5, 33, 3, 0.0), (32, 400, 4, 0.1)])
def test_attention_backward_fused_key_stationary(ob, B, T, H, p_drop):
    """csship_mha_relpos_bwd_fused (one key-stationary pass: probabilities recomputed, dS out once, dK / dV in registers, dq1
    as one partial per key group) + csship_attn_pos_bwd_ds_parts against the path of rounds 3-5 (csship_mha_relpos_bwd_core +
    csship_attn_dvdk + csship_attn_pos_bwd_ds) on the same operands: dS within one bf16 rounding (both form P o (dP - D)
    from scores whose summation orders differ), dK / dV / dq and the window gradient within the bf16 noise of their
    operands; and dK, dV, dq1 against float64 products of the kernel's OWN dS / the old path's P' (tight)."""
    from css_amd import ops
    from css_amd.dropout import DropState
    D = H * 64
    qkv = rnd((B, T, 3 * D), 60, 0.5).to(DEV).to(torch.bfloat16)
    pe = rnd((2000, 64), 61, 0.3).to(DEV)
    pe_pack = ops.pack_relpos(pe, 1000)
    dao = rnd((B * T, D), 62, 0.2).to(DEV).to(torch.bfloat16)
    cfg = None
    if p_drop > 0:
        ds_ = DropState(torch.tensor([1234], dtype=torch.int32, device=DEV), p_drop, 0)
        cfg = ds_.cfg("0.attn.p")
    ao, lse = ops.mha_relpos_train(qkv, pe_pack, H, drop=cfg)
    ao2 = ao.view(B * T, D)
    # the old path
    p_v, ds0, _, dq10, q_rows = ob.mha_bwd_core(qkv, pe_pack, lse, dao, None, None, H, drop=cfg, out=ao2, want_q_rows=True)
    dqkv0 = torch.zeros_like(qkv)
    ob.attn_dvdk(p_v, ds0, dao, qkv, dqkv0, H)
    rt = ob.relpos_tables(pe, T, 1000, want_t=True)
    win, idx, win_t = rt["window"]
    W = win.shape[0]
    acc0 = torch.zeros((W, 64), dtype=torch.float32, device=DEV)
    ob.attn_pos_bwd_ds(ds0, win_t, q_rows, dq10, dqkv0, acc0, H)
    # the fused pass
    ds1, parts, dqkv1 = ob.mha_bwd_fused(qkv, pe_pack, lse, dao, ao2, H, drop=cfg, dqkv=torch.zeros_like(qkv))
    acc1 = torch.zeros((W, 64), dtype=torch.float32, device=DEV)
    ob.attn_pos_bwd_parts(ds1, win_t, qkv, parts, dqkv1, acc1, H)
    torch.cuda.synchronize()
    a, b_ = ds1.float(), ds0.float()
    assert float((a - b_).abs().max()) <= 2.0 ** -6 * float(b_.abs().max()) and froerr(a, b_.cpu()) < 4e-3
    assert bool((ds1[..., T:] == 0).all())
    g = qkv.view(B, T, 3, H, 64)
    q, k = (g[:, :, i].permute(0, 2, 1, 3).float().cpu().double() for i in (0, 1))
    do = dao.view(B, T, H, 64).permute(0, 2, 1, 3).float().cpu().double()
    dsd = ds1[..., :T].float().cpu().double()
    d1 = dqkv1.view(B, T, 3, H, 64)
    # tight: the products of the kernel's own dS, and dV against the old path's P' (the same probabilities up to summation order)
    assert relerr(parts.sum(0), dsd @ k) < 2e-5
    assert relerr(d1[:, :, 1].permute(0, 2, 1, 3).float(), dsd.transpose(-1, -2) @ q) < 6e-3       # (bf16 output)
    assert relerr(d1[:, :, 2].permute(0, 2, 1, 3).float(), p_v[..., :T].float().cpu().double().transpose(-1, -2) @ do) < 6e-3
    # the whole backward against the old path
    assert froerr(dqkv1.float(), dqkv0.float().cpu()) < 6e-3
    assert relerr(acc1, acc0.double().cpu()) < 5e-3
    # deterministic
    ds2, parts2, dqkv2 = ob.mha_bwd_fused(qkv, pe_pack, lse, dao, ao2, H, drop=cfg, dqkv=torch.zeros_like(qkv))
    assert torch.equal(ds2, ds1) and torch.equal(parts2, parts) and torch.equal(dqkv2[..., D:], dqkv1[..., D:])


@pytest.mark.parametrize("B,H,T", [(2, 4, 100), (1, 4, 7), (3, 2, 65), (2, 4, 400), (1, 8, 1100), (32, 4, 400)])
def test_attn_dvdk_streaming(ob, B, H, T):
    """csship_attn_dvdk: dV = P'^T dO and dK = dS^T q in one pass over the two [B,H,T,ldp] operands (the attention's
    value / key gradients, conformer.py:236-253 under autograd) against float64 on the same bf16 operands; keys split over 1, 2 and 5+ workgroups per head, ragged last blocks."""
    ldp = (T + 7) // 8 * 8
    p = torch.zeros(B, H, T, ldp)
    ds = torch.zeros(B, H, T, ldp)
    p[..., :T] = torch.softmax(rnd((B, H, T, T), 300), -1)
    ds[..., :T] = rnd((B, H, T, T), 301, 0.05)
    qkv = rnd((B, T, 3 * H * 64), 302)
    dao = rnd((B * T, H * 64), 303)
    pb, dsb, qb, dob = (x.to(DEV).to(torch.bfloat16).contiguous() for x in (p, ds, qkv, dao))
    dqkv = torch.full_like(qb, float("nan"))
    ob.attn_dvdk(pb, dsb, dob, qb, dqkv, H)
    torch.cuda.synchronize()
    g = dqkv.view(B, T, 3, H, 64)
    q = qb.view(B, T, 3, H, 64)[:, :, 0].permute(0, 2, 1, 3).double().cpu()          # [B,H,T,64]
    do = dob.view(B, T, H, 64).permute(0, 2, 1, 3).double().cpu()
    dv_ref = pb[..., :T].double().cpu().transpose(-1, -2) @ do
    dk_ref = dsb[..., :T].double().cpu().transpose(-1, -2) @ q
    dv = g[:, :, 2].permute(0, 2, 1, 3).double().cpu()
    dk = g[:, :, 1].permute(0, 2, 1, 3).double().cpu()
    assert bool(torch.isnan(g[:, :, 0].float()).all())            # the q third is not touched
    assert bool(torch.isfinite(dv).all()) and bool(torch.isfinite(dk).all())
    assert relerr(dv, dv_ref) < 6e-3 and relerr(dk, dk_ref) < 6e-3   # bf16 output rounding


def test_convmod_backward_next_operand(ob):
    """csship_convmod_bwd_next: the third result is bf16(dropout(dx)) under the next operator's cfg (what
    csship_dropout_apply + the GEMM's rounding give), dx and the parameter gradients unchanged."""
    from css_amd import ops
    from css_amd.dropout import DropState
    B, T, D, K = 4, 130, 256, 33
    x = rnd((B, T, D), 30).to(DEV)
    dy = rnd((B, T, D), 31).to(DEV)
    W = {"ln_g": 1 + 0.1 * rnd((D,), 32), "ln_b": 0.1 * rnd((D,), 33), "pw1_w": rnd((2,), 34), "pw1_b": rnd((2,), 35),
         "dw_w": rnd((K, D), 36, 0.2), "dw_b": 0.1 * rnd((D,), 37), "bn_g": 1 + 0.1 * rnd((D,), 38), "bn_b": 0.1 * rnd((D,), 39),
         "bn_mean": torch.zeros(D), "bn_var": torch.ones(D), "pw2_w": rnd((1,), 40) + 1.0, "pw2_b": rnd((1,), 41)}
    W = {k: v.to(DEV).contiguous() for k, v in W.items()}
    _, fws = ops.convmod(x, W, K, batch_stats=True, return_workspace=True)
    st = DropState(torch.tensor([99], dtype=torch.int32, device=DEV), 0.1, 0)
    dx0, g0 = ob.convmod_bwd(x, dy, W, fws, K)
    for cfg in (None, st.cfg("2.attn.out")):
        dx, g, nxt = ob.convmod_bwd(x, dy, W, fws, K, want_next=True, next_drop=cfg)
        assert torch.equal(dx, dx0) and all(torch.equal(g[k], g0[k]) for k in g0)
        want = dx0.view(B * T, D) if cfg is None else ops.dropout_apply(dx0.view(B * T, D), cfg, out=torch.empty(B * T, D, device=DEV))
        assert torch.equal(nxt, want.to(torch.bfloat16))


def test_bgemm_band_hint_skips_only_zero_tiles(ob):
    """csship_bgemm_params.band: a row-major A that is zero outside m <= k <= m + band gives the same product bit for bit
    with the hint (batched, strided, f32 addend, bf16 output - the skewed-dS product of the attention backward)."""
    Bz, H, T = 3, 2, 400
    W = (2 * T - 1 + 7) // 8 * 8
    dsk = torch.zeros(Bz, T, H, W, dtype=torch.bfloat16, device=DEV)
    i = torch.arange(T, device=DEV)
    vals = rnd((Bz, T, H, T), 200).to(DEV).to(torch.bfloat16)
    for b in range(Bz):
        for h in range(H):
            dsk[b, i[:, None].expand(T, T), h, i[:, None] + torch.arange(T, device=DEV)[None, :]] = vals[b, :, h]
    win = rnd((W, 64), 201).to(DEV).to(torch.bfloat16)
    add = rnd((Bz, H, T, 64), 202).to(DEV)
    a = dsk.permute(0, 2, 1, 3)
    want = ob.bgemm(a, win[None, None], b_kmajor=True, add=add, out_bf16=True)
    got = ob.bgemm(a, win[None, None], b_kmajor=True, add=add, out_bf16=True, band=T - 1)
    assert torch.equal(got, want)
    ref = a.float().cpu().double() @ win.float().cpu().double() + add.cpu().double()
    assert relerr(got.float(), ref) < 1e-2
    # the same matrix as the reduction-strided left operand of d(pe_window) = dS_skew^T q: rows (b, t, h), one K-slab per b
    q_rows = rnd((Bz * T * H, 64), 203).to(DEV).to(torch.bfloat16)
    acc = rnd((W, 64), 204).to(DEV)
    a2 = dsk.view(Bz * T * H, W)
    want2 = ob.bgemm(a2, q_rows, a_kmajor=True, b_kmajor=True, add=acc, splits=Bz)
    got2 = ob.bgemm(a2, q_rows, a_kmajor=True, b_kmajor=True, add=acc, splits=Bz, band=T - 1, band_stride=H)
    assert torch.equal(got2, want2)
    assert relerr(got2, a2.float().cpu().double().T @ q_rows.float().cpu().double() + acc.cpu().double()) < 1e-4


@pytest.mark.parametrize("T", [37, 400, 449, 460])
def test_relpos_tables_match_the_separate_packers(ob, T):
    """one gather through a cached index == ops.pack_relpos + backward.relpos_window + pe_window_t, bit for bit"""
    from css_amd import backward as bw, ops
    pe = (torch.randn(2000, 64, generator=torch.Generator().manual_seed(T)) * 0.3).to(DEV)
    for _ in range(2):   # second pass: the cached index
        rt = ob.relpos_tables(pe, T, 1000, want_t=T <= 449)
        pack, qmin, qmax = ops.pack_relpos(pe, 1000)
        assert rt["pack"][1:] == (qmin, qmax) and torch.equal(rt["pack"][0], pack)
        assert torch.equal(rt["pe_bf"], pe.to(torch.bfloat16))
        win, idx = bw.relpos_window(pe.to(torch.bfloat16), T, 1000)
        assert torch.equal(rt["window"][0], win) and torch.equal(rt["window"][1], idx)
        assert len(rt["window"]) == (3 if T <= 449 else 2)
        if T <= 449:
            assert torch.equal(rt["window"][2], ob.pe_window_t(win, T))
        for t in (rt["pack"][0], rt["window"][0]) + tuple(rt["window"][2:]):
            assert t.data_ptr() % 16 == 0


def test_pit_backward_with_the_upstream_gradient_on_the_device(ob):
    """csship_pit_mse_bwd_dev: the kernel's own multiply by the device scalar == a second pass that multiplies, bit for bit"""
    from css_amd import ops
    B, T, F, K = 3, 50, 257, 2
    g = torch.Generator().manual_seed(5)
    masks = torch.zeros((B, T, 772), device=DEV)[..., :771]
    masks.copy_(torch.rand(B, T, 771, generator=g))
    f = torch.rand(B, T, F, generator=g).to(DEV)
    tg = [torch.rand(B, T, F, generator=g).to(DEV) for _ in range(K)]
    _, perm, _, _ = ops.pit_mse(masks, f, tg)
    gl = torch.tensor(0.37, device=DEV)
    a = ops.pit_mse_bwd(masks, f, tg, perm, 3, 1.0, grad_loss_dev=gl)
    b = ops.pit_mse_bwd(masks, f, tg, perm, 3, 1.0)
    assert a.stride() == masks.stride() and torch.equal(a, b * gl) and b.abs().max() > 0
    assert torch.equal(ops.pit_mse_bwd(masks, f, tg, perm, 3, 1.0, grad_loss_dev=torch.ones((), device=DEV)), b)


def test_ema_update_running_statistics(ob):
    """csship_ema_update: dst = keep * dst + w * src over a table (BatchNorm running mean / variance of every block)"""
    g = torch.Generator().manual_seed(9)
    n = 70   # 140 tensors: two launches
    means, nm = [torch.randn(256, generator=g).to(DEV) for _ in range(n)], [torch.randn(256, generator=g).to(DEV) for _ in range(n)]
    vrs, nv = [torch.rand(256, generator=g).to(DEV) for _ in range(n)], [torch.rand(1024, generator=g).to(DEV)[256:512] for _ in range(n)]
    w_t = 0.1 * 12800 / 12799
    want_m = [0.9 * a.double() + 0.1 * b.double() for a, b in zip(means, nm)]
    want_v = [0.9 * a.double() + w_t * b.double() for a, b in zip(vrs, nv)]
    ob.ema_update(means, nm, 0.1, vrs, nv, w_t, 0.9)
    for got, want in zip(means + vrs, want_m + want_v):
        assert torch.allclose(got.double(), want, rtol=3e-7, atol=1e-7)
    with pytest.raises(RuntimeError):
        ob.ema_update([means[0]], [nm[0][:128]], 0.1, [], [], 0.1, 0.9)
