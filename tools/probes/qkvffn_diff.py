"""which outputs of csship_qkv_ffn_bwd differ from the two launches, and by how much (debug probe)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from css_amd import ops, ops_bwd as ob
from css_amd.dropout import DropState
DEV = "cuda"
def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale
for with_ln2 in (False, True):
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
    try:
        (dg, db, y), got = ob.qkv_ffn_bwd(dqkv, wt, x_attn, dy, ag, ab, x, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5,
                                          drop_hidden=cfg_h, drop_out=cfg_o, ln2=ln2)
    except RuntimeError as e:   # (a library without the chained kernel: the two launches against themselves)
        print("  no chained kernel:", e)
        (dg, db, y), got = (dg1, db1, y1), ref
    try:
        (_, _, _), got2 = ob.qkv_ffn_bwd(dqkv, wt, x_attn, dy, ag, ab, x, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5,
                                         drop_hidden=cfg_h, drop_out=cfg_o, ln2=ln2)
    except RuntimeError:
        got2 = got
    ref2 = ob.ffn_bwd(x, dx1, G["g"], G["b"], w1p, G["b1"], w2tp, w1tp, FF, 0.5, cfg_h, cfg_o, ln2=ln2)
    torch.cuda.synchronize()
    print("  chained twice equal:", torch.equal(got[0], got2[0]), " two-launch twice equal:", torch.equal(ref[0], ref2[0]))
    d = (got[0] - ref[0]).abs()
    big = (d > 1e-5).nonzero()
    print("  big diffs:", big.shape[0], "first", big[:12].tolist(), " cols hist", torch.bincount(big[:, 1] // 32, minlength=8).tolist() if big.numel() else None)
    if not with_ln2:   # LayerNorm-backward invariants of (dx - dy): orthogonal to 1 and to xhat, row by row
        xh = torch.nn.functional.layer_norm(x.double(), (D,))
        for nm, r in (("chained", got[0]), ("two launches", ref[0])):
            e = r.double() - dx1.double()
            print(f"  {nm:13s} row 264: sum {float(e[264].sum()):+.3e}  dot xhat {float((e[264] * xh[264]).sum()):+.3e};  worst row: sum {float(e.sum(1).abs().max()):.3e} dot {float((e * xh).sum(1).abs().max()):.3e}")
    if not with_ln2:
        dlt = (got[0].double() - ref[0].double())[264]
        A = torch.stack([torch.ones(D, dtype=torch.float64, device=dlt.device), xh[264]], 1)
        sol = torch.linalg.lstsq(A, dlt[:, None]).solution[:, 0]
        res = dlt - A @ sol
        print("  row 264 delta = %.3e + %.3e * xhat, residual max %.3e; delta[:6] %s; x row mean %.3f var %.3f" % (
            float(sol[0]), float(sol[1]), float(res.abs().max()), dlt[:6].tolist(), float(x[264].mean()), float(x[264].var())))
        print("  dy(row 264)[:4]", dx1[264, :4].tolist(), " |dy| max", float(dx1[264].abs().max()), " dx ref[:4]", ref[0][264, :4].tolist())
    if not with_ln2:
        w1b = G["w1"].to(torch.bfloat16).double()
        da = got[5].double() @ w1b                       # dh [M, FF] @ W1 [FF, D]
        gm = G["g"].double()
        xd = x.double()
        mu = xd.mean(1, keepdim=True); var = xd.var(1, unbiased=False, keepdim=True)
        rstd = (var + 1e-5).rsqrt(); xhd = (xd - mu) * rstd
        gg = da * gm
        true = dx1.double() + rstd * (gg - gg.mean(1, keepdim=True) - xhd * (gg * xhd).mean(1, keepdim=True))
        for nm, r in (("chained", got[0]), ("two launches", ref[0])):
            err = (r.double() - true).abs()
            if nm != "chained":
                rr = (r.double() - dx1.double())[264]; tt = (true - dx1.double())[264]
                big = tt.abs() > 0.05
                q = (rr / tt)[big]
                print("  row 264 (dx - dy) / truth: min %.6f max %.6f mean %.6f over %d elements; rstd %.6f" % (float(q.min()), float(q.max()), float(q.mean()), int(big.sum()), float(rstd[264])))
            if nm != "chained":
                s1 = gg.mean(1, keepdim=True); s2 = (gg * xhd).mean(1, keepdim=True)
                dap = (((r.double() - dx1.double()) / rstd) + s1 + xhd * s2) / gm
                print("   da true  ", [round(float(v), 6) for v in da[264, :8]])
                print("   da implied", [round(float(v), 6) for v in dap[264, :8]])
                print("   bf16(da)  ", [round(float(v), 6) for v in da[264, :8].float().to(torch.bfloat16).double()])
                print("   da implied, row 265", [round(float(v), 6) for v in dap[265, :4]], "true", [round(float(v), 6) for v in da[265, :4]])
            gsel = got if nm == "chained" else ref
            print(f"   {nm}: dbeta err vs float64 {float((gsel[2].double() - da.sum(0)).abs().max()):.3e}  dgamma err {float((gsel[1].double() - (da * xhd).sum(0)).abs().max()):.3e}")
            print(f"  {nm:13s} vs float64: max err {float(err.max()):.3e} at row {int(err.max(1).values.argmax())}; row 264 err {float(err[264].max()):.3e}; median row err {float(err.max(1).values.median()):.3e}")
    names = ["dx", "dgamma", "dbeta", "a", "h", "dh", "dyd", "dgamma2", "dbeta2"]
    print("ln2", with_ln2, "stage-1:", torch.equal(dg, dg1), torch.equal(db, db1), torch.equal(y, y1))
    for n, a, b in zip(names, got, ref):
        d = (a.float() - b.float()).abs()
        nz = int((d > 0).sum())
        rows = sorted(set((d > 0).nonzero()[:, 0].tolist()))[:10] if d.dim() == 2 and nz else []
        print(f"  {n:8s} differing {nz:8d} of {d.numel():8d}  max abs {float(d.max()):.3e}  ref max {float(b.float().abs().max()):.3e} rows {rows}")
