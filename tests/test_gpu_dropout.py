"""Training-mode dropout (conformer.py:127,249,254,302,317,319; blstm.py:147,158): the HIP kernels regenerate
counter-based masks from (seed, site, element index); the oracle applies the numpy statement of the same masks
(oracle/dropout.py, pinned against the reference's dropout placement by tests/golden/dropout.npz)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from css_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def _cfg(seed_t, stream, p):
    from css_amd import _lib
    c = _lib.DropoutCfg(seed_t.data_ptr(), stream, p)
    return c, C.pointer(c)


@pytest.mark.parametrize("seed,stream,p,n", [(1, 0, 0.1, 1000), (12345, 35, 0.1, 100003), (7, 200, 0.5, 4096), (3, 9, 0.0, 64)])
def test_device_mask_equals_numpy_mask(seed, stream, p, n):
    from css_amd import ops
    st = torch.tensor([seed], dtype=torch.int32, device=DEV)
    keep, ptr = _cfg(st, stream, p)
    got = ops.dropout_mask(n, ptr).cpu().numpy().astype(bool)
    assert np.array_equal(got, oracle.dropout.keep_mask(seed, stream, n, p))
    x = rnd((n // 4 * 4,), 5).to(DEV).view(-1, 4)
    y = ops.dropout_apply(x.clone(), ptr)
    want = x.cpu() * torch.from_numpy(oracle.dropout.keep_mask(seed, stream, x.numel(), p).reshape(-1, 4).astype(np.float32)
                                      * np.float32(oracle.dropout.scale(p)))
    assert torch.allclose(y.cpu(), want, rtol=1e-6, atol=0)


@pytest.mark.parametrize("p_drop", [0.1, 0.0])
def test_fused_feedforward_dropout_vs_oracle(p_drop):
    """csship_ffn_fused_drop: hidden and output dropout inside the fused FeedForward kernel vs the oracle with the same
    masks; the backward kernel (csship_ffn_bwd_drop) vs autograd through that oracle."""
    from css_amd import backward as bw, ops, ops_bwd as ob
    from css_amd.dropout import SITES
    M, D, FF = 150, 256, 1024
    sd = {"ff.layer_norm.weight": 1 + 0.1 * rnd((D,), 1), "ff.layer_norm.bias": 0.1 * rnd((D,), 2),
          "ff.net.0.weight": rnd((FF, D), 3, 1 / 16), "ff.net.0.bias": rnd((FF,), 4, 0.1),
          "ff.net.3.weight": rnd((D, FF), 5, 1 / 32), "ff.net.3.bias": rnd((D,), 6, 0.1)}
    x = rnd((M, D), 7)
    seed = torch.tensor([99], dtype=torch.int32, device=DEV)
    kh, ph = _cfg(seed, 17, p_drop)
    ko, po = _cfg(seed, 18, p_drop)
    d = {k: v.to(DEV) for k, v in sd.items()}
    w1p, w2p = ops.pack_frag(d["ff.net.0.weight"]), ops.pack_frag(d["ff.net.3.weight"])
    out = ops.ffn_fused(x.to(DEV), (d["ff.layer_norm.weight"], d["ff.layer_norm.bias"]), w1p, d["ff.net.0.bias"], w2p,
                        d["ff.net.3.bias"], FF, 0.5, drop_hidden=ph, drop_out=po)
    streams = {"s.hidden": 17, "s.out": 18}
    drop = oracle.dropout.Masks(99, p_drop, lambda site: streams[site])
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xg = x.clone().requires_grad_(True)
    ref = xg + 0.5 * oracle.conformer.feed_forward(xg, sdg, "ff", drop, "s")
    err = float((out.cpu() - ref.detach()).abs().max())
    print(f"[parity] fused FeedForward with dropout vs oracle: max abs {err:.3e}")
    assert err < 3e-2 and float(((out.cpu() - ref.detach()) ** 2).mean()) < 1e-5
    # without the masks the oracle is far away: the comparison is not vacuous
    if p_drop > 0:
        assert float((out.cpu() - (x + 0.5 * oracle.conformer.feed_forward(x, sd, "ff"))).abs().max()) > 0.1
    dy = rnd((M, D), 8, 0.1)
    ref.backward(dy)
    w2tp, w1tp = ops.pack_frag(d["ff.net.3.weight"].t().contiguous()), ops.pack_frag(d["ff.net.0.weight"].t().contiguous())
    dx, g = bw.ffn_bwd_fused(x.to(DEV), dy.to(DEV), d["ff.layer_norm.weight"], d["ff.layer_norm.bias"], d["ff.net.0.bias"],
                             w1p, w2tp, w1tp, FF, 0.5, drop_hidden=ph, drop_out=po)
    names = ["ff.layer_norm.weight", "ff.layer_norm.bias", "ff.net.0.weight", "ff.net.0.bias", "ff.net.3.weight", "ff.net.3.bias"]
    rels = {}
    for nm, gv in zip(names, g):
        r = sdg[nm].grad
        rels[nm] = float((gv.cpu().reshape(r.shape) - r).norm() / r.norm())
    rels["dx"] = float((dx.cpu() - xg.grad).norm() / xg.grad.norm())
    print(f"[parity] FeedForward backward, p = {p_drop}: relative errors vs oracle autograd", {k: round(v, 4) for k, v in rels.items()})
    # bf16 operands in forward and backward against an fp32 graph: the same 2-5 % with and without dropout (a missing
    # 1 / (1 - p) would read > 10 %, a wrong mask > 30 %)
    assert all(v < 6e-2 for v in rels.values()), rels


def _build(nb, seed, p=0.1):
    from css_amd.models import MODELS
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=nb, kernel_size=33, dropout_rate=p,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=nb), seed))
    return m.to(DEV)


def test_conformer_training_forward_with_dropout_vs_oracle():
    """Train mode: batch-statistics BatchNorm + all seven dropout sites per block + the embedding's, through the HIP
    training forward, vs the oracle applying the same masks; the seed advances per forward (fresh masks) and eval mode
    drops nothing."""
    import grad_ref
    m = _build(2, 71).train()
    f = rnd((3, 52, 257), 14).abs().to(DEV)
    masks = m.forward_masks(f)
    assert masks.requires_grad
    s1 = int(m._drop_seed.item())
    ref, _ = grad_ref.oracle_masks(m, f, training_bn=True, drop=grad_ref.drop_of(m))
    mse = float(((masks.detach().cpu() - ref.detach()) ** 2).mean())
    ref0, _ = grad_ref.oracle_masks(m, f, training_bn=True)
    mse0 = float(((masks.detach().cpu() - ref0.detach()) ** 2).mean())
    print(f"[parity] train-mode forward with dropout: mask MSE vs oracle+masks {mse:.3e} (vs oracle without masks {mse0:.3e})")
    assert mse < 1e-4 and mse0 > 20 * mse
    masks2 = m.forward_masks(f)
    assert int(m._drop_seed.item()) == s1 + 1 and not torch.equal(masks2, masks)
    m.eval()
    with torch.no_grad():
        e1, e2 = m.forward_masks(f), m.forward_masks(f)
    assert torch.equal(e1, e2) and int(m._drop_seed.item()) == s1 + 1


def test_blstm_training_forward_and_backward_with_dropout_vs_oracle():
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    m = MODELS["BLSTM"](257, 257, 2, 1, dict(hidden_dim=64, num_layers=2, dropout_rate=0.1))
    m.load_state_dict(syn.fill_state_dict(syn.blstm_shapes(hidden_dim=64, num_layers=2), 53))
    m = m.to(DEV).train()
    B, T = 3, 25
    f = rnd((B, T, 257), 14).abs().to(DEV)
    tg = [(rnd((B, T, 257), 60 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    masks_ref, _ = grad_ref.oracle_masks(m, f, True, grad_ref.drop_of(m))
    ref_loss = float(grad_ref.pit_loss_with_perm(masks_ref, f.cpu(), tc, perm))
    assert abs(float(loss) - ref_loss) < 2e-3 * abs(ref_loss)
    ref = grad_ref.oracle_param_grads(m, lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm), f, training_bn=True)
    err, worst = grad_ref.compare(grads, ref)
    print(f"[parity] BLSTM with dropout: loss {float(loss):.5f} vs {ref_loss:.5f}; gradients rel err {err:.3e}, worst {worst}")
    assert err < 3e-2 and worst[1] < 0.15
