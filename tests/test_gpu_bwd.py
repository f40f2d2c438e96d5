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


@pytest.fixture(scope="module")
def ob():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from css_amd import ops_bwd as o
    o.lib()
    return o


@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 100), (257, 771, 1300), (64, 256, 12800)])
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


def test_colsum(ob):
    for M, N, dt in ((12800, 771, torch.float32), (1000, 1024, torch.bfloat16), (7, 3, torch.float32)):
        ld = (N + 7) // 8 * 8
        x = rnd((M, ld), 10).to(DEV).to(dt)[:, :N]
        got = ob.colsum(x)
        assert relerr(got, x.float().cpu().double().sum(0)) < 1e-5
