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


def test_fused_conv_path_matches():
    """Optional eval path: conv module split over csship_outproj_glu + csship_encoder_chain (panels straddle
    utterance boundaries: T = 70, 64-row panels)."""
    m, sd = build(2, 32)
    f = rnd((3, 70, 257), 21).abs().to(DEV)
    ref = m.forward_masks(f).clone()
    m.fuse_conv = True
    got = m.forward_masks(f)
    m.fuse_conv = False
    assert mask_mse(got, ref) < 1e-6
    o = oracle.conformer.forward(f.cpu(), sd, num_blocks=2)[1]
    assert mask_mse(got, torch.cat(o, 2)) < 1e-4


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
