"""GPU parity of the inference executors: Stitcher vs the reference golden, Separator vs the oracle."""
import numpy as np
import pytest
import torch

import oracle
from css_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


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
    return torch.stack(masks), torch.stack(mags)


def test_stitcher_golden(golden):
    from css_amd.executor import Stitcher
    g = golden("stitcher.npz")
    masks, mags = _stitch_inputs()
    st = Stitcher({"eval_win": 2.4, "eval_hop": 0.8, "hop_size": 256}, 16000)
    assert st.stitch_margin == int(g["margin"])
    perm = st.get_stitch(mags.to(DEV), masks.to(DEV))
    assert np.array_equal(np.array(perm), g["perm"])
    r1, r2, rn = st.get_connect(perm, masks.to(DEV))
    for got, key in ((r1, "res1"), (r2, "res2"), (rn, "noise")):
        assert float((got.cpu() - torch.from_numpy(g[key])).abs().max()) < 1e-6


def test_separator_windows_vs_oracle():
    from css_amd.executor import Separator
    from css_amd.models import MODELS
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=1, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    model = MODELS["Conformer"](257, 257, 2, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=1), 5)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    cfg = {"batch_size": 2, "eval_win": 2.4, "eval_hop": 0.8, "frame_length": 512, "frame_shift": 256}
    sep = Separator(cfg, 16000, device=DEV, backend="hip", model=model)
    rec, _ = syn.synthetic_sources(1, 12800 * 3 + 38656 + 77, seed=9)  # 1 channel, 4 windows, ragged tail dropped
    masks, mags = sep.separate(rec)
    assert masks.shape == (4, 257, 150, 3) and mags.shape == (4, 257, 150)
    win = rec.unfold(-1, 38656, 12800)[0]  # [4, 38656]
    o_mag, o_f, _, _ = oracle.feature.feature_extractor(win)
    _, o_masks = oracle.conformer.forward(o_f.transpose(1, 2), sd, num_blocks=1)
    ref = torch.stack(o_masks, -1).permute(0, 2, 1, 3).clamp(max=1.0)  # [W,257,T,3]
    assert float((mags.cpu() - o_mag).abs().max()) < 1e-4 * float(o_mag.max())
    assert float(((masks.cpu() - ref) ** 2).mean()) < 1e-4
    assert float(masks.max()) <= 1.0


def test_synthetic_dataset_batches_and_validate():
    """The on-GPU dataset yields the reference's batch dict; validate() runs on it end to end."""
    from css_amd.datasets import DATASETS
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    from css_amd.trainer import validate
    ds = DATASETS["SyntheticSeparation"](batch_size=2, frames=40, seed=3)
    b = next(ds)
    assert set(b) == {"mix", "source1", "source2", "lens", "ovl"} and b["mix"].shape == (2, 40, 257)
    mix, src = syn.synthetic_sources(2, 512 + 256 * 39, seed=3)
    ref = oracle.feature.stft(mix)[0].transpose(1, 2)
    assert float((b["mix"].cpu() - ref).abs().max()) < 1e-4 * float(ref.max())
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=1, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    model = MODELS["Conformer"](257, 257, 2, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=1), 5)
    model.load_state_dict(sd)
    model = model.to(DEV)
    loss = validate(ds, model, OBJECTIVES["MSE"].build_objective({}), device=DEV, num_batches=3)
    assert loss.dim() == 0 and torch.isfinite(loss) and model.training
