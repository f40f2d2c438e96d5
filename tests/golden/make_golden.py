#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (desh2608/css) on CPU.

Run ONLY in the build container, where the read-only checkout is mounted:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [/root/reference]

The reference never travels to the GPU box: only the small .npz/.json outputs
written next to this script are committed.  Inputs and weights are regenerated
from seeds by css_amd.utils.synthetic on both sides, so only expected OUTPUTS
(plus small input checksums) are stored.

What is pinned:
  stft.npz        feature.py STFT / FeatureExtractor (1ch, 7ch+IPD)
  conformer.npz   conformer.py sub-modules + full forward (2 and 16 blocks) + PIT loss
  blstm.npz       blstm.py tiny model masks
  pit.npz         pit.py K=2 / K=3 losses and permutations
  stitcher.npz    stitcher.py perms + stitched masks
  trainer.json    lr_scheduler.py LR trace, iteration_types.py 6-step loss trace
  keys.json       state_dict keys/shapes of the reference models
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from css_amd.utils import synthetic as syn  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)

from css.executor.feature import FeatureExtractor  # noqa: E402
from css.executor.stitcher import Stitcher  # noqa: E402
from css.models.conformer import (Conformer, ConvModule, EncoderLayer, FeedForward,  # noqa: E402
                                  MultiHeadedAttention, RelativePositionalEncoding)
from css.models.blstm import BLSTM  # noqa: E402
from css.objectives.pit import pit_loss  # noqa: E402
from css.trainer import LRScheduler, train_one_epoch  # noqa: E402


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def save(name, **arrs):
    out = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: tuple(v.shape) for k, v in out.items()})


def load_synth(model, seed):
    sd = syn.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd)
    model.eval()
    return sd


# ---------------------------------------------------------------- keys
keys = {
    "conformer_base": [[k, list(v.shape)] for k, v in Conformer().state_dict().items()],
    "blstm_default": [[k, list(v.shape)] for k, v in BLSTM().state_dict().items()],
}
assert [k for k, _ in keys["conformer_base"]] == list(syn.conformer_shapes().keys())
assert [k for k, _ in keys["blstm_default"]] == list(syn.blstm_shapes().keys())
json.dump(keys, open(os.path.join(HERE, "keys.json"), "w"))

# ---------------------------------------------------------------- STFT / features
with torch.no_grad():
    n_short = 512 + 256 * 19  # T = 20
    mix, src = syn.synthetic_sources(2, n_short, seed=3)
    x7 = syn.synthetic_multichannel(src, n_ch=7, seed=3)
    fe1 = FeatureExtractor(512, 256)
    fe7 = FeatureExtractor(512, 256, ipd_index="1,0;2,0;3,0;4,0;5,0;6,0")
    mag1, f1, re1, im1 = fe1.forward(mix)
    mag7, f7, re7, im7 = fe7.forward(x7)
    # one full inference window (config 4): 38 656 samples -> 150 frames
    mixw, _ = syn.synthetic_sources(1, 38656, seed=4)
    magw, fw, _, _ = fe1.forward(mixw)
    save("stft.npz", mix_sum=mix.double().sum(), x7_sum=x7.double().sum(),
         mag1=mag1, f1=f1, re1=re1, im1=im1, mag7=mag7, f7=f7, re7=re7, im7=im7,
         magw=magw, fw=fw)

# ---------------------------------------------------------------- Conformer
D, H, FFN, KS = 256, 4, 1024, 33
with torch.no_grad():
    out = {}
    x = rnd((2, 64, D), 11)
    out["x_sum"] = x.double().sum()

    ffn = FeedForward(D, FFN, 0.1)
    load_synth(ffn, 21)
    out["ffn"] = ffn(x)

    pe = RelativePositionalEncoding(D // H, 1000, False)
    load_synth(pe, 22)
    T = x.shape[1]
    pos = torch.arange(T)
    pos_k, _ = pe(pos[:, None] - pos[None, :])
    mha = MultiHeadedAttention(H, D, 0.1)
    load_synth(mha, 23)
    out["mha"] = mha(x, pos_k, None)

    conv = ConvModule(D, KS, 0.1)
    load_synth(conv, 24)
    out["conv_eval"] = conv(x)
    # train-mode BN (batch statistics) with dropout disabled
    conv.train()
    conv.dropout.p = 0.0
    out["conv_trainbn"] = conv(x)
    conv.eval()

    lay = EncoderLayer(D, H, FFN, KS, 0.1)
    load_synth(lay, 25)
    out["layer"] = lay(x, pos_k, None)

    # full models
    f_in = rnd((2, 100, 257), 12).abs()
    out["f_sum"] = f_in.double().sum()
    tg = rnd((2, 2, 100, 257), 13).abs() * 0.5
    for nb in (2, 16):
        conf = dict(attention_dim=D, attention_heads=H, linear_units=FFN, num_blocks=nb,
                    kernel_size=KS, dropout_rate=0.1, relative_pos_emb=True)
        m = Conformer(257, 257, 2, 1, conf)
        load_synth(m, 30 + nb)
        y_pred, masks = m(f_in)
        out[f"masks_b{nb}"] = torch.stack(masks, 0)
        loss = sum(pit_loss(y_pred[b], tg[b], axis=0) for b in range(2)) / 2
        out[f"loss_b{nb}"] = loss
    # long sequence: T > 1000 exercises the clamp of relative positions
    conf = dict(attention_dim=D, attention_heads=H, linear_units=FFN, num_blocks=1,
                kernel_size=KS, dropout_rate=0.1, relative_pos_emb=True)
    m = Conformer(257, 257, 2, 1, conf)
    load_synth(m, 41)
    f_long = rnd((1, 1100, 257), 14).abs()
    _, masks = m(f_long)
    out["masks_long_head"] = torch.stack(masks, 0)[:, :, :40]   # frames 0..39
    out["masks_long_tail"] = torch.stack(masks, 0)[:, :, -40:]  # frames 1060..1099
    # 7ch input dim and 3spk+noise head (configs 3 / 5 shapes, 1 block)
    m = Conformer(1799, 257, 3, 1, conf)
    load_synth(m, 42)
    f7in = rnd((1, 50, 1799), 15)
    f7in[..., :257] = f7in[..., :257].abs()
    try:
        m(f7in)
        out["seven_ch_runs"] = np.array(1)
    except Exception as e:  # mask * f_orig needs F == idim in the reference
        out["seven_ch_runs"] = np.array(0)
        print("reference Conformer(idim=1799) forward raises:", type(e).__name__, e)
    save("conformer.npz", **out)

# ---------------------------------------------------------------- BLSTM tiny
with torch.no_grad():
    m = BLSTM(257, 257, 2, 1, dict(hidden_dim=64, num_layers=2, dropout_rate=0.1))
    load_synth(m, 50)
    fb = rnd((2, 30, 257), 16).abs()
    y_pred, masks = m(fb)
    save("blstm.npz", masks=torch.stack(masks, 0), y_pred=y_pred)

# ---------------------------------------------------------------- PIT
with torch.no_grad():
    out = {}
    for K in (2, 3):
        est = rnd((K, 40, 257), 60 + K).abs()
        tgt = rnd((K, 40, 257), 70 + K).abs()
        l, p = pit_loss(est, tgt, axis=0, return_permutation=True)
        out[f"loss_k{K}"], out[f"perm_k{K}"] = l, np.array(p)
        # estimate = rotated noisy copy of target -> known permutation
        rot = list(range(1, K)) + [0]
        est2 = tgt[rot] + 0.01 * rnd((K, 40, 257), 80 + K)
        l2, p2 = pit_loss(est2, tgt, axis=0, return_permutation=True)
        out[f"loss_rot_k{K}"], out[f"perm_rot_k{K}"] = l2, np.array(p2)
    save("pit.npz", **out)

# ---------------------------------------------------------------- Stitcher
with torch.no_grad():
    W, Fb, Tw, hop = 6, 257, 150, 50
    cfg = {"eval_win": 2.4, "eval_hop": 0.8, "hop_size": 256}
    st = Stitcher(cfg, 16000)
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
    masks_t, mags_t = torch.stack(masks), torch.stack(mags)
    perm = st.get_stitch(mags_t, masks_t)
    r1, r2, rn = st.get_connect(perm, masks_t)
    save("stitcher.npz", perm=np.array(perm), margin=np.array(st.stitch_margin), res1=r1, res2=r2, noise=rn)

# ---------------------------------------------------------------- trainer
tr = {}
opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
args = types.SimpleNamespace(warmup=4, decay=0.1, fixed=2, min_lr=1e-9)
sch = LRScheduler(opt, args)
trace = []
for _ in range(10):
    trace.append(sch.curr_lr)
    sch.step(1.0)
tr["lr_trace"] = trace

# 6-step loss trace of train_one_epoch on a tiny BLSTM (eval-mode forward: dropout off) with
# the tuple-shim objective of SURVEY quirk 1.
class ShimMSE(torch.nn.Module):
    def forward(self, model, sample, device="cpu"):
        y_pred = model(sample["mix"])[0]
        y_true = torch.stack((sample["source1"], sample["source2"]), dim=1)
        return sum(pit_loss(y_pred[b], y_true[b], axis=0) for b in range(y_pred.shape[0])) / y_pred.shape[0]

m = BLSTM(257, 257, 2, 1, dict(hidden_dim=32, num_layers=1, dropout_rate=0.0))
m.load_state_dict(syn.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 51))
m.train()

def gen():
    i = 0
    while True:
        yield {"mix": rnd((2, 20, 257), 200 + i).abs(), "source1": rnd((2, 20, 257), 300 + i).abs() * 0.5,
               "source2": rnd((2, 20, 257), 400 + i).abs() * 0.5, "ovl": 0.5}
        i += 1

opt = torch.optim.Adam(m.parameters(), lr=1e-3)
sch = LRScheduler(opt, types.SimpleNamespace(warmup=3, decay=0.0, fixed=0, min_lr=1e-9))
targs = types.SimpleNamespace(gpu=False, fp16=False, batches_per_epoch=1, grad_thresh=5.0)
g = gen()
losses = []
for _ in range(6):
    losses.append(train_one_epoch(targs, g, m, ShimMSE(), opt, sch, "cpu"))
tr["train_losses"] = losses
json.dump(tr, open(os.path.join(HERE, "trainer.json"), "w"), indent=1)
print("trainer.json", tr)
