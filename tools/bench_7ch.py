"""C3 shape: 7-channel STFT + IPD features (idim 1799) -> Conformer-base forward -> PIT, 32 x 400 frames."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import ops
from css_amd.executor.feature import FeatureExtractor
from css_amd.models import MODELS
from css_amd.utils import synthetic as syn

dev = "cuda"
B, T = 32, 400
n = 512 + 256 * (T - 1)
mix, src = syn.synthetic_sources(B, n, seed=0)
x = syn.synthetic_multichannel(src, n_ch=7).to(dev)          # [B, 7, n]
tg_audio = src.to(dev)                                        # [2, B, n]
fe = FeatureExtractor(ipd_index="1,0;2,0;3,0;4,0;5,0;6,0")
idim = fe.feature_dim
conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=16, kernel_size=33, dropout_rate=0.1,
            relative_pos_emb=True)
m = MODELS["Conformer"](idim, 257, 2, 1, conf)
m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(idim=idim), 0))
m = m.to(dev).eval()


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def feats():
    return fe.forward(x)


def feats_tf():
    return fe.forward_tf(x, want_spectra=False)


def whole():
    with torch.no_grad():
        mag, f, _, _ = fe.forward_tf(x, want_spectra=False)
        tg = ops.stft(tg_audio, ("mag",))["mag"]
        masks = m.forward_masks(f)
        loss, _, _, _ = ops.pit_mse(masks, mag, [tg[0], tg[1]])
    return loss


tf_, _ = timeit(feats)
tt, _ = timeit(feats_tf)
print(f"7ch features, frame-major (forward_tf): {tt:.3f} ms = {B * T * 14364 / tt / 1e6:.1f} GB/s algorithmic")
tw, loss = timeit(whole)
print(f"7ch features, reference layout (forward; STFT x7 + IPD + transposes, idim {idim}): {tf_:.3f} ms = {B * T * 14364 / tf_ / 1e6:.1f} GB/s algorithmic")
print(f"7ch features + Conformer-base + PIT: {tw:.3f} ms/step = {B * T / tw / 1e3:.3f} M frames/s  loss {float(loss):.4f}")
