"""Deterministic synthetic weights and mixtures.

There is no network, no Lhotse and no checkpoint on the GPU box, so every
benchmark / parity input is generated here from integer seeds with numpy's
legacy ``RandomState`` (bit-stable across numpy versions and machines).

* ``fill_state_dict`` gives every tensor of a model ``state_dict`` non-trivial
  values derived from ``crc32(key) ^ seed`` -- independent of parameter creation
  order, so the golden script (which loads them into the *reference* model) and
  the GPU box (which loads them into ``css_amd.models``) agree bit-for-bit.
* ``synthetic_mixture`` restates the semantics of the reference's on-the-fly
  mixer (css/datasets/separation.py:163-267: two sources, the second delayed
  and zero padded, summed) on band-shaped Gaussian noise, as SURVEY.md 8(d)
  prescribes.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch


def _rng(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_tensor(key, shape, seed=0):
    """One tensor of a synthetic state dict (rule chosen from the key suffix)."""
    shape = tuple(int(s) for s in shape)
    r = _rng(key, seed)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    n = lambda: r.standard_normal(shape)  # noqa: E731
    if key.endswith("running_mean"):
        v = 0.1 * n()
    elif key.endswith("running_var"):
        v = 0.5 + r.random_sample(shape)
    elif "lstm.weight" in key:
        h = shape[-1] if "weight_hh" in key else shape[0] // 4
        v = (2.0 * r.random_sample(shape) - 1.0) / np.sqrt(h)
    elif "lstm.bias" in key:
        v = (2.0 * r.random_sample(shape) - 1.0) / np.sqrt(shape[0] // 4)
    elif "pw_conv_1.weight" in key or "pw_conv_2.weight" in key:
        v = 1.0 + 0.3 * n()
    elif "pw_conv_1.bias" in key or "pw_conv_2.bias" in key:
        v = 0.1 * n()
    elif "pe_k.weight" in key or "pe_v.weight" in key:
        v = 0.3 * n()
    elif key.endswith(".weight") and len(shape) == 1:  # LayerNorm / BatchNorm gain
        v = 1.0 + 0.1 * n()
    elif key.endswith(".bias"):
        v = 0.05 * n()
    elif key.endswith(".weight"):  # Linear [out,in] / depthwise conv [C,1,k]
        fan_in = int(np.prod(shape[1:]))
        v = n() / np.sqrt(fan_in)
    else:
        v = 0.1 * n()
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def fill_state_dict(shapes, seed=0):
    """shapes: mapping key -> shape.  Returns OrderedDict key -> tensor."""
    return OrderedDict((k, synth_tensor(k, s, seed)) for k, s in shapes.items())


def conformer_shapes(idim=257, num_bins=257, num_spk=2, num_noise=1, attention_dim=256,
                     attention_heads=4, linear_units=1024, num_blocks=16, kernel_size=33,
                     relative_pos_emb=True):
    """state_dict keys/shapes of the reference Conformer (css/models/conformer.py:63-150)."""
    d, f = attention_dim, linear_units
    s = OrderedDict()
    s["conformer.embed.0.weight"] = (d, idim)
    s["conformer.embed.0.bias"] = (d,)
    s["conformer.embed.1.weight"] = (d,)
    s["conformer.embed.1.bias"] = (d,)
    if relative_pos_emb:
        s["conformer.pos_emb.pe_k.weight"] = (2000, d // attention_heads)
    for l in range(num_blocks):
        p = f"conformer.encoders.{l}."
        def ffn(ff):
            s[p + ff + ".layer_norm.weight"] = (d,)
            s[p + ff + ".layer_norm.bias"] = (d,)
            s[p + ff + ".net.0.weight"] = (f, d)
            s[p + ff + ".net.0.bias"] = (f,)
            s[p + ff + ".net.3.weight"] = (d, f)
            s[p + ff + ".net.3.bias"] = (d,)

        ffn("feed_forward_in")
        s[p + "self_attn.layer_norm.weight"] = (d,)
        s[p + "self_attn.layer_norm.bias"] = (d,)
        for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
            s[p + f"self_attn.{nm}.weight"] = (d, d)
            s[p + f"self_attn.{nm}.bias"] = (d,)
        s[p + "conv.layer_norm.weight"] = (d,)
        s[p + "conv.layer_norm.bias"] = (d,)
        s[p + "conv.pw_conv_1.weight"] = (2, 1, 1, 1)
        s[p + "conv.pw_conv_1.bias"] = (2,)
        s[p + "conv.dw_conv_1d.weight"] = (d, 1, kernel_size)
        s[p + "conv.dw_conv_1d.bias"] = (d,)
        s[p + "conv.BN.weight"] = (d,)
        s[p + "conv.BN.bias"] = (d,)
        s[p + "conv.BN.running_mean"] = (d,)
        s[p + "conv.BN.running_var"] = (d,)
        s[p + "conv.BN.num_batches_tracked"] = ()
        s[p + "conv.pw_conv_2.weight"] = (1, 1, 1, 1)
        s[p + "conv.pw_conv_2.bias"] = (1,)
        ffn("feed_forward_out")
        s[p + "layer_norm.weight"] = (d,)
        s[p + "layer_norm.bias"] = (d,)
    s["linear.weight"] = (num_bins * (num_spk + num_noise), d)
    s["linear.bias"] = (num_bins * (num_spk + num_noise),)
    return s


def blstm_shapes(idim=257, num_bins=257, num_spk=2, num_noise=1, hidden_dim=512, num_layers=4):
    """state_dict keys/shapes of the reference BLSTM (css/models/blstm.py:48-147)."""
    h = hidden_dim
    s = OrderedDict()
    s["blstm.embed.0.weight"] = (h, idim)
    s["blstm.embed.0.bias"] = (h,)
    s["blstm.embed.1.weight"] = (h,)
    s["blstm.embed.1.bias"] = (h,)
    for l in range(num_layers):
        p = f"blstm.encoders.{l}."
        for sfx in ("", "_reverse"):
            s[p + "lstm.weight_ih_l0" + sfx] = (4 * (h // 2), h)
            s[p + "lstm.weight_hh_l0" + sfx] = (4 * (h // 2), h // 2)
            s[p + "lstm.bias_ih_l0" + sfx] = (4 * (h // 2),)
            s[p + "lstm.bias_hh_l0" + sfx] = (4 * (h // 2),)
        s[p + "layer_norm.weight"] = (h,)
        s[p + "layer_norm.bias"] = (h,)
    s["linear.weight"] = (num_bins * (num_spk + num_noise), h)
    s["linear.bias"] = (num_bins * (num_spk + num_noise),)
    return s


def synthetic_sources(batch, n_samples, n_src=2, seed=0, noise=0.0):
    """Band-shaped Gaussian "speakers", later ones delayed + zero padded.

    Returns (mix [B,n], sources [n_src,B,n]) float32 tensors.  Source k>0 of chunk b
    starts at a delay drawn from U(0, n/2) (separation.py:176-190 semantics:
    offset mix, pad to mixture length).
    """
    r = np.random.RandomState((1000003 * seed + 17) % (2 ** 32))
    fir = np.array([0.25, 0.5, 0.25])
    src = np.zeros((n_src, batch, n_samples), dtype=np.float32)
    for k in range(n_src):
        raw = 0.1 * r.standard_normal((batch, n_samples + 2))
        shaped = fir[0] * raw[:, :-2] + fir[1] * raw[:, 1:-1] + fir[2] * raw[:, 2:]
        if k > 0:
            for b in range(batch):
                d = int(r.randint(0, n_samples // 2))
                shaped[b, d:] = shaped[b, : n_samples - d].copy()
                shaped[b, :d] = 0.0
        src[k] = shaped.astype(np.float32)
    mix = src.sum(axis=0)
    if noise > 0:
        mix = mix + (noise * r.standard_normal(mix.shape)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(mix, dtype=np.float32)), torch.from_numpy(src)


def synthetic_multichannel(sources, n_ch=7, max_delay=3, seed=0, sensor_noise=0.01):
    """sources [n_src,B,n] -> array mixture [B,n_ch,n].

    Source k reaches channel c with an integer delay ((k+1)*c) % (max_delay+1), so
    the two sources have different inter-channel phase patterns and the IPD
    varies over time with the local source dominance; a little independent sensor
    noise keeps atan2 away from 0/0.
    """
    n_src, B, n = sources.shape
    out = torch.zeros(B, n_ch, n, dtype=sources.dtype)
    for k in range(n_src):
        for c in range(n_ch):
            d = ((k + 1) * c) % (max_delay + 1)
            out[:, c, d:] += sources[k, :, : n - d]
    if sensor_noise > 0:
        r = np.random.RandomState(7919 * seed + 3)
        out += torch.from_numpy((sensor_noise * r.standard_normal(out.shape)).astype(np.float32))
    return out


def mvdr_inputs(D=7, F=257, T=40, seed=0):
    """Complex multichannel spectrum [D,F,T] (complex64) with one coherent source per bin plus diffuse noise, and three
    soft masks [3,F,T] - shared by tests/golden/make_golden_mvdr.py and the MVDR parity tests."""
    import numpy as np
    rs = np.random.RandomState(seed)
    mix = (rs.standard_normal((D, F, T)) + 1j * rs.standard_normal((D, F, T))).astype(np.complex64)
    steer = np.exp(1j * rs.uniform(0, 2 * np.pi, (D, F, 1))).astype(np.complex64)
    mix = mix * 0.3 + steer * (rs.standard_normal((1, F, T)) + 1j * rs.standard_normal((1, F, T))).astype(np.complex64)
    masks = rs.uniform(0.05, 1.0, (3, F, T)).astype(np.float32)
    return mix, masks


def doa_inputs(sv, B=2, F=257, T=40, seed=3, dirs=((4, 4), (3, 13))):
    """7-channel spectra X [B,7,F,T] complex64 built from plane waves of the steering table `sv` [F,A,7] and two
    complementary soft masks [B,F,T,2] - shared by tests/golden/make_golden_separator.py and the DOA-merge tests.
    Window 0: both masks select the same direction (one output gets silenced); window 1: two directions 120 degrees
    apart (nothing changes)."""
    rs = np.random.RandomState(seed)
    s = torch.from_numpy((rs.standard_normal((2, B, F, T)) + 1j * rs.standard_normal((2, B, F, T))).astype(np.complex64))
    X = torch.zeros((B, 7, F, T), dtype=torch.complex64)
    mask = torch.zeros((B, F, T, 2))
    sel = torch.from_numpy(rs.uniform(0, 1, (B, F, T)) > 0.5)
    for b in range(B):
        for k in range(2):
            X[b] += sv[:, dirs[b][k], :].conj().t()[:, :, None] * s[k, b][None] * (sel[b] if k == 0 else ~sel[b])[None]
        mask[b, ..., 0] = sel[b].float() * 0.9 + 0.05
        mask[b, ..., 1] = (~sel[b]).float() * 0.9 + 0.05
    X += 0.05 * torch.from_numpy((rs.standard_normal(tuple(X.shape)) + 1j * rs.standard_normal(tuple(X.shape))).astype(np.complex64))
    return X, mask


class FakeMaskSession:
    """Stand-in for the onnxruntime session of separator.py:94-103 with a closed-form "model": three mask arrays of
    shape (1, 257, L) computed from the first 257 rows of the (1, F, L) input, some of them above 1 so that the clamp
    of separator.py:104 is exercised.  Shared by tests/golden/make_golden_separator.py (plugged into the REFERENCE
    Separator) and the tests (plugged into css_amd's)."""

    class _In:
        name = "input"

    def get_inputs(self):
        return [self._In()]

    def run(self, output_names, feed):
        f = np.asarray(feed["input"], dtype=np.float32)[0, :257, :]
        return [(np.abs(np.sin(f * (k + 1) + 0.3 * k)) * 1.3).astype(np.float32)[None] for k in range(3)]
