"""Oracle: STFT / IPD / MVN features (reference css/executor/feature.py).

Test infrastructure only -- see oracle/__init__.py.
"""
import math

import numpy as np
import torch

EPS = float(np.finfo(np.float32).eps)  # feature.py:9


def hann_periodic(n):
    """torch.hann_window(n) (periodic) -- feature.py:23."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * math.pi * k / n)


def dft_basis(frame_len=512, round_pow_of_two=True):
    """Windowed one-sided DFT basis, feature.py:13-30.

    Returns (cos_w, sin_w), each [N/2+1, frame_len] float32, such that
    re[k] = sum_n x[n] cos_w[k, n] and im[k] = sum_n x[n] sin_w[k, n] with
    sin_w = -sin(2 pi k n / N) * w[n]  (rfft sign convention).
    """
    n_fft = 2 ** math.ceil(math.log2(frame_len)) if round_pow_of_two else frame_len
    n = np.arange(frame_len, dtype=np.float64)[None, :]
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * math.pi * k * n / n_fft
    w = hann_periodic(frame_len)[None, :]
    return (
        torch.from_numpy((np.cos(ang) * w).astype(np.float32)),
        torch.from_numpy((-np.sin(ang) * w).astype(np.float32)),
    )


def stft(x, frame_len=512, frame_hop=256, round_pow_of_two=True):
    """STFT.forward, feature.py:45-82: no centre padding, T=(n-L)/hop+1.

    x: [B, n] or [B, D, n] float32 -> (mag, phase, re, im), each [B,(D,)F,T].
    """
    if x.dim() not in (2, 3):
        raise RuntimeError("expect 2D/3D tensor")
    cw, sw = dft_basis(frame_len, round_pow_of_two)
    frames = x.unfold(-1, frame_len, frame_hop)  # [..., T, L]
    re = torch.matmul(frames, cw.t()).transpose(-1, -2).contiguous()
    im = torch.matmul(frames, sw.t()).transpose(-1, -2).contiguous()
    mag = torch.sqrt(re * re + im * im)
    pha = torch.atan2(im, re)
    return mag, pha, re, im


def parse_ipd_index(s):
    """feature.py:95-101."""
    pairs = [tuple(int(v) for v in p.split(",")) for p in s.split(";")]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def ipd(pha, ipd_index="1,0;2,0;3,0;4,0;5,0;6,0"):
    """IPDFeature.forward, feature.py:105-134.  pha [N,C,F,T] -> [N, P*F, T]."""
    if pha.dim() == 3:
        pha = pha.unsqueeze(0)
    li, ri = parse_ipd_index(ipd_index)
    d = pha[:, li] - pha[:, ri]
    c, s = torch.cos(d), torch.sin(d)
    out = torch.atan2(s - s.mean(-1, keepdim=True), c - c.mean(-1, keepdim=True))
    return out.reshape(out.shape[0], -1, out.shape[-1])


def mvn_time_last(f):
    """feature.py:169-171: clamp, then MVN over the last (time) axis, unbiased std."""
    f = torch.clamp(f, min=EPS)
    return (f - f.mean(-1, keepdim=True)) / (f.std(-1, keepdim=True) + EPS)


def feature_extractor(x, frame_len=512, frame_hop=256, round_pow_of_two=True, ipd_index=None):
    """FeatureExtractor.forward, feature.py:156-175 -> (mag, f, re, im)."""
    mag, pha, re, im = stft(x, frame_len, frame_hop, round_pow_of_two)
    if mag.dim() == 4:
        mag = mag[:, 0]
    f = mvn_time_last(mag)
    if ipd_index:
        f = torch.cat([f, ipd(pha, ipd_index)], dim=1)
    return mag, f, re, im


def log_power(x, eps=1e-10, frame_len=512, frame_hop=256):
    """Build-defined log-power features log(max(|X|^2, eps)) on the reference's STFT (feature.py:45-82); the reference's
    own training features come from Lhotse (third-party, unpinned: SURVEY.md 8c).  -> [B,(D,)257,T]."""
    _, _, re, im = stft(x, frame_len, frame_hop)
    return torch.log(torch.clamp(re * re + im * im, min=eps))


def istft_center(re, im, n_fft=512, hop=256, length=None):
    """torch.istft(center=True, hann, onesided) as used at beamformer.py:83-102.

    Own overlap-add restatement: irfft of every frame, synthesis window, OLA,
    divide by the summed squared-window envelope, drop n_fft/2 centre padding.
    re/im: [B, F, T] -> [B, hop*(T-1)] (or `length`).
    """
    B, F, T = re.shape
    spec = torch.complex(re, im).transpose(1, 2)  # [B,T,F]
    frames = torch.fft.irfft(spec, n=n_fft, dim=-1)  # [B,T,n_fft]
    w = torch.from_numpy(hann_periodic(n_fft).astype(np.float32))
    frames = frames * w
    total = n_fft + hop * (T - 1)
    y = torch.zeros(B, total)
    env = torch.zeros(total)
    for t in range(T):
        y[:, t * hop : t * hop + n_fft] += frames[:, t]
        env[t * hop : t * hop + n_fft] += w * w
    start = n_fft // 2
    end = total - n_fft // 2 if length is None else start + length
    return y[:, start:end] / env[start:end]
