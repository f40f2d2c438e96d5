"""Spectral / spatial features -- drop-in for css.executor.feature (feature.py:33-175).

Outputs keep the reference's [B, F, T] layout.  Internally the HIP kernels work frame-major
([B, T, F], what the models consume): ``stft_tf`` exposes that layout so the training hot
path never pays for the transpose.
"""
import torch

from .. import ops

EPSILON = ops.F32_EPS


def _check_geometry(frame_len, frame_hop, round_pow_of_two):
    if frame_len != ops.FRAME_LEN or frame_hop != ops.FRAME_HOP:
        raise NotImplementedError("css_amd builds the 512/256 STFT of conf/config_*.yaml only")


class STFT:
    """STFT.forward (feature.py:45-82): returns (mag, phase, re, im) as B x (D x) F x T."""

    def __init__(self, frame_len=512, frame_hop=256, round_pow_of_two=True):
        _check_geometry(frame_len, frame_hop, round_pow_of_two)
        self.stride = frame_hop
        self.num_bins = frame_len // 2 + 1

    def forward_tf(self, x, want=("mag",)):
        return ops.stft(x, want)

    def forward(self, x):
        o = ops.stft(x, ("mag", "pha", "re", "im"))
        res = []
        for k in ("mag", "pha", "re", "im"):
            t = o[k]
            lead = t.shape[:-2]
            res.append(ops.transpose_norm(t.reshape(-1, t.shape[-2], t.shape[-1])).reshape(*lead, t.shape[-1], t.shape[-2]))
        return tuple(res)


class FeatureExtractor(torch.nn.Module):
    """FeatureExtractor.forward (feature.py:156-175) -> (mag [B,F,T], f [B,F*(1+P),T], re, im)."""

    def __init__(self, frame_len=512, frame_hop=256, round_pow_of_two=True, ipd_index=None):
        super().__init__()
        self.forward_stft = STFT(frame_len, frame_hop, round_pow_of_two)
        self.num_bins = self.feature_dim = self.forward_stft.num_bins
        self.ipd_index = ipd_index
        if ipd_index:
            pairs = [tuple(map(int, p.split(","))) for p in ipd_index.split(";")]
            self.index_l = [p[0] for p in pairs]
            self.index_r = [p[1] for p in pairs]
            self.feature_dim = self.num_bins * (1 + len(pairs))

    def forward_tf(self, x, want_spectra=True):
        """Frame-major features, the layout the mask estimators consume (no transposes anywhere):
        (mag [B,T,257] of channel 0, f [B,T,257*(1+P)], re, im [B,(C,)T,257] or None).  Same values as ``forward``."""
        multi = x.dim() == 3
        if multi and (self.ipd_index or want_spectra):
            o = ops.stft(x, ("re", "im"))                      # all channels: the spectra feed IPD / MVDR / DOA
            re_tf, im_tf = o["re"], o["im"]
            mag0 = ops.stft(x[:, 0], ("mag",))["mag"]          # only channel 0's magnitude is a feature
        else:
            x0 = x[:, 0] if multi else x
            want = ("mag", "re", "im") if want_spectra else ("mag",)
            o = ops.stft(x0, want)
            mag0, re_tf, im_tf = o["mag"], o.get("re"), o.get("im")
        B, T, F = mag0.shape
        mean, rinv = ops.mvn_stats(mag0, clamp_min=EPSILON)   # [B, F] each
        f = torch.empty(B, T, self.feature_dim, device=x.device, dtype=torch.float32)
        fm = f[:, :, :F]
        torch.clamp(mag0, min=EPSILON, out=fm)
        fm.sub_(mean.view(B, 1, F)).mul_(rinv.view(B, 1, F))
        if self.ipd_index:
            if not multi:
                raise RuntimeError("IPD features need a multi-channel signal")
            ops.ipd_tf(re_tf, im_tf, self.index_l, self.index_r, out=f, f_off=F)
        return mag0, f, re_tf, im_tf

    def forward(self, x):
        o = ops.stft(x, ("mag", "re", "im"))
        mag_tf, re_tf, im_tf = o["mag"], o["re"], o["im"]
        multi = mag_tf.dim() == 4
        mag0 = mag_tf[:, 0] if multi else mag_tf  # "just pick first channel" (feature.py:166-168)
        B, T, F = mag0.shape
        mean, rinv = ops.mvn_stats(mag0, clamp_min=EPSILON)
        f = torch.empty(B, self.feature_dim, T, device=x.device, dtype=torch.float32)
        ops.transpose_norm(mag0, mean, rinv, clamp_min=EPSILON, out=f, f_off=0)
        if self.ipd_index:
            if not multi:
                raise RuntimeError("IPD features need a multi-channel signal")
            ops.ipd(re_tf, im_tf, self.index_l, self.index_r, out=f, f_off=F)
        mag = ops.transpose_norm(mag0)
        lead = re_tf.shape[:-2]
        re = ops.transpose_norm(re_tf.reshape(-1, T, F)).reshape(*lead, F, T)
        im = ops.transpose_norm(im_tf.reshape(-1, T, F)).reshape(*lead, F, T)
        return mag, f, re, im
