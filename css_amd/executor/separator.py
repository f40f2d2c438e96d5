"""Sliding-window separator -- drop-in for css.executor.separator.Separator (separator.py:9-111) with the
mask model running on the HIP path instead of an ONNX session.

Differences from the reference, all stated:
  * ``backend="hip"``: the mask estimator is a css_amd model (``model=`` argument or
    ``separation_config["model"]``).  Any batch size is accepted (the ONNX restriction to 1 does not apply).
  * ``backend="onnx"``: the reference's session protocol, line by line (separator.py:94-103: features reshaped to
    ``(1, F, B*T)``, ``session.run``, three mask arrays stacked and reshaped to ``(B, F', T, 3)``), with a
    ``css_amd.utils.model_util.HipSession`` as the session (onnxruntime and the .onnx checkpoints are third-party
    artefacts outside the repository).  ``batch_size`` must be 1 as in the reference (:21-24).
  * ``merge`` defaults to False when the key is missing (conf/config_1ch.yaml has none; the reference raises
    KeyError).  With ``merge: True`` (conf/config_7ch.yaml) the DOA-based ``angle_merge`` of
    separator.py:113-250 runs: the likelihood over 30 steering directions is a HIP kernel (csship_doa_likelihood),
    the per-window decision is a handful of [B, 2, 30] tensor ops.  The reference always silences output 0
    (``masked_magnitude[b]`` indexes the frequency axis); that is reproduced by default,
    ``fix_merge_indexing=True`` silences the output with less masked energy instead.
  * With torch.distributed initialised, windows are sharded contiguously across ranks and gathered
    (long-recording inference): no collective on the per-window data path.
"""
import math

import torch

from .. import ops, parallel
from .._lib import check, lib
from .feature import FeatureExtractor


class Separator:
    def __init__(self, separation_config, sr=16000, sess_options=None, device="cuda", backend="hip", lowcut=80,
                 highcut=2000, model=None, fix_merge_indexing=False, device_batch=256):
        if backend not in ("hip", "onnx"):
            raise ValueError(f"Unknown backend: {backend}")
        self.backend = backend
        self.separator = model if model is not None else separation_config.get("model")
        if self.separator is None:
            raise ValueError("the separator needs a css_amd model (model=... or separation_config['model'])")
        if backend == "onnx":
            assert separation_config["batch_size"] == 1, "ONNX models only support batch size 1"  # separator.py:21-24
            from ..utils.model_util import get_onnx_model
            self.session, _ = get_onnx_model(separation_config.get("model_path"), sess_options, device, model=self.separator)
        self.batch_size = separation_config["batch_size"]
        # Windows are independent through features -> model -> masks -> merge (separator.py:73-111), so the configured
        # batch size (a memory knob of the reference: 16 windows = 2 400 frames, 38 of this chip's 256 row panels) does
        # not bound how many windows one launch sequence takes: a 10-minute 7-channel session (748 windows) separates
        # in 18 ms with 256 windows per pass against 95 ms with 16.
        self.device_batch = max(int(self.batch_size), int(device_batch)) if backend == "hip" else int(self.batch_size)
        self.device = device
        self.eval_win = int(separation_config["eval_win"] * sr)
        self.eval_hop = int(separation_config["eval_hop"] * sr)
        self.sr = sr
        self.frame_length = separation_config["frame_length"]
        self.frame_shift = separation_config["frame_shift"]
        self.ipd = separation_config.get("ipd", None)
        self.feature = FeatureExtractor(frame_len=self.frame_length, frame_hop=self.frame_shift, ipd_index=self.ipd)
        self.merge = separation_config.get("merge", False)
        self.merge_threshold = separation_config.get("merge_threshold", None)
        self.fix_merge_indexing = fix_merge_indexing
        if self.merge:
            sv, self.angles = self.steervec_7ch(nfreqs=257, nvecs=30, samplerate=sr, return_angles=True)
            self.steervec = sv
            self._sv_re = sv.real.contiguous().float().to(device)
            self._sv_im = sv.imag.contiguous().float().to(device)
            freq_step = (sr // 2) / (257 - 1)
            self._lo = int(math.floor(lowcut / freq_step))
            self._hi = int(math.ceil(highcut / freq_step))

    @staticmethod
    def steervec_7ch(nfreqs, *, nvecs, radius=0.0425, sndvelocity=340, samplerate=16000, return_angles=False,
                     reference=0, inverse_shift=False):
        """separator.py:113-163: plane-wave steering vectors of the 7-microphone circular array (constant table)."""
        angles = torch.tensor([2 * math.pi * k / nvecs for k in range(nvecs)])
        distances = radius * torch.stack((torch.zeros(nvecs), torch.cos(angles + math.pi / 6),
                                          torch.cos(angles - math.pi / 6), torch.cos(angles - math.pi / 2),
                                          torch.cos(angles - 5 * math.pi / 6), torch.cos(angles + 5 * math.pi / 6),
                                          torch.cos(angles + math.pi / 2)), dim=1)
        if reference != 0:
            distances -= distances[:, reference, None]
        deltas = distances / sndvelocity * samplerate
        sign = -1j if inverse_shift else 1j
        steervecs = torch.stack([torch.exp(sign * deltas * math.pi * f / (nfreqs - 1)) for f in range(nfreqs)])
        steervecs = steervecs / math.sqrt(7)
        return (steervecs, angles * 360 / (2 * math.pi)) if return_angles else steervecs

    def doa_likelihood(self, re, im, mask, mask_binarize_thres=0.5, compression=0.5):
        """separator.py:206-250 (softmax=False) on the GPU: re / im [B,7,F,T], mask [B,F,T,2] ->
        (likelihood [B,2,30], masked energy [B,2])."""
        B, D, F, T = re.shape
        if D != 7:
            raise RuntimeError("the DOA merge is defined for the 7-channel array (separator.py:113)")
        re, im, mask = re.contiguous(), im.contiguous(), mask.contiguous()
        like = torch.empty((B, 2, 30), dtype=torch.float32, device=re.device)
        en = torch.empty((B, 2), dtype=torch.float32, device=re.device)
        nws = lib().csship_doa_workspace(B, F, self._lo, self._hi, 30)
        ws = torch.empty(nws, dtype=torch.uint8, device=re.device)
        check(lib().csship_doa_likelihood(ops._ptr(re), ops._ptr(im), ops._ptr(mask), ops._ptr(self._sv_re),
                                          ops._ptr(self._sv_im), B, F, T, 30, self._lo, self._hi, mask_binarize_thres,
                                          compression, ops._ptr(like), ops._ptr(en), ops._ptr(ws), nws, ops._stream()))
        return like, en

    def angle_merge(self, re, im, mask, thresh=16, mask_binarize_thres=0.5, compression=0.5):
        """separator.py:165-204: where the two outputs point at the same direction (<= thresh degrees apart) one of
        them is silenced (mask = 1e-12)."""
        like, en = self.doa_likelihood(re, im, mask, mask_binarize_thres, compression)
        doa = self.angles.to(like.device)[torch.argmax(like, dim=-1)]                      # [B,2] degrees
        diff = torch.minimum((doa[:, 0] - doa[:, 1]) % 360, (doa[:, 1] - doa[:, 0]) % 360)
        close = diff <= thresh
        kill = torch.argmin(en, dim=1) if self.fix_merge_indexing else torch.zeros_like(close, dtype=torch.long)
        new_mask = mask.clone()
        sel = torch.zeros((mask.shape[0], 2), dtype=torch.bool, device=mask.device)
        sel[torch.arange(mask.shape[0], device=mask.device), kill] = close
        new_mask.masked_fill_(sel[:, None, None, :], 1e-12)
        return new_mask

    def separate(self, s):
        """s: waveform [D, n] -> (masks [W, 257, T, 3] clamped to <= 1, mags [W, 257, T])."""
        s = s.to(self.device)
        win = s.unfold(-1, self.eval_win + 256, self.eval_hop)  # D x W x n_win: a view, read in place (separator.py:67)
        W = win.shape[1]
        rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if parallel.is_distributed() else (0, 1)
        lo, hi = parallel.shard_range(W, rank, world)
        masks, mags = [], []
        for b0 in range(lo, hi, self.device_batch):
            b1 = min(b0 + self.device_batch, hi)
            bm, bg = self.separate_one_batch(win[:, b0:b1, :].permute(1, 0, 2))
            masks.append(bm)
            mags.append(bg)
        F = self.feature.num_bins
        T = ops.num_frames(self.eval_win + 256)
        masks = torch.cat(masks, 0) if masks else torch.empty(0, F, T, 3, device=self.device)
        mags = torch.cat(mags, 0) if mags else torch.empty(0, F, T, device=self.device)
        if world > 1:
            masks = parallel.gather_windows(masks, W, rank, world)
            mags = parallel.gather_windows(mags, W, rank, world)
        ops.lstm_dist_check()  # BLSTM separators: a timed-out recurrence hand-off must not pass as masks
        return masks, mags

    def _separate_one_batch_session(self, s):
        """separator.py:86-111 through the session interface (the (1, F, B*T) layout of the ONNX models)."""
        import numpy as np
        from ..utils.model_util import to_numpy
        if s.shape[1] == 1 and self.ipd is None:
            s = s[:, 0]
        mag_specs, f, re, im = self.feature.forward(s.contiguous())
        B, F, T = f.shape
        f = f.reshape(1, F, B * T)
        ort_inputs = {self.session.get_inputs()[0].name: to_numpy(f)}
        this_res = self.session.run(None, ort_inputs)
        masks = torch.Tensor(np.stack((this_res[0], this_res[1], this_res[2]), axis=0))
        masks = masks.reshape(3, B, -1, T).permute(1, 2, 3, 0).to(self.device)
        masks = torch.clamp(masks, max=1.0)
        if self.merge:
            masks[..., :2] = self.angle_merge(re, im, masks[..., :2].contiguous(), thresh=self.merge_threshold)
        return masks.contiguous(), mag_specs

    def separate_one_batch(self, s):
        """s: [B, D, n] (may be a strided view) -> (masks [B,257,T,3], mag [B,257,T])."""
        if self.backend == "onnx":
            return self._separate_one_batch_session(s)
        if s.shape[1] == 1 and self.ipd is None:
            s = s[:, 0]
        # frame-major features: what the mask estimator consumes (the reference's [B, F, T] tensors are produced only
        # where this method hands them on)
        mag_tf, f, re_tf, im_tf = self.feature.forward_tf(s, want_spectra=bool(self.merge))
        B, T, Fd = f.shape
        m = self.separator.forward_masks(f)  # [B, T, 3*257]
        mag_specs = ops.transpose_norm(mag_tf)  # [B, 257, T]
        nb = self.feature.num_bins
        masks = m.reshape(B, T, -1, nb).permute(0, 3, 1, 2)[..., :3]  # (B, 257, T, 3): spk1, spk2, noise
        masks = torch.clamp(masks, max=1.0).contiguous()  # separator.py:104
        if self.merge:  # separator.py:106-110
            lead = re_tf.shape[:-2]
            re = ops.transpose_norm(re_tf.reshape(-1, T, re_tf.shape[-1])).reshape(*lead, re_tf.shape[-1], T)
            im = ops.transpose_norm(im_tf.reshape(-1, T, im_tf.shape[-1])).reshape(*lead, im_tf.shape[-1], T)
            masks[..., :2] = self.angle_merge(re, im, masks[..., :2].contiguous(), thresh=self.merge_threshold)
        return masks, mag_specs
