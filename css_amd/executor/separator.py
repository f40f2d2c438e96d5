"""Sliding-window separator -- drop-in for css.executor.separator.Separator (separator.py:9-111) with the
mask model running on the HIP path instead of an ONNX session.

Differences from the reference, all stated:
  * ``backend="hip"``: the mask estimator is a css_amd model (``model=`` argument or
    ``separation_config["model"]``); the reference's ONNX path (third-party runtime, checkpoints not in the
    repo) is out of scope.  Any batch size is accepted (the ONNX restriction to 1 does not apply).
  * ``merge`` defaults to False when the key is missing (conf/config_1ch.yaml has none; the reference raises
    KeyError).  The DOA-based ``angle_merge`` is out of scope and raises if requested.
  * With torch.distributed initialised, windows are sharded contiguously across ranks and gathered
    (long-recording inference): no collective on the per-window data path.
"""
import math

import torch

from .. import ops, parallel
from .feature import FeatureExtractor


class Separator:
    def __init__(self, separation_config, sr=16000, sess_options=None, device="cuda", backend="hip", lowcut=80,
                 highcut=2000, model=None):
        if backend != "hip":
            raise ValueError(f"Unknown backend: {backend}")
        self.separator = model if model is not None else separation_config.get("model")
        if self.separator is None:
            raise ValueError("backend='hip' needs a css_amd model (model=... or separation_config['model'])")
        self.batch_size = separation_config["batch_size"]
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
        if self.merge:
            raise NotImplementedError("DOA-based angle_merge (separator.py:113-250) is out of scope for css_amd")

    def separate(self, s):
        """s: waveform [D, n] -> (masks [W, 257, T, 3] clamped to <= 1, mags [W, 257, T])."""
        s = s.to(self.device)
        win = s.unfold(-1, self.eval_win + 256, self.eval_hop)  # D x W x n_win: a view, read in place (separator.py:67)
        W = win.shape[1]
        rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if parallel.is_distributed() else (0, 1)
        lo, hi = parallel.shard_range(W, rank, world)
        masks, mags = [], []
        for b0 in range(lo, hi, self.batch_size):
            b1 = min(b0 + self.batch_size, hi)
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
        return masks, mags

    def separate_one_batch(self, s):
        """s: [B, D, n] (may be a strided view) -> (masks [B,257,T,3], mag [B,257,T])."""
        if s.shape[1] == 1 and self.ipd is None:
            s = s[:, 0]
        mag_specs, f, re, im = self.feature.forward(s)
        B, Fd, T = f.shape
        m = self.separator.forward_masks(f.transpose(1, 2).contiguous())  # [B, T, 3*257]
        nb = self.feature.num_bins
        masks = m.reshape(B, T, -1, nb).permute(0, 3, 1, 2)[..., :3]  # (B, 257, T, 3): spk1, spk2, noise
        masks = torch.clamp(masks, max=1.0).contiguous()  # separator.py:104
        return masks, mag_specs
