"""On-GPU synthetic separation batches.

An iterator with the reference dataset's contract (css/datasets/separation.py:163-267: ``__next__`` returns
``{"mix", "source1", "source2": [B,T,F], "lens", "ovl"}``), but mixtures are band-shaped Gaussian sources mixed
as the reference mixes (second source delayed, zero padded, summed) and the features are STFT magnitudes computed
by the HIP front end on the device -- no CPU data loader, no host->device copy per step.
"""
import torch

from .. import ops
from ..utils import synthetic as syn


class SyntheticSeparationDataset:
    @staticmethod
    def add_args(parser):
        parser.add_argument("--batch-size", type=int, default=32)
        parser.add_argument("--window-frames", type=int, default=400)

    @classmethod
    def build_dataset(cls, conf):
        return cls(conf.get("batch_size", 32), conf.get("window_frames", 400), seed=conf.get("seed", 0))

    def __init__(self, batch_size=32, frames=400, seed=0, device="cuda", pool=4):
        self.batch_size, self.frames, self.device = batch_size, frames, device
        self.n_samples = ops.FRAME_LEN + ops.FRAME_HOP * (frames - 1)
        self.seed, self.pool, self._i = seed, pool, 0
        self._audio = []  # a small pool of device-resident mixtures, generated once

    def _ensure_pool(self):
        while len(self._audio) < self.pool:
            mix, src = syn.synthetic_sources(self.batch_size, self.n_samples, seed=self.seed + len(self._audio))
            delay_ratio = float((src[1].abs().sum(-1) > 0).float().mean())
            self._audio.append((torch.stack([mix, src[0], src[1]], 0).to(self.device), delay_ratio))

    def __iter__(self):
        return self

    def __next__(self):
        self._ensure_pool()
        audio, _ = self._audio[self._i % self.pool]
        self._i += 1
        mags = ops.stft(audio, ("mag",))["mag"]  # [3, B, T, 257]
        # overlap ratio: fraction of frames in which both sources are active (logging only, like b["ovl"])
        act = (mags[1:].sum(-1) > 1e-6).float()
        ovl = float((act[0] * act[1]).mean())
        return {"mix": mags[0], "source1": mags[1], "source2": mags[2],
                "lens": torch.full((self.batch_size,), self.frames, dtype=torch.int64), "ovl": ovl}
