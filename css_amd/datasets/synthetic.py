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


class OnDeviceMixtureDataset:
    """The reference's on-the-fly mixing (css/datasets/separation.py:163-267) with every sample touched on the GPU only.

    A pool of synthetic "utterances" (band-shaped Gaussian speech stand-ins of random lengths, grouped by speaker) and a
    noise pool live in HBM.  Per batch the HOST draws what the reference draws with ``random`` - two different speakers,
    one utterance each (the longer one first, :186-188), the second one's start in U(0, dur1 / 2) (:191), the window
    length, and per window the augmentation coins (impulse response p = 0.5, noise p = 0.5 at SNR U(5, 20) dB, the
    constructor defaults of separation.py:76-101) - a few dozen integers, uploaded in one small copy; ``csship_mix_batch``
    builds mixture / source1 / source2 windows and the HIP STFT turns them into features (magnitude or log-power).
    Windows of one mixture are consecutive (`cut_into_windows`, :212) and the last, shorter one is dropped (:214).
    Yields the reference's batch dict ``{"mix", "source1", "source2", "lens", "ovl"}``."""

    @staticmethod
    def add_args(parser):
        parser.add_argument("--batch-size", type=int, default=32)
        parser.add_argument("--window-frames", type=int, default=400)
        parser.add_argument("--features", choices=("mag", "logpow"), default="mag")

    @classmethod
    def build_dataset(cls, conf):
        return cls(conf.get("batch_size", 32), conf.get("window_frames", 400), seed=conf.get("seed", 0),
                   features=conf.get("features", "mag"))

    def __init__(self, batch_size=32, frames=400, seed=0, device="cuda", features="mag", n_speakers=8, utts_per_speaker=2,
                 max_utt_frames=1400, rir_prob=0.5, noise_prob=0.5, snr=(5.0, 20.0)):
        import numpy as np
        self.batch_size, self.frames, self.device, self.features = batch_size, frames, device, features
        self.n = ops.FRAME_LEN + ops.FRAME_HOP * (frames - 1)
        self.rng = np.random.RandomState(seed)
        self.rir_prob, self.noise_prob, self.snr = rir_prob, noise_prob, snr
        U = n_speakers * utts_per_speaker
        L = ops.FRAME_LEN + ops.FRAME_HOP * (max_utt_frames - 1)
        # utterances: shaped noise with a slowly varying envelope, lengths 1.0 .. 1.0 x the pool row
        _, src = syn.synthetic_sources(U, L, n_src=1, seed=seed)
        env = np.repeat(self.rng.uniform(0.2, 1.0, size=(U, L // 4000 + 1)), 4000, axis=1)[:, :L].astype(np.float32)
        self.pool = (src[0] * torch.from_numpy(env)).contiguous().to(device)
        self.lens = self.rng.randint(int(0.6 * L), L + 1, size=U)
        self.speaker = np.arange(U) // utts_per_speaker
        noise, _ = syn.synthetic_sources(4, L, seed=seed + 7)
        self.noise = (0.3 * noise).contiguous().to(device)
        self.last_params = None

    def draw(self):
        """The batch's chunk records (host): what separation.py:176-237 draws, window by window."""
        rng, n = self.rng, self.n
        recs, total_len, total_ovl = [], 0.0, 0.0
        while len(recs) < self.batch_size:
            s1, s2 = rng.choice(self.speaker.max() + 1, size=2, replace=False)
            u1 = int(rng.choice(np_where(self.speaker == s1)))
            u2 = int(rng.choice(np_where(self.speaker == s2)))
            if self.lens[u1] < self.lens[u2]:
                u1, u2 = u2, u1
            l1, l2 = int(self.lens[u1]), int(self.lens[u2])
            delay = int(rng.uniform(0, l1 / 2))
            end = max(l1, delay + l2)
            total_len += end
            total_ovl += min(l2, l1 - delay)
            for k in range(end // n):  # the last, shorter window is dropped
                r = dict(u1=u1, u2=u2, w0=k * n, delay=delay, len1=l1, len2=l2, rir_on=0, d1=0, d2=0, h0=1.0, h1=0.0,
                         h2=0.0, noise_on=0, nu=0, n0=0, snr_db=0.0)
                if rng.uniform() < self.rir_prob:
                    r.update(rir_on=1, d1=int(rng.randint(40, 400)), d2=int(rng.randint(400, 2400)), h0=1.0,
                             h1=float(rng.uniform(0.2, 0.6)), h2=float(rng.uniform(0.05, 0.3)))
                if rng.uniform() < self.noise_prob:
                    r.update(noise_on=1, nu=int(rng.randint(self.noise.shape[0])),
                             n0=int(rng.randint(0, self.noise.shape[1] - n + 1)), snr_db=float(rng.uniform(*self.snr)))
                recs.append(r)
        return recs[:self.batch_size], total_ovl / total_len

    def __iter__(self):
        return self

    def __next__(self):
        recs, ovl = self.draw()
        self.last_params = recs
        params = ops.mix_params(recs).to(self.device, non_blocking=True)
        mix, s1, s2 = ops.mix_batch(self.pool, params, self.n, noise_pool=self.noise)
        audio = torch.stack([mix, s1, s2], 0)
        key = "logpow" if self.features == "logpow" else "mag"
        feats = ops.stft(audio, (key,))[key]
        return {"mix": feats[0], "source1": feats[1], "source2": feats[2],
                "lens": torch.full((self.batch_size,), self.frames, dtype=torch.int64), "ovl": ovl}


def np_where(mask):
    import numpy as np
    return np.nonzero(mask)[0]
