"""Oracle: training mixtures as css/datasets/separation.py:163-267 builds them (numpy, per chunk).

The Lhotse calls of the reference (cut.pad, cut.mix(offset_other_by), cut_into_windows, ReverbWithImpulseResponse,
CutMix) are third-party and unpinned: the SEMANTICS restated here are the ones visible in the reference's
own code - second utterance delayed, both zero padded to the mixture length, summed, windowed; augmentations on the
mixture only; clean padded sources as targets - with a sparse 3-tap impulse response and additive noise at a given SNR
as the stand-ins for the two Lhotse transforms.  Parity unpinned for those two; the mixing itself is pinned by
construction (it is index arithmetic).  Test infrastructure only.
"""
import numpy as np


def _src(pool, u, length, t):
    t = np.asarray(t)
    ok = (t >= 0) & (t < length)
    return np.where(ok, pool[u][np.clip(t, 0, pool.shape[1] - 1)], 0.0).astype(np.float32)


def mix_chunk(pool, p, n, noise_pool=None):
    """p: dict with the fields of css_amd.ops.MIX_FIELDS -> (mix, source1, source2) float32 [n]."""
    t = p["w0"] + np.arange(n, dtype=np.int64)
    s1 = _src(pool, p["u1"], p["len1"], t)
    s2 = _src(pool, p["u2"], p["len2"], t - p["delay"])

    def clean(tt):
        return _src(pool, p["u1"], p["len1"], tt) + _src(pool, p["u2"], p["len2"], tt - p["delay"])
    m = s1 + s2
    if p["rir_on"]:
        m = (np.float32(p["h0"]) * m + np.float32(p["h1"]) * clean(t - p["d1"]) + np.float32(p["h2"]) * clean(t - p["d2"])).astype(np.float32)
    if p["noise_on"]:
        z = noise_pool[p["nu"], p["n0"]:p["n0"] + n].astype(np.float32)
        em, en = float(np.sum(m.astype(np.float64) ** 2)), float(np.sum(z.astype(np.float64) ** 2))
        g = np.sqrt(em / (en * 10.0 ** (p["snr_db"] / 10.0))) if en > 0 else 0.0
        m = (m + np.float32(g) * z).astype(np.float32)
    return m, s1, s2
