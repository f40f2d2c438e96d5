"""Statistics of the counter-based dropout masks (oracle/dropout.py = css_amd/csrc/common.hpp), CPU only.

Round 6 changed the mask function from "one lowbias32 per PAIR of elements" to "one lowbias32 + one cheap derived word
per FOUR elements" (the two quarter-rate 32-bit multiplies per pair were 0.25 ms of the 2.2 ms train-mode forward).
This prints, for both functions and several (seed, site) keys: the keep rate against 1 - round(p 2^16) / 2^16, the
worst serial correlation of the keep bits over lags 1..64 and the row strides the kernels use, the variance of row /
column keep rates of the masks reshaped to [rows, 1024 | 408 | 256] against the binomial value, a chi-square of the
fields' top byte, the correlation between sites, and the correlations between the four fields of a group.
    python tools/dropout_stats.py [--log2n 24]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dropout as od

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=24)
a = ap.parse_args()
N, P = 1 << a.log2n, 0.1
U = np.uint64


def keep_old(seed, stream, n, p):
    """rounds 2-5: element e kept iff field (e & 1) of hash32(key + (e >> 1)) >= thresh"""
    thresh = int(p * 65536.0 + 0.5)
    key = int(od.hash32(U((int(seed) ^ ((int(stream) * 0x9E3779B9) & 0xffffffff)) & 0xffffffff)))
    e = np.arange(n, dtype=np.uint64)
    h = od.hash32((U(key) + (e >> U(1))) & U(0xffffffff))
    return np.where((e & U(1)) == 1, h >> U(16), h & U(0xffff)) >= thresh


def report(k, name):
    k = k.astype(np.float64)
    n, p = len(k), k.mean()
    sd2 = p * (1 - p)
    kc = k - p
    worst, wl = 0.0, 0
    for lag in list(range(1, 65)) + [128, 256, 257, 400, 408, 512, 771, 1024, 2048, 4096, 65536, 1 << 20]:
        z = (kc[:-lag] * kc[lag:]).mean() / sd2 * np.sqrt(n - lag)
        if abs(z) > worst:
            worst, wl = abs(z), lag
    shape = []
    for w in (1024, 408, 256):
        m = k[:n // w * w].reshape(-1, w)
        shape.append(f"{w}: rows {m.mean(1).var() / (sd2 / w):.3f} cols {m.mean(0).var() / (sd2 / m.shape[0]):.3f}")
    print(f"{name}: keep {p:.5f} (nominal {1 - int(P * 65536 + 0.5) / 65536:.5f}, sigma {np.sqrt(sd2 / n):.1e}); worst serial "
          f"|z| {worst:.2f} at lag {wl}; mean-variance / binomial at width " + "; ".join(shape))


keys = ((12345, 17), (12345, 18), (7, 33), (99, 4113))
for seed, stream in keys:
    report(keep_old(seed, stream, N, P), f"old (seed {seed}, site {stream})")
    report(od.keep_mask(seed, stream, N, P), f"new (seed {seed}, site {stream})")
m = [od.keep_mask(s, t, N, P).astype(np.float64) for s, t in keys[:3]]
print("correlation between sites / seeds:", [round(float(np.corrcoef(m[0], m[i])[0, 1]), 5) for i in (1, 2)])
f = od.keep_mask(12345, 17, N, P).reshape(-1, 4)
for i in range(4):
    for j in range(i + 1, 4):
        c = np.corrcoef(f[:, i], f[:, j])[0, 1]
        print(f"fields {i},{j} of a group: corr {c:+.2e} (z = {c * np.sqrt(len(f)):+.1f})")
