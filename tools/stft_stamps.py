"""Debug: per-wave start/end cycle stamps of the STFT kernel (needs a -DSTFT_STAMPS build: tools/stft_stamps.py <lib.so>)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from css_amd import ops
B, n = int(os.environ.get("SB", 32)), int(os.environ.get("SN", 102656))
a = torch.randn(int(os.environ.get("S3", 3)), B, n, device="cuda")
h = ctypes.CDLL(_lib.LIB_PATH)
for e in range(1, 7):
    torch.cuda.synchronize()
    h.csship_debug_stft_epoch(ctypes.c_uint(e))
    for _ in range(3):
        ops.stft(a, ("mag",))
torch.cuda.synchronize()
buf = np.zeros(3 * 8192, dtype=np.uint64)
assert h.csship_debug_stft_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(-1, 3)
wid_all = np.arange(len(st))
wid_all = wid_all[st[:, 1] > 0]
st = st[st[:, 1] > 0]
ep = (st[:, 2] >> 48).astype(np.int64)
print("epochs", np.bincount(ep).tolist())
wid_all = wid_all[ep == ep.max()]
st = st[ep == ep.max()]
xcc = ((st[:, 2] >> 32) & 0xff).astype(np.int64)
hw = (st[:, 2] & 0xffffffff).astype(np.int64)
a0 = st[:, 0].astype(np.int64); a1 = st[:, 1].astype(np.int64)
keep = a1 > a1.max() - 3000  # last launch (10 ns ticks)
print("kept", keep.sum(), "of", len(a0))
a0, a1, xcc = a0[keep], a1[keep], xcc[keep]
wid = wid_all[keep]
base = a0.min(); s0 = (a0 - base) / 100.0; s1 = (a1 - base) / 100.0  # microseconds
print("all: start pct0/10/50/90/100", np.percentile(s0, [0, 10, 50, 90, 100]).round(2).tolist(), " end pct 0/10/50/90/100",
      np.percentile(s1, [0, 10, 50, 90, 100]).round(2).tolist(), " dur pct 10/50/90", np.percentile(s1 - s0, [10, 50, 90]).round(2).tolist())
if len(a0) < 64:
    print('durations us', np.round(s1 - s0, 2).tolist()); sys.exit(0)
print("start hist (1us bins)", np.histogram(s0, bins=np.arange(0, 26))[0].tolist())
print("end hist   (1us bins)", np.histogram(s1, bins=np.arange(0, 26))[0].tolist())
for x in range(8):
    m = xcc == x
    print(f"xcc {x}: n {m.sum()} start min/med/max {s0[m].min():.2f} {np.median(s0[m]):.2f} {s0[m].max():.2f}  end max {s1[m].max():.2f}")
hw = hw[keep]
r1 = s0 < 2.0
cu_key = (xcc << 16) | (hw >> 8 & 0xffff)
simd = hw >> 4 & 3
wv = hw & 15
u, c = np.unique(cu_key[r1], return_counts=True)
print("round 1: distinct CUs", len(u), "waves/CU histogram", np.bincount(c).tolist())
u2, c2 = np.unique(cu_key[r1] * 4 + simd[r1], return_counts=True)
print("round 1: distinct SIMDs", len(u2), "waves/SIMD histogram", np.bincount(c2).tolist())
print("wave slot ids used", np.bincount(wv[r1]).tolist())
print("hw id bit usage: or=%08x" % np.bitwise_or.reduce(hw))

dur = s1 - s0
def grp(name, key):
    ks = np.unique(key)
    print(name, " ".join(f"{k}:{dur[key == k].mean():.1f}" for k in ks[:40]))
grp("by xcc", xcc)
grp("by wave-in-wg", wid & 3)
gx = int(os.environ.get("GX", "10"))
grp("by blockIdx.x", (wid >> 2) % gx)
grp("by signal/8", (wid >> 2) // gx // 8)
cnt = dict(zip(*np.unique(cu_key, return_counts=True)))
grp("by waves on the same CU", np.array([cnt[k] for k in cu_key]))
se = hw >> 13 & 7
grp("by se", se)
grp("by cu id", hw >> 8 & 15)
