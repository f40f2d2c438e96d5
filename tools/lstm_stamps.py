"""Per-step phase timeline of the distributed LSTM forward (library built with -DLSTM_STAMPS in lstm.o):
python tools/lstm_stamps.py build/lib_lstm_stamps.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from css_amd import ops
B, T, H = 32, 400, 512
g = torch.Generator().manual_seed(0)
gx = (torch.randn((B * T, 8 * H), generator=g) * 0.5).cuda()
whh = torch.cat([ops.pack_frag((torch.randn((4 * H, H), generator=g) / H ** 0.5).cuda()) for _ in range(2)]).contiguous()
nwave = 2 * ((B + 15) // 16) * (H // 32) * 2
buf = torch.zeros((nwave, 32, 6), dtype=torch.int32, device="cuda")
h = C.CDLL(_lib.LIB_PATH)
h.csship_debug_lstm_stamps.argtypes = [C.c_void_p]
h.csship_debug_lstm_stamps(buf.data_ptr())
for _ in range(3):
    ops.lstm_bidir(gx, whh, B, T, H)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
d = lambda i, j: ((s[:, :, i] - s[:, :, j]) & 0xFFFFFFFF).astype(np.float64).ravel()
def show(name, v):
    print(f"  {name:40s} median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f}  cycles")
show("step start -> poll starts", d(1, 0))
show("poll (loads until every tag is valid)", d(2, 1))
show("gate math + publish store issued", d(3, 2))
show("output stores issued", d(4, 3))
show("whole step", d(4, 0))
sp = s[:, :, 5].ravel()
print("  poll attempts beyond the first: mean", sp.mean(), "max", sp.max(), " histogram", np.bincount(sp.astype(int))[:8])
