"""Per-workgroup phase timeline of one bgemm launch.  Needs a library whose bgemm.o was built with -DBG_STAMPS:
  cd css_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DBG_STAMPS -c bgemm.hip -o /tmp/bgs.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/lib_stamps.so $(ls *.o | grep -v '^bgemm.o$') /tmp/bgs.o
python tools/bgemm_stamps.py build/lib_stamps.so M N K akm bkm af bf [splits]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from css_amd import ops_bwd as ob
M, N, K, akm, bkm, af, bf = [int(x) for x in sys.argv[2:9]]
splits = int(sys.argv[9]) if len(sys.argv) > 9 else 0
dev = "cuda"
a = torch.randn((K, M) if akm else (M, K), device=dev, dtype=torch.float32 if af else torch.bfloat16)
b = torch.randn((K, N) if bkm else (N, K), device=dev, dtype=torch.float32 if bf else torch.bfloat16)
for _ in range(3):
    ob.bgemm(a, b, a_kmajor=bool(akm), b_kmajor=bool(bkm), splits=splits)
torch.cuda.synchronize()
h = C.CDLL(_lib.LIB_PATH)
nwg = ((M + 127) // 128) * ((N + 127) // 128) * max(splits, 1)
buf = torch.zeros((nwg, 96), dtype=torch.int32, device=dev)
h.csship_debug_bgemm_stamps.argtypes = [C.c_void_p]
assert h.csship_debug_bgemm_stamps(buf.data_ptr()) == 0
ob.bgemm(a, b, a_kmajor=bool(akm), b_kmajor=bool(bkm), splits=splits)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
d = lambda i, j: ((s[:, i] - s[:, j]) & 0xFFFFFFFF).astype(np.float64)
print(f"{nwg} workgroups; cycles (median / p90 over workgroups)")
def show(name, v):
    print(f"  {name:34s} {np.median(v):9.0f} {np.percentile(v, 90):9.0f}")
show("entry -> prologue done", d(1, 0))
nk = min(12, (K // max(splits, 1) + 63) // 64)
for kt in range(nk):
    base = 2 + kt * 5
    prev = 1 if kt == 0 else base - 1
    show(f"kt{kt}: issue loads", d(base, prev))
    show(f"kt{kt}: fragments from LDS", d(base + 1, base))
    show(f"kt{kt}: MFMA issue", d(base + 2, base + 1))
    show(f"kt{kt}: wait loads + LDS store", d(base + 3, base + 2))
    show(f"kt{kt}: barrier", d(base + 4, base + 3))
show("loop end -> stores drained", d(63, 62))
show("whole workgroup", d(63, 0))
