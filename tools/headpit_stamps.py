"""Per-wave phase timeline of one csship_head_pit launch.  Needs a library whose headpit.o was built with -DHP_STAMPS:
  cd css_amd/csrc && mkdir -p probe && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DHP_STAMPS -c headpit.hip -o probe/headpit_st.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o probe/libhp.so $(ls *.o | grep -v '^headpit.o$') probe/headpit_st.o
python tools/headpit_stamps.py css_amd/csrc/probe/libhp.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ["CSSHIP_HEADPIT_ONE_ROLE"] = "1"   # the stamped kernel is the 8-wave one
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from css_amd import ops
B, T, F = 32, 400, 257
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn(B, T, 256, device=dev)
w = torch.randn(3 * F, 256, device=dev) * 0.05
b = torch.randn(3 * F, device=dev) * 0.1
f = torch.rand(B, T, F, device=dev)
tg = [torch.rand(B, T, F, device=dev) for _ in range(2)]
wp, bp = ops.pack_head_pit(w, b, 2, F)
for _ in range(3):
    ops.head_pit(x, wp, bp, f, tg)
torch.cuda.synchronize()
h = C.CDLL(_lib.LIB_PATH)
nwg = (B * T + 63) // 64
buf = torch.zeros((nwg, 8, 16), dtype=torch.int32, device=dev)
h.csship_debug_headpit_stamps.argtypes = [C.c_void_p]
assert h.csship_debug_headpit_stamps(buf.data_ptr()) == 0
# queue two launches back to back so that the stamped one starts behind work, as in the step
ops.head_pit(x, wp, bp, f, tg)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
names = ["x rows -> LDS, f / target loads requested", "barrier", "pass 0 MFMAs", "pass 0 squared errors", "pass 1 MFMAs",
         "pass 1 squared errors", "tail tile (wave 0)", "reduction + partial store"]
print(f"{nwg} workgroups; cycles of the stamp clock, median over workgroups; waves 0 / 3 / 7")
for k, nme in enumerate(names):
    d = (s[:, :, k + 1] - s[:, :, k]) & 0xFFFFFFFF
    print(f"  {nme:48s} " + " ".join(f"{np.median(d[:, w]):7.0f}" for w in (0, 3, 7)))
whole = (s[:, :, 8] - s[:, :, 0]) & 0xFFFFFFFF
print(f"  whole workgroup (wave 0): median {np.median(whole[:, 0]):.0f}  p90 {np.percentile(whole[:, 0], 90):.0f}")
t0 = s[:, 0, 0].min()
print(f"  first stamp spread over workgroups: {(s[:, 0, 0] - t0).max()} cycles; last end - first start: {(s[:, 0, 8].max() - t0)} cycles")
