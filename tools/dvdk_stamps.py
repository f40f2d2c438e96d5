"""Per-wave phase timeline of one csship_attn_dvdk launch.  Needs a library whose attnbwd.o was built with -DDV_STAMPS:
  cd css_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DDV_STAMPS -c attnbwd.hip -o probe/attnbwd_st.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o probe/libst.so $(ls *.o | grep -v '^attnbwd.o$') probe/attnbwd_st.o
python tools/dvdk_stamps.py css_amd/csrc/probe/libst.so [B H T]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from css_amd import ops_bwd as ob
B, H, T = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (32, 4, 400)
ldp = (T + 7) // 8 * 8
dev = "cuda:0"
p = torch.rand(B, H, T, ldp, device=dev).to(torch.bfloat16)
ds = (torch.rand(B, H, T, ldp, device=dev) - 0.5).to(torch.bfloat16)
qkv = torch.randn(B, T, 3 * H * 64, device=dev).to(torch.bfloat16)
dao = torch.randn(B * T, H * 64, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ob.attn_dvdk(p, ds, dao, qkv, dqkv, H)
torch.cuda.synchronize()
h = C.CDLL(_lib.LIB_PATH)
nwg = 4096
buf = torch.zeros((nwg, 8, 16, 6), dtype=torch.int32, device=dev)
h.csship_debug_dvdk_stamps.argtypes = [C.c_void_p]
assert h.csship_debug_dvdk_stamps(buf.data_ptr()) == 0
ob.attn_dvdk(p, ds, dao, qkv, dqkv, H)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
used = s[:, 0, 0, 0] != 0
s = s[used]
print(f"{s.shape[0]} workgroups stamped; cycles of the stamp clock, median over workgroups; waves 0 / 3 / 6 / 7")
names = ["issue loads", "fragments + MFMAs", "wait block j+1 (vmcnt)", "LDS stores (lgkmcnt 0)", "barrier"]
nsteps = min(16, (T + 31) // 32)
for j in range(nsteps):
    row = []
    for k in range(5):
        d = (s[:, :, j, k + 1] - s[:, :, j, k]) & 0xFFFFFFFF
        row.append(" ".join(f"{np.median(d[:, w]):6.0f}" for w in (0, 3, 6, 7)))
    tot = (s[:, 0, j, 5] - s[:, 0, j, 0]) & 0xFFFFFFFF
    print(f"step {j:2d} [{np.median(tot):6.0f}]: " + " | ".join(f"{n}: {r}" for n, r in zip(names, row)))
whole = (s[:, 0, nsteps - 1, 5] - s[:, 0, 0, 0]) & 0xFFFFFFFF
print(f"loop, wave 0: median {np.median(whole):.0f}, p90 {np.percentile(whole, 90):.0f} stamp cycles")
