#!/bin/bash
# Ablation builds of the fused attention backward (diagnostic only): which part of attn_bwd_fused_kernel costs what.
# Runs on the GPU box: tools/attn_fb_ablate.sh   (rebuilds css_amd/csrc/attnfb.o with -DFB_ABL=<bits>, times, restores)
set -u
cd css_amd/csrc
for abl in 0 1 2 4 8 16 6 31; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DFB_ABL=$abl -c attnfb.hip -o attnfb.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o libcsship.so *.o
  echo "FB_ABL=$abl: $(cd ../.. && timeout -k 10 120 python tools/bench_attn_bwd.py 2>/dev/null | grep 'fused key')"
done
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c attnfb.hip -o attnfb.o && hipcc -shared -fPIC --offload-arch=gfx950 -o libcsship.so *.o
