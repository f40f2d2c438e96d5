#!/bin/bash
# Ablation builds of the MXFP8 FeedForward (csship_ffn_mx, d_model 512 / ffn 2048, 12 800 rows): where do its ~50 us go when
# its scaled MFMAs are 13.7 us of matrix-pipe time?    Diagnostic only; results are wrong by design.
# Runs on the GPU box: tools/ffn_mx_ablate.sh
set -u
cd css_amd/csrc
for abl in 0 1 2 4 8 16 32 3 35 39 63; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMX_ABL=$abl -c ffnw.hip -o ffnw.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o libcsship.so *.o
  echo "MX_ABL=$abl: $(cd ../.. && timeout -k 10 200 python tools/bench_large.py --blocks 1 2>/dev/null | grep -E '^ffn_mxfp8|^ffn_wide')" | tr '\n' ' '; echo
done
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c ffnw.hip -o ffnw.o && hipcc -shared -fPIC --offload-arch=gfx950 -o libcsship.so *.o
