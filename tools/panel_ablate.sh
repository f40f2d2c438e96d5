#!/bin/bash
# Diagnostic: libcsship variants with parts of the row-panel k-loop compiled out (-DP_ABL=<mask>), under build/abl/.
#   tools/panel_ablate.sh build 0 1 2 ...   (here)      tools/panel_ablate.sh run 0 1 2 ... (under gpurun)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
mkdir -p build/abl
for m in "$@"; do
  if [ "$mode" = build ]; then
    fl="-DP_ABL=$m"; [ "$m" = stamps ] && fl="-DP_STAMPS"; hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $fl -c css_amd/csrc/panel.hip -o build/abl/panel_$m.o
    hipcc -shared -fPIC --offload-arch=gfx950 -o build/abl/libp_$m.so $(ls css_amd/csrc/*.o | grep -v '/panel.o') build/abl/panel_$m.o
  else
    echo "== P_ABL=$m"
    python tools/with_lib.py build/abl/libp_$m.so tools/bench_kernels.py chain
  fi
done
