#!/bin/bash
# Diagnostic: build libcsship variants with parts of mha_fwd_kernel compiled out (-DA2_ABL=<mask>) under build/abl/,
# then (on the GPU box) time each one.  Outputs are wrong by construction: only the timings mean anything.
#   tools/attn_ablate.sh build 0 1 2 4 8 16 ...     (here: cross-compiles)
#   tools/attn_ablate.sh run 0 1 2 4 8 16 ...       (under gpurun)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
mkdir -p build/abl
for m in "$@"; do
  if [ "$mode" = build ]; then
    fl="-DA2_ABL=$m"; [ "$m" = stamps ] && fl="-DA2_STAMPS"; hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $fl -c css_amd/csrc/attn.hip -o build/abl/attn_$m.o
    objs=$(ls css_amd/csrc/*.o | grep -v '/attn.o')
    hipcc -shared -fPIC --offload-arch=gfx950 -o build/abl/lib_$m.so $objs build/abl/attn_$m.o
  else
    echo "== A2_ABL=$m"
    python tools/with_lib.py build/abl/lib_$m.so tools/bench_kernels.py attn | head -1
  fi
done
