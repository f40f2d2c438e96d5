#!/usr/bin/env python3
"""s_waitcnt vmcnt(N) histogram per kernel, counted behind the kernel's first MFMA (i.e. inside its matrix loops).

hipcc merges the pending-load state of the two sides of a branch pessimistically: behind a PREDICATED global load or
store (a run-time `if` around it, an exec-masked bounds test) every later wait is computed for the path with fewer
operations in flight, so a kernel that prefetches N loads ahead ends up waiting for loads it issued a moment ago
(round 5: the chain kernel's W2 passes ended on vmcnt(1) where vmcnt(8) was meant).  A kernel whose loop shows many small
N next to its intended prefetch depth is a candidate.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S css_amd/csrc/panel.hip -o /tmp/panel.s
    python tools/isa_waits.py /tmp/panel.s [name filter]
"""
import collections
import re
import sys


def main():
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur, stats = None, {}
    for line in open(sys.argv[1]):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
            stats[cur] = {"mfma": 0, "loads": 0, "stores": 0, "waits": collections.Counter()}
            continue
        if cur is None:
            continue
        st = stats[cur]
        if "v_mfma" in line:
            st["mfma"] += 1
        if "global_load" in line or "buffer_load" in line:
            st["loads"] += 1
        if "global_store" in line or "buffer_store" in line:
            st["stores"] += 1
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
        if m and st["mfma"] > 0:
            st["waits"][int(m.group(1))] += 1
    for name, st in stats.items():
        if st["mfma"] >= 16 and flt in name:
            print(f"{name[:100]}\n    mfma {st['mfma']} loads {st['loads']} stores {st['stores']}  vmcnt: "
                  + " ".join(f"{k}x{v}" for k, v in sorted(st["waits"].items())))


if __name__ == "__main__":
    main()
