"""Opcode-class stream of one kernel from a hipcc -S listing: python tools/isa_stream.py file.s <label substring>
M mfma, G global/buffer load, S global store, r/w LDS read/write, | s_waitcnt, B s_barrier, a accvgpr move, v VALU, s SALU, J branch"""
import re, sys
s = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and l.split(":")[0].endswith("E") or (l.startswith("_Z") and sys.argv[2] in l and ":" in l))
ops = []
for l in s[start + 1:]:
    t = l.strip()
    if t.startswith(".end_amdhsa_kernel") or t.startswith("s_endpgm"):
        break
    if not t or t[0] in ";." or t.endswith(":"):
        if t.endswith(":") and not t.startswith(";"):
            ops.append("L")
        continue
    ops.append(t.split()[0])
def cls(o):
    if o == "L": return "\n"
    if "mfma" in o: return "M"
    if o.startswith(("buffer_load", "global_load", "flat_load")): return "G"
    if o.startswith(("buffer_store", "global_store", "flat_store")): return "S"
    if o.startswith("ds_read") or o.startswith("ds_load"): return "r"
    if o.startswith("ds_write") or o.startswith("ds_store"): return "w"
    if o.startswith("s_waitcnt"): return "|"
    if o.startswith("s_barrier"): return "B"
    if o.startswith("v_accvgpr"): return "a"
    if o.startswith("v_"): return "v"
    if o.startswith(("s_cbranch", "s_branch")): return "J"
    return "s"
st = "".join(cls(o) for o in ops)
for blk in st.split("\n"):
    for i in range(0, len(blk), 160):
        print(blk[i:i + 160])
    print("--")
