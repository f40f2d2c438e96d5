"""Summarise a rocprofv3 sqlite result: per-kernel time per step.  Usage: python tools/prof_db.py <results.db> <steps>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d "
                      f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
n = sum(r[1] for r in rows)
print(f"total {tot / steps:.1f} us/step, {n / steps:.0f} launches/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r[2] / steps:10.1f} us/step  n/step={r[1] / steps:6.1f} avg={r[3]:8.1f}  {r[0][:100]}")
