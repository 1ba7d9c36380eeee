#!/usr/bin/env python3
"""Kernel dispatches of a rocprofv3 results.db in launch order:  python tools/prof_seq.py results.db [first] [count] [filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 100
flt = sys.argv[4] if len(sys.argv) > 4 else ""
rows = c.execute("select name, start, end, grid_x, workgroup_x, lds_size from kernels order by start").fetchall() \
    if False else None
cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
q = "select name, start, end, grid_x, workgroup_x, lds_size from kernels order by start"
try:
    rows = c.execute(q).fetchall()
except Exception:
    print(cols); raise
t_prev = None
i = 0
for name, s, e, gx, wx, lds in rows:
    if flt and flt not in name: continue
    if i >= first and i < first + count:
        gap = (s - t_prev) / 1e3 if t_prev else 0.0
        print(f"{i:5d} {(e-s)/1e3:8.2f} us  gap {gap:7.2f}  grid {gx//max(wx,1):5d}x{wx:<4d} lds {lds:6d}  {name[:90]}")
    t_prev = e
    i += 1
