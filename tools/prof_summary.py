#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total / average duration, share.
usage: python tools/prof_summary.py results.db [steps] > profiles/xxx.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches" + (f" ({tot/1e6/steps:.3f} ms per step over {steps:g} steps incl. warm-up)" if steps else ""))
print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'share':>6}")
for n, c, s, a, mn, mx in rows:
    n = re.sub(r"\s+", " ", n)
    n = n if len(n) <= 88 else n[:85] + "..."
    print(f"{n:<90} {c:>7} {s/1e6:>10.3f} {a/1e3:>9.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {100*s/tot:>5.1f}%")
