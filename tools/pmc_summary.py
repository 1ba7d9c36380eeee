#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc results.db:  python tools/pmc_summary.py results.db [filter]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name, ctr, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if flt in name:
        acc[name][ctr].append(val)
for name, d in acc.items():
    print(name[:110])
    for ctr, v in sorted(d.items()):
        print(f"   {ctr:32s} n={len(v):4d}  avg={sum(v)/len(v):16.1f}")
