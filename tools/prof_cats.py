#!/usr/bin/env python3
"""Per-step kernel time by kernel family from a rocprofv3 results.db:  python tools/prof_cats.py results.db steps"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2])
cat = collections.Counter(); cnt = collections.Counter()
for n, s, e in db.execute("select name, start, end from kernels"):
    m = re.search(r"(?:sx::|_ZN2sx\d+)([A-Za-z0-9_]+?)(?:_kernel|I|E|<|\()", n)
    k = m.group(1) if m else n[:40]
    k = re.sub(r"_kernel$", "", k)
    cat[k] += e - s; cnt[k] += 1
tot = sum(cat.values())
print(f"total {tot/1e3/steps:.1f} us/step")
for k, v in cat.most_common(30): print(f"{k:36s} {v/1e3/steps:8.1f} us  {cnt[k]/steps:6.1f}/step")
