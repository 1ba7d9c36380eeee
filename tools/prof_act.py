#!/usr/bin/env python3
"""Per-launch durations of the act_* kernels in the last step of a bench.py kernel trace: python tools/prof_act.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
# average over the last 10 steps, by position in the step
n = idx[-1] - idx[-2]
out = []
for pos in range(n):
    name = rows[idx[-2] + 1 + pos][0]
    if 'act_' not in name: continue
    d = [rows[idx[-k - 1] + 1 + pos][2] - rows[idx[-k - 1] + 1 + pos][1] for k in range(1, 11)]
    gx, wx = rows[idx[-2] + 1 + pos][3:5]
    kind = 'fwd' if 'fwd' in name else ('reduce' if 'reduce' in name else 'apply')
    out.append(f"{pos:3d} {sum(d)/len(d)/1e3:7.2f} us grid {gx//max(wx,1):4d} {kind:6s} {'M0' if name.startswith('_Z') else 'M1'}")
print("\n".join(out))
