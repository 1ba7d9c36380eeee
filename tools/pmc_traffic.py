#!/usr/bin/env python3
"""HBM traffic of one kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.
usage: pmc_traffic.py <fetch results.db> <write results.db> <kernel substring> <steps in the run> <out.json>
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; both counters
are in KiB."""
import json, sqlite3, sys

def total(db, counter, flt):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    sel = [v for n, v in rows if flt in n]
    return sum(sel), len(sel)

fetch_db, write_db, flt, steps, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
f, nf = total(fetch_db, "FETCH_SIZE", flt)
w, nw = total(write_db, "WRITE_SIZE", flt)
res = {"kernel": flt, "steps": steps, "dispatches_per_step": nf / steps,
       "fetch_bytes_per_step": 2.0 * f * 1024 / steps, "write_bytes_per_step": w * 1024 / steps,
       "hbm_bytes_per_step": (2.0 * f + w) * 1024 / steps,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of bench.py; FETCH_SIZE doubled (gfx950), KiB -> bytes"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
