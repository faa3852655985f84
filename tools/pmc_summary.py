#!/usr/bin/env python3
"""Sum rocprofv3 PMC counters per kernel (last dispatch of each kernel whose name contains the filter)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
rows = db.execute("select substr(name,1,44), dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events "
                  "where name like ? group by dispatch_id, counter_name order by dispatch_id", (f"%{flt}%",)).fetchall()
d, names, dur = defaultdict(dict), {}, {}
for n, disp, c, v, du in rows:
    d[disp][c] = v
    names[disp] = n
    dur[disp] = du
last = {}
for disp in sorted(d):
    last[names[disp]] = disp
for n, disp in last.items():
    print(n, f"{dur[disp] / 1e3:.0f}us", {k: f"{v:.3g}" for k, v in d[disp].items()})
