#!/usr/bin/env python3
"""Timeline of ONE step from a rocpd database (start / end / duration in us relative to the step's first kernel).
Steps are delimited by `fl_keys_kernel` (or `embed_gather_kernel`) launches; the last one with 10..kmax kernels is printed (the stand-alone
launches of bench.py's roofline leg are not steps).  usage: rocpd_one_step.py <db> [kmax=200]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "fl_keys_kernel" in r[0]]          # fused lookup: the keys launch opens a step
if len(marks) < 3:
    marks = [i for i, r in enumerate(rows) if "embed_gather_kernel" in r[0]]
steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if 10 <= marks[i + 1] - marks[i] <= kmax]
a, b = steps[-2] if len(steps) > 1 else steps[-1]
t0 = rows[a][1]
for n, s, e in rows[a:b + 1]:
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  {n[:90]}")
