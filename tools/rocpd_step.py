#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocpd database: steps are delimited by `embed_gather_kernel` launches;
reports the median over the last N steps of: step span, busy time, and per-kernel-name totals."""
import sqlite3
import statistics
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 200      # longer "steps" are the stand-alone launches of bench.py's roofline leg
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "fl_keys_kernel" in r[0]]       # fused lookup: the keys launch opens a step
    if len(marks) < 3:
        marks = [i for i, r in enumerate(rows) if "embed_gather_kernel" in r[0]]
    steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if 10 <= marks[i + 1] - marks[i] <= kmax][-nlast:]  # skip the
    # stand-alone gather launches of bench.py's roofline leg
    spans, busys, per = [], [], defaultdict(list)
    for a, b in steps:
        ks = rows[a:b]
        spans.append((rows[b][1] - ks[0][1]) / 1e3)
        busys.append(sum(e - s for _, s, e in ks) / 1e3)
        agg = defaultdict(float)
        cnt = defaultdict(int)
        for n, s, e in ks:
            agg[n[:70]] += (e - s) / 1e3
            cnt[n[:70]] += 1
        for n in agg:
            per[n].append((agg[n], cnt[n]))
    print(f"steps {len(steps)}: median span {statistics.median(spans):.1f} us, median busy {statistics.median(busys):.1f} us, "
          f"kernels/step {steps[-1][1] - steps[-1][0]}")
    for n, v in sorted(per.items(), key=lambda kv: -statistics.median(x[0] for x in kv[1])):
        print(f"  {statistics.median(x[0] for x in v):9.1f} us  x{v[-1][1]:<3d} {n}")


if __name__ == "__main__":
    main()
