#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results database (rocprofv3 --kernel-trace --stats writes a rocpd
SQLite file on this image).  Usage: python tools/rocpd_stats.py <results.db> [--skip-first N] > summary.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        d = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    total = sum(v[1] for v in stats.values())
    print(f"{'kernel':<90} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, (n, tot, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 88 else name[:85] + "..."
        print(f"{short:<90} {n:>7} {tot:>12.1f} {tot / n:>10.2f} {mn:>10.2f} {mx:>10.2f} {100 * tot / total:>6.2f}")
    print(f"{'TOTAL':<90} {sum(v[0] for v in stats.values()):>7} {total:>12.1f}")


if __name__ == "__main__":
    main()
