#!/usr/bin/env python3
"""Busy time vs gaps of the kernel timeline in a rocprofv3 rocpd database (last N kernels)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rows = db.execute("select name, start, end from kernels order by start").fetchall()[-last:]
    busy = sum(e - s for _, s, e in rows) / 1e3
    span = (rows[-1][2] - rows[0][1]) / 1e3
    gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"kernels {len(rows)}  span {span:.1f} us  busy {busy:.1f} us  gaps {sum(pos):.1f} us  "
          f"(mean gap {sum(pos) / max(1, len(pos)):.2f} us, max {max(gaps):.1f} us, overlaps {sum(1 for g in gaps if g < 0)})")
    big = sorted(((g, rows[i][0][:60], rows[i + 1][0][:60]) for i, g in enumerate(gaps)), reverse=True)[:8]
    for g, a, b in big:
        print(f"  gap {g:8.1f} us after {a} -> {b}")


if __name__ == "__main__":
    main()
