#!/usr/bin/env python3
"""Matrix-pipe / wave-state fractions per kernel from two rocprofv3 PMC passes (rocpd databases):
    pmc_mfma_summary.py <sq.db> <grbm.db>
sq.db   : SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
          SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS (summed over the chip, mean over the launches of a kernel)
grbm.db : GRBM_GUI_ACTIVE (cycles the launch kept the chip busy)
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the fraction of matrix-pipe cycles in use.  Both
counters arrive summed over the 8 XCDs (each XCD has its own GRBM: the sum is 8 x the launch's cycles -- checked against
the kernel-trace durations); SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16
(MI355X_MICROARCH.md).  kernel_us = GRBM_GUI_ACTIVE / 8 / 2.1 GHz.  The wave-state columns are fractions of
SQ_WAVE_CYCLES (quad-cycles a resident wave spent: waiting at s_waitcnt / barriers, stalled at issue, issuing)."""
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events "
                      "group by dispatch_id, counter_name").fetchall()
    per = defaultdict(lambda: defaultdict(list))
    for name, _d, c, v in rows:
        per[name.split("(")[0]][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}, \
           {k: max(len(v) for v in cs.values()) for k, cs in per.items()}


def main():
    sq, n = per_kernel(sys.argv[1])
    gui, _ = per_kernel(sys.argv[2])
    if len(sys.argv) > 4 and sys.argv[3] == "--json":
        import datetime
        import json
        import socket
        res = {}
        for k, c in sq.items():
            g = gui.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0)
            if g < 2000:
                continue
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            res[k] = {"launches": n[k], "gui_cycles": g, "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 128.0),
                      "wait": c.get("SQ_WAIT_ANY", 0) / wc, "valu": c.get("SQ_ACTIVE_INST_VALU", 0) / wc}
        json.dump({"source": "rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE (separate passes) over bench.py; mfma_busy = "
                             "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)",
                   "collected": {"utc": datetime.datetime.utcnow().isoformat(timespec="seconds"), "host": socket.gethostname()},
                   "kernels": res}, open(sys.argv[4], "w"), indent=1)
    print(f"{'kernel':<56} {'launches':>8} {'gui_cycles':>11} {'mfma_busy':>9} {'wait':>6} {'stall':>6} {'issue':>6} {'valu':>6} {'lds':>6}")
    order = sorted(sq, key=lambda k: -gui.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0))
    for k in order:
        c = sq[k]
        g = gui.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 128.0) if g else 0.0
        if g < 2000:
            continue
        print(f"{k[:56]:<56} {n[k]:>8} {g:>11.0f} {mfma:>9.3f} {c.get('SQ_WAIT_ANY', 0) / wc:>6.2f} "
              f"{c.get('SQ_WAIT_INST_ANY', 0) / wc:>6.2f} {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:>6.2f} "
              f"{c.get('SQ_ACTIVE_INST_VALU', 0) / wc:>6.2f} {c.get('SQ_ACTIVE_INST_LDS', 0) / wc:>6.2f}")


if __name__ == "__main__":
    main()
