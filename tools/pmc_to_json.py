#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; rocpd databases) ->
profiles/<tag>_pmc_hbm.json, read by bench.py for the `roofline.traffic` field.

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are reported in KiB; on gfx950 FETCH_SIZE counts
128-byte requests at 64 bytes, i.e. HALF the bytes read (checked here against the 21 MB input of the tower
affine kernel and the 174 MB the weight-gradient product must read) -> fetched bytes = 2 * FETCH_SIZE KiB;
WRITE_SIZE matched a known byte count (143 MB written by the gather) as is."""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name = ? "
                      "group by dispatch_id", (counter,)).fetchall()
    acc = defaultdict(list)
    for name, _disp, v in rows:
        acc[name.split("(")[0]].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        res[k] = {"fetch_size_kib_raw": fk, "write_size_kib_raw": wk,
                  "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "launches": f.get(k, w.get(k))[1]}
    import datetime
    import socket
    import subprocess
    try:
        gpu = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=20).stdout
        gpu = next((l.split(":", 1)[1].strip() for l in gpu.split("\n") if "Marketing Name" in l and "Instinct" in l), "")
    except Exception:
        gpu = ""
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py",
               "collected": {"utc": datetime.datetime.utcnow().isoformat(timespec="seconds"), "host": socket.gethostname(), "gpu": gpu},
               "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        if v["hbm_bytes_per_launch"] > 5e6:
            print(f"{k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  x{v['launches']}")


if __name__ == "__main__":
    main()
