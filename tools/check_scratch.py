#!/usr/bin/env python3
"""Register-spill check of every kernel of libswr: compiles each csrc/*.hip with -Rpass-analysis=kernel-resource-usage (device
code only, no object kept) and lists the kernels whose ScratchSize is not zero -- a spilled kernel reads and writes HBM behind
the programmer's back (bnmix_bwd lost 40 % to 108 bytes per lane before it was found).

    python tools/check_scratch.py            # exit status 1 if any kernel spills (tests/test_abi_cpu.py runs this)
    python tools/check_scratch.py --all      # every kernel with its VGPR / AGPR / scratch numbers
"""
import concurrent.futures
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import build_native as B          # noqa: E402

# kernels allowed to use scratch (name substring -> reason); empty: none
ALLOWED = {}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def analyse(src):
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
                                                               os.path.join(B.CSRC, src), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-2000:]}")
    kernels, cur = [], None
    for line in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"file": src, "name": m.group(1)}
            kernels.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return kernels


def main():
    sources = sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        kernels = [k for ks in ex.map(analyse, sources) for k in ks]
    names = demangle([k["name"] for k in kernels])
    bad = []
    for k in kernels:
        k["pretty"] = names[k["name"]].split("(")[0]
        if k.get("scratch", 0) > 0 and not any(a in k["pretty"] for a in ALLOWED):
            bad.append(k)
    if "--all" in sys.argv:
        for k in sorted(kernels, key=lambda k: (k["file"], k["pretty"])):
            print(f"{k['file']:<18} {k['pretty'][:80]:<80} vgpr {k.get('vgpr', 0):>3} agpr {k.get('agpr', 0):>3} scratch {k.get('scratch', 0):>4}")
    print(f"{len(kernels)} kernels in {len(sources)} files, {len(bad)} with scratch")
    for k in bad:
        print(f"  SPILL {k['file']}: {k['pretty']}  scratch {k['scratch']} B/lane at {k.get('vgpr', 0)} VGPRs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
