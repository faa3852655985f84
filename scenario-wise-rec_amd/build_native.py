#!/usr/bin/env python3
"""Build libswr.so (the C-ABI library of include/swr.h) for gfx950, in-tree.

    python scenario-wise-rec_amd/build_native.py [--force]

hipcc cross-compiles without a GPU.  One object per .hip source (compiled in
parallel, skipped when up to date), linked into
scenario-wise-rec_amd/scenario_wise_rec/_lib/libswr.so.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(HERE, "build")
LIB_DIR = os.path.join(HERE, "scenario_wise_rec", "_lib")
LIB = os.path.join(LIB_DIR, "libswr.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
         "-Wno-unused-result", "-DNDEBUG"]
# per-source flags.  gemm.hip: no SLP vectorizer -- it packs the fp32 subtractions / column sums of the bf16-split staging code into
# v_pk_add_f32, which share the matrix pipe's side of the SIMD and slow the partner wave's MFMAs (wide weight-gradient kernel:
# 62 -> 57 us with the product + reduce, tools/micro/fl_probe.py)
EXTRA_FLAGS = {"gemm.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "swr.h")]
    jobs = []
    for src in sources:
        obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stderr

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for src, rc, err in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[build_native] hipcc {src}: {'ok' if rc == 0 else 'FAILED'}", flush=True)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n{err}")
    objs = [os.path.join(OBJ_DIR, s[:-4] + ".o") for s in sources]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print(f"[build_native] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
