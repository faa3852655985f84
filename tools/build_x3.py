#!/usr/bin/env python3
"""Measurement build: libswr with every fp32 product as THREE bf16 products instead of six (-DSWR_X3 on the three sources that hold
bf16-split products) -> scenario_wise_rec/_lib/variants/libswr_x3.so.  Use: SWR_LIB=<that file> python -m pytest ... / bench.py."""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import build_native as B          # noqa: E402

SRCS = ("gemm.hip", "first_layer.hip", "dw_tr.hip")


def main():
    B.build(verbose=False)
    out_dir = os.path.join(B.LIB_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)

    def one(src):
        obj = os.path.join(B.OBJ_DIR, f"variant_x3_{src[:-4]}.o")
        subprocess.run([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DSWR_X3", "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
        return src, obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
        objs_x3 = dict(ex.map(one, SRCS))
    objs = [objs_x3.get(s, os.path.join(B.OBJ_DIR, s[:-4] + ".o")) for s in sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))]
    lib = os.path.join(out_dir, "libswr_x3.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
