#!/usr/bin/env python3
"""A/B builds of libswr.so: recompile ONE source with extra -D flags and link it with the other objects of the regular
build into scenario_wise_rec/_lib/variants/libswr_<name>.so (git-ignored; travels with gpurun like the main library).

    python tools/build_variant.py <name> <file.hip> -DBM_ROWS=32 [...]

Use on the GPU box:  SWR_LIB=scenario-wise-rec_amd/scenario_wise_rec/_lib/variants/libswr_<name>.so python bench.py ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import build_native as B          # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build(verbose=False)
    out_dir = os.path.join(B.LIB_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(B.OBJ_DIR, f"variant_{name}_{src[:-4]}.o")
    subprocess.run([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
    objs = [obj if s == src else os.path.join(B.OBJ_DIR, s[:-4] + ".o")
            for s in sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))]
    lib = os.path.join(out_dir, f"libswr_{name}.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
