"""Stand-alone timing of the embedding backward (K3, swr_embed_bwd through the C ABI) over the mid-size tables of config 2
(35 .. 1 472 rows, dim 16, batch 65 536, dE = the compact [B, 144] dX of the step): MFMA segment sums (default) against
the fixed-point direct sums (SWR_K3_MFMA=0).  20 back-to-back calls between two HIP events, 4 rotated inputs.
usage: python tools/micro/k3_probe.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")):
    sys.path.insert(0, p)
import torch

from scenario_wise_rec import _hip as H
from scenario_wise_rec._hip import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
vocabs = [1000, 51, 1472, 35, 119, 455, 200, 300]
dim, ld = 16, 144
g = torch.Generator(device="cuda").manual_seed(0)
sets = []
for j in range(4):
    keys = torch.stack([torch.randint(0, v, (B,), device="cuda", generator=g, dtype=torch.int32) for v in vocabs]).contiguous()
    # K3_SCALE: magnitude of the gradients (the step's dE at batch 65 536 has a median of 4.5e-6 and a maximum of 1e-4)
    dE = torch.randn(B, ld, device="cuda", generator=g) * float(os.environ.get("K3_SCALE", "5e-6"))
    sets.append((keys, dE))
grads = [torch.zeros(v, dim, device="cuda") for v in vocabs]
slots = (H.EmbedGradSlot * len(vocabs))()
for s, v in enumerate(vocabs):
    slots[s] = H.EmbedGradSlot(v, dim, 16 * s, s, 0, grads[s].data_ptr(), None, None)
flag = H.err_flag(torch.device("cuda"))


def bench(label):
    nbytes = lib.swr_embed_bwd_workspace_bytes(slots, len(vocabs), B)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def call(j):
        keys, dE = sets[j % 4]
        H.check(lib.swr_embed_bwd(slots, len(vocabs), H.ptr(keys), H.ptr(dE), ld, B, H.ptr(ws), nbytes, H.ptr(flag), H.stream()), "bwd")
    for j in range(4):
        call(j)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 20
    for j in range(n):
        call(j)
    b.record()
    b.synchronize()
    H.check_errors()
    total = a.elapsed_time(b) / n * 1e3

    def timed(fn):
        for j in range(4):
            fn(j)
        torch.cuda.synchronize()
        a.record()
        for j in range(n):
            fn(j)
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / n * 1e3
    t_sort = timed(lambda j: H.check(lib.swr_embed_bwd_sort(slots, len(vocabs), H.ptr(sets[j % 4][0]), B, H.ptr(ws), nbytes, H.stream()), "sort"))
    t_late = timed(lambda j: H.check(lib.swr_embed_bwd_reduce_part(slots, len(vocabs), H.ptr(sets[j % 4][0]), H.ptr(sets[j % 4][1]), ld, B, 2,
                                                                  H.ptr(ws), nbytes, H.ptr(flag), H.stream()), "part2"))
    print(f"{label}: {total:.1f} us per swr_embed_bwd ({nbytes >> 20} MB workspace); zero-fill + sort half {t_sort:.1f}, "
          f"sums + finalise {t_late:.1f}", flush=True)


for rows in (("64", "128", "256", "512", "1024", "4096") if os.environ.get("K3_PROBE_MFMA") else ()):
    os.environ["SWR_K3_MFMA"] = "1"
    os.environ["SWR_K3_MFMA_MAX_ROWS"] = rows
    for wgs in ("256", "768"):
        os.environ["SWR_K3_MFMA_WGS"] = wgs
        bench(f"mfma up to {rows} rows, wgs {wgs}")
os.environ["SWR_K3_MFMA"] = "0"
bench("fixed-point direct")
