"""swr_rowmat_bwd's two halves timed apart (dT only / dHm only / dHm accumulating) at HAMUR's config-5 shape.
usage: SWR_ROWMAT_MFMA=0|1 python tools/micro/rowmat_parts.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")):
    sys.path.insert(0, p)
import torch

from scenario_wise_rec import _hip as H
from scenario_wise_rec._hip import lib

B, D, k = 32768, 8, 35
g = torch.Generator(device="cuda").manual_seed(1)
T = [torch.randn(B, D, k, device="cuda", generator=g) for _ in range(3)]
Hm = [torch.randn(B, k, k, device="cuda", generator=g) for _ in range(3)]
dO = [torch.randn(B, D, k, device="cuda", generator=g) for _ in range(3)]
dT, dH = torch.empty_like(T[0]), torch.empty_like(Hm[0])


def timed(fn, n=30):
    for j in range(3):
        fn(j)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for j in range(n):
        fn(j)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


def bwd(j, want_dt, want_dh, acc):
    H.check(lib.swr_rowmat_bwd(H.ptr(dO[j % 3]), H.ptr(T[j % 3]), H.ptr(Hm[j % 3]), H.ptr(dT) if want_dt else None,
                               H.ptr(dH) if want_dh else None, acc, B, D, k, H.stream()), "bwd")


print(f"mfma={os.environ.get('SWR_ROWMAT_MFMA', '1')}: dT only {timed(lambda j: bwd(j, True, False, 0)):6.1f} us   "
      f"dHm only {timed(lambda j: bwd(j, False, True, 0)):6.1f} us   dHm accumulating {timed(lambda j: bwd(j, False, True, 1)):6.1f} us")
