"""Stand-alone times of swr_rowmat_fwd / _bwd at HAMUR's config-5 shape ([32 768, 8, 35] x [35, 35] per sample).
usage: SWR_ROWMAT_PIPE=0|1 python tools/micro/rowmat_probe.py"""
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
out, dT, dH = torch.empty_like(T[0]), torch.empty_like(T[0]), torch.empty_like(Hm[0])


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


fwd = lambda j: H.check(lib.swr_rowmat_fwd(H.ptr(T[j % 3]), H.ptr(Hm[j % 3]), H.ptr(out), B, D, k, H.stream()), "fwd")
bwd = lambda j, acc=0: H.check(lib.swr_rowmat_bwd(H.ptr(dO[j % 3]), H.ptr(T[j % 3]), H.ptr(Hm[j % 3]), H.ptr(dT), H.ptr(dH), acc, B, D, k, H.stream()), "bwd")
print(f"pipe={os.environ.get('SWR_ROWMAT_PIPE', '1')}: fwd {timed(fwd):6.1f} us (234 MB)   bwd {timed(bwd):6.1f} us (431 MB)   "
      f"bwd accumulating {timed(lambda j: bwd(j, 1)):6.1f} us (592 MB)")
