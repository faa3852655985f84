"""Times swr_bnmix_fwd / swr_bnmix_bwd alone at the headline shape (B = 65 536, 4 experts x 32, 5 gates) with HIP events.

    python tools/micro/bnmix_time.py [B]
Rotates over four copies of every operand so the 256 MB Infinity Cache does not serve the reads.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scenario-wise-rec_amd"))
from scenario_wise_rec import _hip as H  # noqa: E402

lib = H.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ne, Hh, D = 4, 32, 5
N = ne * Hh + D * ne
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = 4
Z = [torch.randn(B, N, device=dev) for _ in range(R)]
dP = [torch.randn(B, D * Hh, device=dev) for _ in range(R)]
P = [torch.empty(B, D * Hh, device=dev) for _ in range(R)]
G = [torch.empty(B, D * ne, device=dev) for _ in range(R)]
dY = [torch.empty(B, N, device=dev) for _ in range(R)]
scale = torch.rand(N, device=dev) + 0.5
shift = torch.randn(N, device=dev) * 0.1
mean = torch.randn(N, device=dev) * 0.1
rstd = torch.rand(N, device=dev) + 0.5
tile = lib.swr_bnmix_tile_rows()
part = torch.empty((B + tile - 1) // tile, N, 2, device=dev)


def args(i):
    a = H.BnMixArgs()
    a.M, a.ne, a.H, a.D = B, ne, Hh, D
    a.Z, a.ldz = Z[i].data_ptr(), N
    a.scale, a.shift = scale.data_ptr(), shift.data_ptr()
    a.P, a.ldp = P[i].data_ptr(), D * Hh
    a.dP, a.lddp = dP[i].data_ptr(), D * Hh
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    a.dY, a.lddy = dY[i].data_ptr(), N
    a.bn_partials = part.data_ptr()
    a.G = G[i].data_ptr()
    return a


A = [args(i) for i in range(R)]
st = H.stream()
for name, fn, mb in (("bnmix_fwd", lib.swr_bnmix_fwd, (N + D * Hh + D * ne) * 4 * B / 1e6),
                     ("bnmix_bwd", lib.swr_bnmix_bwd, (2 * N + D * Hh + D * ne) * 4 * B / 1e6)):
    for i in range(R):
        H.check(fn(C.byref(A[i]), st), name)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n):
        fn(C.byref(A[i % R]), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name}: {us:.1f} us per launch, {mb:.0f} MB algorithmic -> {mb / us:.2f} TB/s")
print("checksum", float(dY[0].double().abs().sum()), float(part.double().abs().sum()), float(P[0].double().abs().sum()))
