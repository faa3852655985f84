"""Times swr_tower_fwd_linear / swr_tower_bwd alone at the headline shape (B = 65 536, 5 towers, 32 -> 16 -> 1) with HIP
events, rotating over four copies of the batch-sized operands (past the Infinity Cache), and prints checksums.

    python tools/micro/tower_time.py [B]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scenario-wise-rec_amd"))
from scenario_wise_rec import _hip as H  # noqa: E402

lib = H.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
G, K, Hd = 5, 32, 16
N = G * Hd
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = 4
X = [torch.randn(B, G * K, device=dev) for _ in range(R)]
Z1 = [torch.empty(B, N, device=dev) for _ in range(R)]
dZ1 = [torch.empty(B, N, device=dev) for _ in range(R)]
dX = [torch.empty(B, G * K, device=dev) for _ in range(R)]
dV = [torch.randn(B, G, device=dev) for _ in range(R)]
W1 = torch.randn(G, Hd, K, device=dev) * 0.2
b1 = torch.randn(G, Hd, device=dev) * 0.1
part = torch.empty((B + 31) // 32, N, 2, device=dev)
scale, shift, mean, rstd, gamma, w2 = (torch.rand(N, device=dev) + 0.5 for _ in range(6))
shift = shift - 1.0
ca, cb, cc, dgamma, dbeta, dw2 = (torch.zeros(N, device=dev) for _ in range(6))
db2 = torch.zeros(G, device=dev)
ws_bytes = lib.swr_tower_bwd_workspace_bytes(B, G, Hd)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)


def args(i):
    a = H.TowerArgs()
    a.M, a.G, a.K, a.H, a.accumulate = B, G, K, Hd, 0
    a.X, a.ldx = X[i].data_ptr(), G * K
    a.W1, a.b1 = W1.data_ptr(), b1.data_ptr()
    a.Z1, a.ldz = Z1[i].data_ptr(), N
    a.stat_partials = part.data_ptr()
    a.scale, a.shift, a.mean, a.rstd, a.gamma = (t.data_ptr() for t in (scale, shift, mean, rstd, gamma))
    a.w2 = w2.data_ptr()
    a.dV, a.lddv = dV[i].data_ptr(), G
    a.ca, a.cb, a.cc = ca.data_ptr(), cb.data_ptr(), cc.data_ptr()
    a.dgamma, a.dbeta, a.dw2, a.db2 = dgamma.data_ptr(), dbeta.data_ptr(), dw2.data_ptr(), db2.data_ptr()
    a.dZ1, a.lddz = dZ1[i].data_ptr(), N
    a.dX, a.lddx = dX[i].data_ptr(), G * K
    return a


A = [args(i) for i in range(R)]
st = H.stream()


def fwd(i):
    return lib.swr_tower_fwd_linear(C.byref(A[i]), st)


def bwd(i):
    return lib.swr_tower_bwd(C.byref(A[i]), ws.data_ptr(), ws_bytes, st)


for name, fn, mb in (("tower_fwd_linear", fwd, (G * K + N) * 4 * B / 1e6),
                     ("tower_bwd (stats + finalize + apply)", bwd, (2 * N + N + G * K + 2 * G) * 4 * B / 1e6)):
    for i in range(R):
        H.check(fn(i), name)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n):
        fn(i % R)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name}: {us:.1f} us per launch, {mb:.0f} MB algorithmic -> {mb / us:.2f} TB/s")
ref = torch.einsum("bgk,ghk->bgh", X[0].view(B, G, K).double(), W1.double()) + b1.double()
print("fwd max |err| vs fp64:", float((Z1[0].view(B, G, Hd).double() - ref).abs().max()))
print("checksum", float(Z1[0].double().abs().sum()), float(part.double().abs().sum()), float(dZ1[0].double().abs().sum()),
      float(dX[0].double().abs().sum()), float(dw2.double().abs().sum()))
