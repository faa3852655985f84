#!/usr/bin/env python3
"""Time the tower-head kernels (csrc/tower.hip) at the config-2 shapes (graph-captured launches, HIP events)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
sys.path.insert(0, ROOT)
import torch
from scenario_wise_rec import _hip as H
from scenario_wise_rec._hip import lib
from bench import time_kernel_events

M, G, K, Hd = 65536, 5, 32, 16
N = G * Hd
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(M, G * K, device=dev, generator=g)
W1 = torch.randn(G, Hd, K, device=dev, generator=g) * 0.1
b1 = torch.randn(N, device=dev, generator=g)
Z1 = torch.empty(M, N, device=dev)
part = torch.empty((M + 31) // 32, N, 2, device=dev)
scale = torch.rand(N, device=dev) + 0.5
shift = torch.randn(N, device=dev) * 0.1
mean = torch.randn(N, device=dev) * 0.1
rstd = torch.rand(N, device=dev) + 0.5
gamma = torch.rand(N, device=dev) + 0.5
w2 = torch.randn(N, device=dev)
b2 = torch.randn(G, device=dev)
V = torch.empty(M, G, device=dev)
dV = torch.randn(M, G, device=dev)
ca, cb, cc, dgamma, dbeta, dw2 = (torch.zeros(N, device=dev) for _ in range(6))
db2 = torch.zeros(G, device=dev)
dZ1 = torch.empty(M, N, device=dev)
dX = torch.empty(M, G * K, device=dev)
a = H.TowerArgs()
a.M, a.G, a.K, a.H, a.accumulate = M, G, K, Hd, 0
a.X, a.ldx, a.W1, a.b1 = X.data_ptr(), G * K, W1.data_ptr(), b1.data_ptr()
a.Z1, a.ldz, a.stat_partials = Z1.data_ptr(), N, part.data_ptr()
a.scale, a.shift, a.mean, a.rstd, a.gamma = (t.data_ptr() for t in (scale, shift, mean, rstd, gamma))
a.w2, a.b2, a.V, a.ldv, a.dV, a.lddv = w2.data_ptr(), b2.data_ptr(), V.data_ptr(), G, dV.data_ptr(), G
a.ca, a.cb, a.cc = ca.data_ptr(), cb.data_ptr(), cc.data_ptr()
a.dgamma, a.dbeta, a.dw2, a.db2 = dgamma.data_ptr(), dbeta.data_ptr(), dw2.data_ptr(), db2.data_ptr()
a.dZ1, a.lddz, a.dX, a.lddx = dZ1.data_ptr(), N, dX.data_ptr(), G * K
nb = lib.swr_tower_bwd_workspace_bytes(M, G, Hd)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()


def t(name, fn, mb):
    ms = time_kernel_events(fn, 20, st, reps=10)
    print(f"{name:34s} {ms * 1e3:7.1f} us   {mb / ms / 1e3:6.2f} TB/s algorithmic")


t("fwd_linear (with stats)", lambda: H.check(lib.swr_tower_fwd_linear(C.byref(a), H.stream()), "x"), (M * G * K + M * N) * 4 / 1e6)
a.stat_partials = None
t("fwd_linear (no stats)", lambda: H.check(lib.swr_tower_fwd_linear(C.byref(a), H.stream()), "x"), (M * G * K + M * N) * 4 / 1e6)
a.stat_partials = part.data_ptr()
t("fwd_head", lambda: H.check(lib.swr_tower_fwd_head(C.byref(a), H.stream()), "x"), (M * N + M * G) * 4 / 1e6)
t("bwd (stats + finalize + apply)", lambda: H.check(lib.swr_tower_bwd(C.byref(a), H.ptr(ws), nb, H.stream()), "x"),
  (3 * M * N + M * G * K) * 4 / 1e6)
a.dX = None
t("bwd without dX", lambda: H.check(lib.swr_tower_bwd(C.byref(a), H.ptr(ws), nb, H.stream()), "x"), (3 * M * N) * 4 / 1e6)
