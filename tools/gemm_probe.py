#!/usr/bin/env python3
"""Run the three f32-MFMA products at the config-2 shapes a few times (for rocprofv3 --pmc / timing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import torch
from scenario_wise_rec import ops
M, N, K = 65536, 148, 516
g = torch.Generator(device="cuda").manual_seed(0)
E = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g)
dZ = torch.randn(M, N, device="cuda", generator=g)
Z = torch.empty(M, N, device="cuda")
dE = torch.empty(M, K, device="cuda")
dW = torch.empty(N, K, device="cuda")
db = torch.empty(N, device="cuda")
part = torch.empty((M + 31) // 32, N, 2, device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, fn in (("fwd nt", lambda: ops.gemm("nt", E, W, Z, M, N, K, stat_partials=part)),
                 ("dX nn", lambda: ops.gemm("nn", dZ, W, dE, M, K, N)),
                 ("dW tn", lambda: ops.gemm_tn(dZ, E, dW, M, N, K, colsum=db))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e6:.1f} us  {2 * M * N * K / dt / 1e12:.1f} TFLOP/s")
