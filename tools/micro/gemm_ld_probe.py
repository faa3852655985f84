"""Forward products of HAMUR's hyper-net output layer shape ([32768, K] x [K, N], K small, N wide): where does the time go?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
from scenario_wise_rec import ops

M = 32768
def run(fn, n=30):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, ld, K in ((1225, 1225, 64), (1225, 1248, 64), (2450, 2450, 64), (2450, 2464, 64), (1225, 1225, 32), (1225, 1225, 128),
                 (1225, 1225, 256), (1216, 1216, 64), (148, 148, 276)):
    A = [torch.randn(M, K, device="cuda") for _ in range(4)]
    W = torch.randn(N, K, device="cuda") * 0.1
    b = torch.zeros(N, device="cuda")
    Cs = [torch.empty(M, ld, device="cuda") for _ in range(4)]
    parts = torch.empty(((M + 31) // 32, N, 2), device="cuda")
    t = run(lambda i: ops.gemm("nt", A[i % 4], W, Cs[i % 4], M, N, K, bias=b, stat_partials=parts, ldc=ld))
    t2 = run(lambda i: ops.gemm("nt", A[i % 4], W, Cs[i % 4], M, N, K, bias=b, ldc=ld))
    print(f"N {N} ldc {ld} K {K}: {t:.1f} us with BN partials, {t2:.1f} us without ({4e-6 * M * N / t2:.2f} TB/s of C)")
