"""Forward product of HAMUR's hyper-net output layer ([32768, 64] x [64, 1225]): does the row pitch of C matter?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
from scenario_wise_rec import ops

M, K = 32768, 64
A = [torch.randn(M, K, device="cuda") for _ in range(4)]
def run(fn, n=30):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, ld in ((1225, 1225), (1225, 1228), (1225, 1248), (1225, 1280), (1248, 1248), (1024, 1024)):
    W = torch.randn(N, K, device="cuda") * 0.1
    b = torch.zeros(N, device="cuda")
    Cs = [torch.empty(M, ld, device="cuda") for _ in range(4)]
    parts = torch.empty(((M + 31) // 32, N, 2), device="cuda")
    t = run(lambda i: ops.gemm("nt", A[i % 4], W, Cs[i % 4], M, N, K, bias=b, stat_partials=parts, ldc=ld))
    t2 = run(lambda i: ops.gemm("nt", A[i % 4], W, Cs[i % 4], M, N, K, bias=b, ldc=ld))
    print(f"N {N} ldc {ld}: {t:.1f} us with BN partials, {t2:.1f} us without ({4e-6 * M * N / t2:.2f} TB/s of C)")
