"""Row products over random short-batch shapes ("nt" and "nn", grouped and not, bias, BatchNorm partials in the epilogue, ragged last
column tiles): prints one checksum line per case.  The grid-fill heuristic of launch_rows (SWR_GEMM_FILL, csrc/gemm.hip) cuts the
columns into more groups on short batches and must not change a bit:
    SWR_GEMM_FILL=0 python tools/micro/gemm_fill_stress.py > a; SWR_GEMM_FILL=1 python tools/micro/gemm_fill_stress.py > b; cmp a b
Every case is also checked against torch in fp64 (2e-5 of the largest entry)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from scenario_wise_rec import ops

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
worst = 0.0
for case in range(n_cases):
    M = int(rng.choice([1, 31, 128, 250, 1000, 4096, 8192]))
    N = int(rng.choice([4, 20, 96, 148, 160, 200, 588, 768, 1225]))
    K = int(rng.choice([4, 52, 96, 128, 324, 588]))
    G = int(rng.choice([1, 1, 1, 3, 8]))
    kind = str(rng.choice(["nt", "nn"]))
    if G > 1:
        N = min(N, 160)
    g = torch.Generator(device="cuda").manual_seed(case)
    A = torch.randn((M, G * K), device="cuda", generator=g)
    Bm = torch.randn((G * N, K) if kind == "nt" else (G * K, N), device="cuda", generator=g) * 0.2
    bias = torch.randn(G * N, device="cuda", generator=g) if rng.random() < 0.5 else None
    Cm = torch.full((M, G * N), float("nan"), device="cuda")
    stats = G == 1 and rng.random() < 0.5
    part = torch.full(((M + 31) // 32, N, 2), float("nan"), device="cuda") if stats else None
    ops.gemm(kind, A, Bm, Cm, M, N, K, bias=bias, stat_partials=part, groups=G, gsA=K if G > 1 else 0, gsB=N * K if G > 1 else 0,
             gsC=N if G > 1 else 0, gsBias=N if G > 1 else 0)
    torch.cuda.synchronize()
    ref = torch.empty((M, G * N), dtype=torch.float64, device="cuda")
    for q in range(G):
        a = A[:, q * K:(q + 1) * K].double()
        b = (Bm[q * N:(q + 1) * N].double().t() if kind == "nt" else Bm[q * K:(q + 1) * K].double())
        ref[:, q * N:(q + 1) * N] = a @ b + (bias[q * N:(q + 1) * N].double() if bias is not None else 0.0)
    err = float((Cm.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    worst = max(worst, err)
    cs = int(Cm.view(torch.int32).to(torch.int64).sum()) + (int(part.view(torch.int32).to(torch.int64).sum()) if stats else 0)
    print(f"case {case}: {kind} M {M} N {N} K {K} groups {G} bias {bias is not None} stats {stats}: checksum {cs}" + ("  ERROR %.2e" % err if err > 2e-5 else ""))
print(f"worst relative error {worst:.2e}", file=sys.stderr)
