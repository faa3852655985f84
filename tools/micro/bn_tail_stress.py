"""Randomised cross-check of the float4 + tail-workgroup form of swr_affine_act_fwd / swr_bn_act_bwd_stats / swr_act_bwd_apply
(csrc/bn.hip, v4_split) against the thread-per-column form: the same data in a buffer whose row stride is NOT a multiple of four floats
forces the latter.  Y, dZ and the tail columns' partial sums must agree bit for bit (same operations in the same order).  usage: python tools/micro/bn_tail_stress.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from scenario_wise_rec import _hip as H
from scenario_wise_rec._hip import lib

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for case in range(n_cases):
    M = int(rng.choice([1, 3, 31, 64, 65, 257, 1000, 4099]))
    n_relu = 4 * int(rng.integers(1, 80))
    group = int(rng.integers(2, 10))
    n_groups = int(rng.integers(1, max(2, 64 // group)))
    extra = int(rng.choice([0, 0, 4, 8]))                       # an element-wise range behind the softmax groups
    N = n_relu + group * n_groups + extra
    if N % 4:
        extra += 4 - N % 4
        N = n_relu + group * n_groups + extra
    ranges = [(0, n_relu, "relu", 1), (n_relu, n_relu + group * n_groups, "softmax", group)]
    if extra:
        ranges.append((n_relu + group * n_groups, N, "sigmoid", 1))
    acts, n_acts = H.act_ranges(ranges, N)
    g = torch.Generator(device="cuda").manual_seed(case)
    Z0 = torch.randn((M, N), device="cuda", generator=g)
    dY0 = torch.randn((M, N), device="cuda", generator=g)
    sc, sh = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    mean, rstd = torch.randn(N, device="cuda", generator=g), torch.rand(N, device="cuda", generator=g) + 0.5
    ca, cb, cc = (torch.randn(N, device="cuda", generator=g) for _ in range(3))
    outs = []
    for ld in (N, N + 1):                                      # N: float4 + tail; N + 1: thread-per-column
        def buf(src=None):
            t = torch.full((M, ld), float("nan"), device="cuda")[:, :N]
            if src is not None:
                t.copy_(src)
            return t
        Z, dY, Y, dZ = buf(Z0), buf(dY0), buf(), buf()
        nt = (M + 63) // 64
        part = torch.full((nt, N, 2), float("nan"), device="cuda")
        H.check(lib.swr_affine_act_fwd(H.ptr(Z), ld, H.ptr(sc), H.ptr(sh), acts, n_acts, H.ptr(Y), ld, M, N, H.stream()), "fwd")
        H.check(lib.swr_bn_act_bwd_stats(H.ptr(dY), ld, H.ptr(Y), ld, H.ptr(Z), ld, H.ptr(mean), H.ptr(rstd), acts, n_acts, H.ptr(part), M, N,
                                         H.stream()), "stats")
        H.check(lib.swr_act_bwd_apply(H.ptr(dY), ld, H.ptr(Y), ld, H.ptr(Z), ld, H.ptr(ca), H.ptr(cb), H.ptr(cc), H.ptr(mean), acts, n_acts,
                                      H.ptr(dZ), ld, M, N, H.stream()), "apply")
        torch.cuda.synchronize()
        outs.append((Y.contiguous(), part, dZ.contiguous()))
    # Y and dZ: bit for bit everywhere.  The partial sums: bit for bit in the tail columns (the same code in both forms); the float4
    # columns add their rows in another order than the thread-per-column kernel (by design), so those agree to rounding
    (Ya, Pa, Da), (Yb, Pb, Db) = outs
    ok = torch.equal(Ya.view(torch.int32), Yb.view(torch.int32)) and torch.equal(Da.view(torch.int32), Db.view(torch.int32))
    if group == 4:          # aligned groups of one float4 are the float4 kernels' own: no tail (partials to rounding everywhere)
        n_relu = N
    ok = ok and torch.equal(Pa[:, n_relu:].contiguous().view(torch.int32), Pb[:, n_relu:].contiguous().view(torch.int32))
    scale = float(Pb.abs().max()) + 1e-30
    ok = ok and float((Pa - Pb).abs().max()) <= 2e-6 * scale
    outs = [(Ya, Pa[:, n_relu:].contiguous(), Da), (Yb, Pb[:, n_relu:].contiguous(), Db)]
    c0 = ranges[1][0]
    ref = torch.softmax((Z0 * sc + sh)[:, c0:c0 + group].double(), 1)
    ok = ok and float((outs[0][0][:, c0:c0 + group].double() - ref).abs().max()) < 1e-6
    if not ok:
        bad += 1
        msg = []
        for name, a, b in zip(("Y", "partials", "dZ"), *outs):
            ne = a.view(torch.int32) != b.view(torch.int32)
            if bool(ne.any()):
                idx = ne.nonzero()[0].tolist()
                cols = sorted(set(ne.nonzero()[:, 1].tolist()))
                msg.append(f"{name}: {int(ne.sum())} words differ, first at {idx} ({float(a[tuple(idx)])!r} vs {float(b[tuple(idx)])!r}), columns {cols[:6]}..{cols[-1]}")
        print(f"MISMATCH case {case}: M {M} N {N} relu {n_relu} softmax {n_groups} x {group} extra {extra}: " + "; ".join(msg))
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
