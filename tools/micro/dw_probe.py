"""Weight gradient of the fused first layer at config 2 (swr_fl_dw_bn): the transpose-read kernel (csrc/dw_tr.hip, mode 1) against the
wide register-transposing kernel (gemm_tn_x6w_kernel, mode 0) and against fp64 torch on the written block A'.
Prints max errors and stand-alone times (40 launches between two HIP events, four rotating operand sets).
usage: python tools/micro/dw_probe.py [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")):
    sys.path.insert(0, p)
import torch

import bench
from scenario_wise_rec import _hip as H
from scenario_wise_rec import ops
from scenario_wise_rec._hip import lib
from scenario_wise_rec.trainers import CTRTrainer

cfg = dict(bench.CONFIGS[2])
B = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["batch"]
model, feats = bench.build_model(cfg)
trainer = CTRTrainer(model, "probe", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
model.train()
batches = []
for j in range(4):
    xh, yh = bench.synth_batch(cfg, B, seed=100 + j)
    batches.append({k: torch.from_numpy(v).cuda() for k, v in xh.items()})
infos = []
for x in batches:
    out = model.embedding(x, model.features, squeeze_dim=True, onehot=True)
    infos.append(out._swr_onehot)
torch.cuda.synchronize()
W = ops._cat_params([m.block(0)[0].weight for m in model.experts] + [g.block(0)[0].weight for g in model.gates])
N, K = W.shape
oh = infos[0]
tabs = (H.OnehotTable * len(oh.tables_p))()
for j, (p_t, vocab, dim, off, col) in enumerate(oh.tables_p):
    tabs[j] = H.OnehotTable(p_t.data_ptr(), vocab, dim, off, col)
Wt = torch.empty((oh.n_sel, N), device="cuda")
flag = H.err_flag(torch.device("cuda"))
for f_ in infos:
    f = f_.fl
    f["plan"].N = N
    H.check(lib.swr_fl_keys(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(flag), H.stream()), "keys")
    H.check(lib.swr_fl_prep(C.byref(f["plan"]), H.ptr(W), W.stride(0), K, H.ptr(oh.ohtab), tabs, len(oh.tables_p), H.ptr(oh.sel), oh.n_sel,
                            H.ptr(Wt), N, H.ptr(f["ws"]), H.stream()), "prep")
Np = (N + 31) // 32 * 32
g = torch.Generator(device="cuda").manual_seed(1)
ca, cb, cc, mean = (torch.randn(N, device="cuda", generator=g) * s for s in (1.0, 0.05, 1e-5, 0.3))
dY = [torch.zeros((B, Np), device="cuda") for _ in range(4)]
Z = [torch.zeros((B, Np), device="cuda") for _ in range(4)]
for j in range(4):
    dY[j][:, :N] = torch.randn((B, N), device="cuda", generator=g) * 1e-4
    Z[j][:, :N] = torch.randn((B, N), device="cuda", generator=g)
    dY[j][:, N:] = float("nan")           # the pad columns must never reach a product
    Z[j][:, N:] = float("nan")
Kf = oh.Kp + oh.oh_width
f0 = infos[0].fl
nb = lib.swr_fl_dw_workspace_bytes(C.byref(f0["plan"]))
wsd = torch.empty(nb, dtype=torch.uint8, device="cuda")
print(f"B {B}  N {N}  Kp {oh.Kp}  one-hot {oh.oh_width}  supported {lib.swr_fl_dw_bn_supported(C.byref(f0['plan']), Np, Np)}")


def dw(j, dWp, dbp):
    f = infos[j % 4].fl
    H.check(lib.swr_fl_dw_bn(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(dY[j % 4]), Np, H.ptr(Z[j % 4]), Np, H.ptr(ca), H.ptr(cb), H.ptr(cc),
                             H.ptr(mean), H.ptr(dWp), Kf, H.ptr(dbp), H.ptr(wsd), nb, H.stream()), "dw_bn")


def timed(fn, n=40):
    for j in range(4):
        fn(j)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for j in range(n):
        fn(j)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


if os.environ.get("DW_TIME_ONLY"):
    lib.swr_dw_tr_mode(1)
    sw_, sb_ = torch.empty((N, Kf), device="cuda"), torch.empty(N, device="cuda")
    print(f"{os.environ.get('SWR_LIB', 'base').split('/')[-1]}: swr_fl_dw_bn (+ reduce) {timed(lambda j: dw(j, sw_, sb_), 100):7.1f} us")
    sys.exit(0)
res = {}
for mode in (0, 1):
    lib.swr_dw_tr_mode(mode)
    outs = []
    for rep in range(2):
        dWp = torch.full((N, Kf), float("nan"), device="cuda")
        dbp = torch.full((N,), float("nan"), device="cuda")
        dw(1, dWp, dbp)
        torch.cuda.synchronize()
        outs.append((dWp, dbp))
    print(f"mode {mode}: deterministic {torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])}, "
          f"finite {bool(torch.isfinite(outs[0][0]).all())} / {bool(torch.isfinite(outs[0][1]).all())}")
    res[mode] = outs[0]
    scratch_w, scratch_b = torch.empty((N, Kf), device="cuda"), torch.empty(N, device="cuda")
    print(f"mode {mode}: swr_fl_dw_bn (+ reduce) {timed(lambda j: dw(j, scratch_w, scratch_b)):7.1f} us")

# fp64 reference on the written block of batch 1
A = infos[1].materialize()[:, oh.col0:oh.col0 + Kf].double()
dZ = (ca.double() * dY[1][:, :N].double() + cb.double() * (Z[1][:, :N].double() - mean.double()) + cc.double())
ref_w = dZ.t() @ A
ref_b = dZ.sum(0)
sw, sb = float(ref_w.abs().max()), float(ref_b.abs().max())
for mode in (0, 1):
    ew = float((res[mode][0].double() - ref_w).abs().max()) / sw
    eb = float((res[mode][1].double() - ref_b).abs().max()) / sb
    print(f"mode {mode}: max |dWp - fp64| / max|dWp| = {ew:.2e}   colsum {eb:.2e}")
d = (res[0][0] - res[1][0]).abs()
print(f"mode 1 vs mode 0: max diff / max = {float(d.max()) / sw:.2e}; worst at {divmod(int(d.argmax()), Kf)}")
if float(d.max()) / sw > 1e-4:
    bad = (d > 1e-4 * sw)
    rows, cols = bad.nonzero(as_tuple=True)
    print("bad rows (n):", sorted(set((rows // 32).tolist())), "x32; bad col tiles:", sorted(set((cols // 32).tolist())))
H.check_errors()
