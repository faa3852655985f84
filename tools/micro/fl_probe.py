"""Stand-alone timing of the fused lookup + first layer launches (csrc/first_layer.hip) at config 2: swr_fl_keys, swr_fl_prep, swr_fl_fwd, swr_fl_dw.
Each: 40 back-to-back launches between two HIP events, four rotating batches.  usage: python tools/micro/fl_probe.py"""
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

cfg = bench.CONFIGS[2]
B = cfg["batch"]
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
f0 = infos[0].fl
assert f0 is not None
flag = H.err_flag(torch.device("cuda"))


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


def keys(j):
    f = infos[j % 4].fl
    H.check(lib.swr_fl_keys(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(flag), H.stream()), "keys")


print(f"swr_fl_keys  {timed(keys):7.1f} us")

# the layer's parameters: the stacked expert + gate weights
W = ops._cat_params([m.block(0)[0].weight for m in model.experts] + [g.block(0)[0].weight for g in model.gates])
bias = ops._cat_params([m.block(0)[0].bias for m in model.experts] + [g.block(0)[0].bias for g in model.gates])
N, K = W.shape
oh = infos[0]
tabs = (H.OnehotTable * len(oh.tables_p))()
for j, (p_t, vocab, dim, off, col) in enumerate(oh.tables_p):
    tabs[j] = H.OnehotTable(p_t.data_ptr(), vocab, dim, off, col)
Wt = torch.empty((oh.n_sel, N), device="cuda")
for i_ in infos:
    i_.fl["plan"].N = N


def prep(j):
    f = infos[j % 4].fl
    H.check(lib.swr_fl_prep(C.byref(f["plan"]), H.ptr(W), W.stride(0), K, H.ptr(oh.ohtab), tabs, len(oh.tables_p), H.ptr(oh.sel), oh.n_sel,
                            H.ptr(Wt), N, H.ptr(f["ws"]), H.stream()), "prep")


LDZ = int(os.environ.get("FL_LDZ", (N + 31) // 32 * 32))         # the step pads the rows of Z to 128 bytes (ops.PAD_ROWS)
Z = [torch.empty((B, LDZ), device="cuda")[:, :N] for _ in range(4)]
part = torch.empty(((B + 31) // 32, N, 2), device="cuda")


def fwd(j):
    f = infos[j % 4].fl
    H.check(lib.swr_fl_fwd(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(bias), H.ptr(Z[j % 4]), LDZ, H.ptr(part), H.stream()), "fwd")


print(f"swr_fl_prep  {timed(prep):7.1f} us")
print(f"swr_fl_fwd   {timed(fwd):7.1f} us")
if os.environ.get("FL_ONLY") == "fwd":
    part.zero_()
    fwd(1)
    torch.cuda.synchronize()
    print(f"checksum Z {int(Z[1].contiguous().view(torch.int32).to(torch.int64).sum())}  partials {int(part.view(torch.int32).to(torch.int64).sum())}")
    print(f"{os.environ.get('SWR_LIB', 'base').split('/')[-1]}: swr_fl_fwd {timed(fwd, 100):7.1f} us")
    sys.exit(0)
dZ = [torch.randn((B, N), device="cuda") * 1e-4 for _ in range(4)]
Kf = oh.Kp + oh.oh_width
dWp = torch.empty((N, Kf), device="cuda")
dbp = torch.empty(N, device="cuda")
nb = lib.swr_fl_dw_workspace_bytes(C.byref(f0["plan"]))
wsd = torch.empty(nb, dtype=torch.uint8, device="cuda")


def dw(j):
    f = infos[j % 4].fl
    H.check(lib.swr_fl_dw(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(dZ[j % 4]), N, H.ptr(dWp), Kf, H.ptr(dbp), H.ptr(wsd), nb, H.stream()), "dw")


print(f"swr_fl_dw (+ reduce) {timed(dw):7.1f} us")

# BatchNorm backward + dX in one pass against the two launches it replaces
ca, cb, cc, mean = (torch.randn(N, device="cuda") * 0.1 for _ in range(4))
dY = [torch.randn((B, N), device="cuda") * 1e-4 for _ in range(4)]
Zs = [torch.randn((B, N), device="cuda") for _ in range(4)]
dZo = torch.empty((B, N), device="cuda")
dsel = torch.empty((B, oh.n_sel), device="cuda")
for j in range(4):
    prep(j)                        # (writes the B3X image of `sel` into each workspace)


def dx_fused(j):
    f = infos[j % 4].fl
    H.check(lib.swr_bn_bwd_dx(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(dY[j % 4]), N, H.ptr(Zs[j % 4]), N, H.ptr(ca), H.ptr(cb), H.ptr(cc),
                              H.ptr(mean), oh.n_sel, H.ptr(dZo), N, H.ptr(dsel), oh.n_sel, H.stream()), "bn_bwd_dx")


acts, n_acts = H.act_ranges(None, N)


def dx_two(j):
    H.check(lib.swr_act_bwd_apply(H.ptr(dY[j % 4]), N, H.ptr(Zs[j % 4]), N, H.ptr(Zs[j % 4]), N, H.ptr(ca), H.ptr(cb), H.ptr(cc), H.ptr(mean),
                                  acts, n_acts, H.ptr(dZo), N, B, N, H.stream()), "act_bwd_apply")
    ops.gemm("nt", dZo, Wt, dsel, B, oh.n_sel, N)


print(f"swr_bn_bwd_dx (BN backward + dX)      {timed(dx_fused):7.1f} us")

# the form the step launches: rows padded to 128 bytes, dZ not written (the weight gradient recomputes it)
Np = (N + 31) // 32 * 32
dYp = [torch.zeros((B, Np), device="cuda").copy_(torch.nn.functional.pad(d, (0, Np - N))) for d in dY]
Zp = [torch.zeros((B, Np), device="cuda").copy_(torch.nn.functional.pad(z, (0, Np - N))) for z in Zs]
nsp = (oh.n_sel + 31) // 32 * 32
dselp = torch.empty((B, nsp), device="cuda")


def dx_step(j):
    f = infos[j % 4].fl
    H.check(lib.swr_bn_bwd_dx(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(dYp[j % 4]), Np, H.ptr(Zp[j % 4]), Np, H.ptr(ca), H.ptr(cb), H.ptr(cc),
                              H.ptr(mean), oh.n_sel, None, Np, H.ptr(dselp), nsp, H.stream()), "bn_bwd_dx")


print(f"swr_bn_bwd_dx, padded rows, no dZ     {timed(dx_step):7.1f} us")
print(f"swr_act_bwd_apply + swr_gemm_nt (dX)  {timed(dx_two):7.1f} us")
H.check_errors()
