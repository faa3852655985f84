"""Device ops of the hot path: thin launchers over the C ABI (include/swr.h) and the
`torch.autograd.Function`s the model mirror composes.

Every function here launches hand-written HIP kernels on the current HIP stream; torch supplies
device memory and the autograd tape only.  All matrices are fp32 row-major 2-D tensors whose rows may
be strided (`ld = stride(0)`), so column slices of a wider buffer are passed without copies.
"""
import ctypes as C

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip as H
from ._hip import lib


import os


# =========================================================================== raw launchers
def _ld(t):
    return t.stride(0) if t.dim() == 2 and t.shape[0] > 1 else (t.shape[-1] if t.dim() == 2 else 0)


def gemm(kind, A, B, C_out, M, N, K, bias=None, a_scale=None, a_shift=None, a_relu=False, accumulate=False,
         stat_partials=None, groups=1, gsA=0, gsB=0, gsC=0, gsBias=0, gsScale=0, lda=None, ldb=None, ldc=None,
         B_split=None, n_compute=0, a_exact_from=0, c_act=0):
    """kind 'nt': C[m,n] = sum_k A[m,k] B[n,k];  'nn': C[m,n] = sum_k A[m,k] B[k,n].
    B_split: (planes, ld, plane_stride) from split_weights -- B already split into bf16 planes ('nt', one group).
    a_exact_from: columns k >= this of A hold bf16-exact values (0 / 1 of a one-hot block): half the products there."""
    a = H.GemmArgs()
    a.a_exact_from = int(a_exact_from)
    a.c_act = int(c_act)                # 1 ReLU / 2 sigmoid applied as C is stored (layers without BatchNorm)
    if B_split is not None:
        a.B_split, a.ld_split, a.plane_stride = B_split[0].data_ptr(), B_split[1], B_split[2]
    a.M, a.N, a.K = M, N, K
    a.n_compute = int(n_compute)        # columns >= n_compute of C are written as zeros, not computed (0 = all)
    a.A, a.lda = A.data_ptr(), lda if lda is not None else _ld(A)
    a.B, a.ldb = B.data_ptr(), ldb if ldb is not None else _ld(B)
    a.bias = bias.data_ptr() if bias is not None else None
    a.C, a.ldc = C_out.data_ptr(), ldc if ldc is not None else _ld(C_out)
    a.a_scale = a_scale.data_ptr() if a_scale is not None else None
    a.a_shift = a_shift.data_ptr() if a_shift is not None else None
    a.a_relu = int(a_relu)
    a.accumulate = int(accumulate)
    a.stat_partials = stat_partials.data_ptr() if stat_partials is not None else None
    a.groups, a.gsA, a.gsB, a.gsC, a.gsBias, a.gsScale = groups, gsA, gsB, gsC, gsBias, gsScale
    fn = lib.swr_gemm_nt if kind == "nt" else lib.swr_gemm_nn
    H.check(fn(C.byref(a), H.stream()), f"swr_gemm_{kind}")
    if _side["deferred"]:
        _flush_deferred()          # side-stream work parked until the main stream had its next kernel enqueued


def split_weights(W, want_t):
    """Three bf16 planes (x = h + m + l) of W [N, K] for the bf16-split GEMM, and of W^T when `want_t`: ONE small
    launch per step instead of a split of the weight tile in every workgroup of the product (and, for dX, instead of
    the fp32 transpose copy).  Returns ((planes, ld, plane_stride), (planes_t, ld_t, plane_stride_t) or None)."""
    N, K = W.shape
    ld, ld_t = int(lib.swr_split_ld(K)), int(lib.swr_split_ld(N))
    P = torch.empty(3 * N * ld, dtype=torch.bfloat16, device=W.device)
    Pt = torch.empty(3 * K * ld_t, dtype=torch.bfloat16, device=W.device) if want_t else None
    H.check(lib.swr_split_weights(H.ptr(W), W.stride(0), N, K, H.ptr(P), H.ptr(Pt), H.stream()), "swr_split_weights")
    return (P, ld, N * ld), ((Pt, ld_t, K * ld_t) if want_t else None)


def gemm_tn(A, B, C_out, M, K1, K2, colsum=None, accumulate=False, groups=1, gsA=0, gsB=0, gsC=0, gsColsum=0,
            lda=None, ldb=None, ldc=None, C2=None, c2_from=0):
    """C[k1,k2] = sum_m A[m,k1] B[m,k2] (+ colsum[k1] = sum_m A[m,k1]); deterministic split over m.
    `C2` (with `c2_from`): columns k2 >= c2_from go to C2[k1, k2 - c2_from] instead (overwritten)."""
    a = H.GemmTnArgs()
    if C2 is not None:
        a.C2, a.ldc2, a.c2_from = C2.data_ptr(), C2.stride(0), int(c2_from)
    a.M, a.K1, a.K2 = M, K1, K2
    a.A, a.lda = A.data_ptr(), lda if lda is not None else _ld(A)
    a.B, a.ldb = B.data_ptr(), ldb if ldb is not None else _ld(B)
    a.C, a.ldc = C_out.data_ptr(), ldc if ldc is not None else _ld(C_out)
    a.colsum = colsum.data_ptr() if colsum is not None else None
    a.accumulate, a.groups = int(accumulate), groups
    a.gsA, a.gsB, a.gsC, a.gsColsum = gsA, gsB, gsC, gsColsum
    nbytes = lib.swr_gemm_tn_workspace_bytes(C.byref(a))
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=A.device)
    H.check(lib.swr_gemm_tn(C.byref(a), H.ptr(ws), nbytes, H.stream()), "swr_gemm_tn")


def colsum(X, M, N, out=None, accumulate=False):
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    nbytes = lib.swr_colsum_workspace_bytes(M, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
    H.check(lib.swr_colsum(H.ptr(X), _ld(X), M, N, H.ptr(out), int(accumulate), H.ptr(ws), nbytes, H.stream()), "swr_colsum")
    return out


def _cat_params(tensors):
    """One tensor aliasing `tensors` back to back.  Models lay the parameters of fused layers out
    adjacently in the parameter arena (basic/module.py), in which case this is a zero-copy view of the
    arena; otherwise it is a concatenation copy."""
    t0 = tensors[0].detach()
    if len(tensors) == 1:
        return t0
    tail = tuple(t0.shape[1:])
    adjacent = True
    p, store = t0.data_ptr(), t0.untyped_storage().data_ptr()
    for t in tensors:
        if (not t.is_contiguous() or t.data_ptr() != p or t.dtype != t0.dtype or tuple(t.shape[1:]) != tail
                or t.untyped_storage().data_ptr() != store):
            adjacent = False
            break
        p += t.numel() * t.element_size()
    rows = sum(t.shape[0] if t.dim() > 0 else 1 for t in tensors)
    shape = (rows,) + tail
    if adjacent:
        strides, acc = [], 1
        for n in reversed(shape):
            strides.append(acc)
            acc *= n
        return t0.as_strided(shape, tuple(reversed(strides)))
    return torch.cat([t.detach().reshape((-1,) + tail) for t in tensors])


def _grad_alias(params, kind=1):
    """The `.grad` tensors of `params` as ONE tensor when they sit back to back in the gradient arena
    (basic/module.py), else None.  Backward kernels then accumulate straight into the arena instead of
    returning per-parameter gradients for autograd to add one tiny launch at a time."""
    gs = []
    for p in params:
        if not p.is_leaf or not p.requires_grad:       # (derived tensors, e.g. STAR's effective weights: autograd carries those)
            return None
        if not hasattr(p, "_swr_hooked"):              # only parameters of a gradient arena (basic/module.py build_arena): a foreign
            return None                                # leaf keeps autograd's semantics (torch.autograd.grad, tensor hooks, its own .grad)
        g = getattr(p, "grad", None)
        if g is None or g.dtype != torch.float32:
            return None
        gs.append(g)
    if len(gs) == 1:
        return gs[0] if gs[0].is_contiguous() else None
    t0 = gs[0]
    tail = tuple(t0.shape[1:])
    ptr, store = t0.data_ptr(), t0.untyped_storage().data_ptr()
    for g in gs:
        if not g.is_contiguous() or g.data_ptr() != ptr or tuple(g.shape[1:]) != tail or g.untyped_storage().data_ptr() != store:
            return None
        ptr += g.numel() * 4
    rows = sum(g.shape[0] if g.dim() > 0 else 1 for g in gs)
    shape = (rows,) + tail
    strides, acc = [], 1
    for n in reversed(shape):
        strides.append(acc)
        acc *= n
    return t0.as_strided(shape, tuple(reversed(strides)))


def _mark_touched(params):
    for p in params:
        p._swr_touched = True
        p._swr_grad_clean = False


# ------------------------------------------------------------------ work off the critical path (second HIP stream)
# 1. Embedding backward, sort half: grouping the large tables' entries by row needs only the lookup keys, so
#    EmbedGather.forward forks it onto a side stream and the backward joins before its reduction.  A chain of a dozen
#    latency-bound launches (~75 us) disappears behind the dense forward / backward.
# 2. Weight gradients of mid-size products (_fork_dw): nothing in the backward pass reads them, so a dW chain that writes
#    straight into the gradient arena is forked after dX is enqueued and joined when the autograd engine finishes
#    (queue_callback).
# Under hipGraph capture the forks / joins become parallel branches of the graph.
# SWR_SIDE_STREAM: "1" side streams (the large tables' sort behind the forward pass, the first layer's weight gradient beside the
# embedding backward), "0" everything on one stream, unset = by batch size: a replayed multi-stream graph costs the HOST ~8 us per
# node against ~1.3 us for a single-stream one, and the side branches only pay for themselves on big steps -- measured per-GPU
# shards (round 5): config 1 (batch 4 096) 0.240 -> 0.202 ms on one stream, config 3 (16 384) 1.316 -> 1.290, config 4 (8 192) the
# same, configs 5 / 6 (32 768) and config 2 (65 536) 1-5 % faster WITH the branches.  Decided by the lookup that opens a step.
_SIDE_MODE = os.environ.get("SWR_SIDE_STREAM", "auto")
SIDE_MIN_BATCH = int(os.environ.get("SWR_SIDE_MIN_BATCH", 32768))
SIDE_MIN_BATCH_FUSED = int(os.environ.get("SWR_SIDE_MIN_BATCH_FUSED", 4096))
SIDE_STREAM = _SIDE_MODE != "0"
# weight-gradient products of 2e9 .. 2e10 flop run on a stream of their own (_fork_dw) next to the dX -> K3 chain: 0.535 ->
# 0.506 ms at config 2; a fork / join pair costs ~10-20 us of edges, so smaller products stay on the main stream (forking
# every product LOSES 0.02 ms at configs 1, 3, 4) and chip-filling ones only slow what they overlap
SIDE_DW_MIN_FLOP, SIDE_DW_MAX_FLOP = float(os.environ.get("SWR_SIDE_DW_MIN_FLOP", 5e8)), 2e10      # (5e8: the first layer of config 2 from 4 096 rows on)
PAD_ROWS = os.environ.get("SWR_PAD_ROWS", "1") != "0"          # 128-byte aligned rows for the tensors of a fused gate-mix level
FUSE_BN_DX = os.environ.get("SWR_FUSE_BN_DX", "1") != "0"      # BatchNorm backward applied inside the first layer's dX product
FUSE_BN_DX_SINGLE = os.environ.get("SWR_FUSE_BN_DX_SINGLE", "1") != "0"   # ... also where the step runs on ONE stream (short batches)
_side = {"streams": {}, "keep": [], "queued": False, "deferred": [], "pending": 0,
         "jobs": [],        # one-shot callables that ride the next forward-time fork (the trainer's zero_grad)
         "wt": {},          # (ptr, N, K) -> {"src": W view, "buf": W^T, "epoch": fork that refreshed it}
         "epoch": 0, "extras_ev": None}


# ---- stream-skew harness (tests/test_skew_gpu.py): with a seed set (SWR_SKEW=<seed> or ops.set_skew(seed)) every fork
# point of the step injects an idle-spinning kernel of pseudo-random length (0 .. 400 us, swr_spin_us) on the forked
# stream and / or on the forking one, stretching the branches of the stream graph against each other.  Results must not
# change by a bit: a missing cross-stream edge (a reader that merely happened to start after its writer) shows up as a
# difference.  Works under hipGraph capture (the spin becomes a kernel node of fixed length).
_SKEW = {"seed": int(os.environ["SWR_SKEW"]) if os.environ.get("SWR_SKEW") else None, "n": 0}
_SKEW_US = (0, 15, 50, 140, 400)


def set_skew(seed):
    """seed (int) switches the harness on, None off; the sequence of spin lengths restarts."""
    _SKEW["seed"], _SKEW["n"] = (None if seed is None else int(seed)), 0


# ---- schedule stamps (measurement aid): SWR_STAMPS=all or a comma list of point names.  `_stamp(name)` enqueues a one-lane kernel
# on the CURRENT stream that writes the device wall clock; `read_stamps()` -> {name: us since the earliest stamp} of the last pass.
_STAMPS = {"on": os.environ.get("SWR_STAMPS", ""), "buf": None, "slot": {}}


def _stamp(name):
    want = _STAMPS["on"]
    if not want or (want != "all" and name not in want.split(",")):
        return
    if _STAMPS["buf"] is None:
        _STAMPS["buf"] = torch.zeros(64, dtype=torch.int64, device="cuda")
    slot = _STAMPS["slot"].setdefault(name, len(_STAMPS["slot"]))
    H.check(lib.swr_stamp(_STAMPS["buf"].data_ptr() + 8 * slot, H.stream()), "swr_stamp")


def read_stamps():
    if _STAMPS["buf"] is None:
        return {}
    v = _STAMPS["buf"].cpu().tolist()
    t = {n: v[i] for n, i in _STAMPS["slot"].items() if v[i]}
    t0 = min(t.values()) if t else 0
    return {n: (x - t0) / 100.0 for n, x in sorted(t.items(), key=lambda kv: kv[1])}


def _skew(point):
    if _SKEW["seed"] is None:
        return
    _SKEW["n"] += 1
    h = (_SKEW["seed"] * 1000003 + _SKEW["n"] * 7919 + point * 104729) % 2147483647
    us = _SKEW_US[(h >> 3) % len(_SKEW_US)]
    if us:
        H.check(lib.swr_spin_us(us, H.stream()), "swr_spin_us")


# ---- the optimizer's step bookkeeping as a rider of the fused loss launch (swr.h swr_select_bce_fwd_adv): the trainer offers a
# callable that returns (hyper, hist, hist_cap) -- and books the step on the host -- when the advance may ride, or None; the
# fused select + BCE launch of the step takes it; `flush_loss_rider()` (after the forward pass) runs it as the ordinary side job
# when no such launch came by.  SWR_LOSS_RIDER=0: never ride.
LOSS_RIDER = os.environ.get("SWR_LOSS_RIDER", "1") != "0"
_rider = {"offer": None, "fallback": None}


def offer_loss_rider(offer, fallback):
    """Single-stream steps (short batches) only: there the 1-thread launch sits on the critical path (-1.5 us per step at the
    8 192-row shard of config 2 and at config 1); a step with side branches runs it on the forward-time fork, off the critical
    path, where riding the loss launch instead was measured 1-3 us SLOWER (config 2 at batch 65 536)."""
    if LOSS_RIDER and not SIDE_STREAM:
        _rider["offer"], _rider["fallback"] = offer, fallback
    else:
        _rider["offer"] = _rider["fallback"] = None
        add_side_job(fallback, backward_needs=False)


def take_loss_rider():
    """-> (hyper, hist, cap) tensors / int for the launch's rider arguments, or (None, None, 0)."""
    offer, _rider["offer"] = _rider["offer"], None
    if offer is None:
        return None, None, 0
    got = offer()
    if got is None:
        return None, None, 0          # (the fallback stays: flush_loss_rider runs it)
    _rider["fallback"] = None
    return got


def flush_loss_rider():
    fb, _rider["fallback"], _rider["offer"] = _rider["fallback"], None, None
    if fb is not None:
        fb()


def add_side_job(fn, backward_needs=True):
    """Run `fn()` on the side stream inside the next forward-time fork (EmbedGather.forward); `run_side_jobs()` runs
    whatever is still pending on the current stream.  `backward_needs=False`: nothing in the backward pass reads what the
    job writes (the optimizer's step counter); a job that returns False launched nothing (a zero_grad with nothing to fill)."""
    _side["jobs"].append((fn, backward_needs))


def run_side_jobs():
    """-> True when a job launched something the backward pass has to wait for."""
    jobs, _side["jobs"] = _side["jobs"], []
    needed = False
    for fn, backward_needs in jobs:
        launched = fn()
        needed = needed or (backward_needs and launched is not False)
    return needed


def _fork_extras():
    """Inside the forward-time fork, on the side stream: pending one-shot jobs and the transposed copies of the weights
    whose dX product wants W^T (registered by the previous step's backward): launches that would otherwise sit on
    the critical path of the backward pass.  An event marks their end: the backward pass needs THEM from its first
    kernel on (zero_grad), the sort that follows on the same stream only when the embedding backward runs."""
    _side["epoch"] += 1
    needed = run_side_jobs()
    for ent in _side["wt"].values():
        needed = True
        if ent.get("sel") is not None:
            torch.index_select(ent["src"].t(), 0, ent["sel"], out=ent["buf"])
        else:
            src = ent["src"]
            if src.is_contiguous() and src.dtype == torch.float32 and src.dim() == 2:
                Gq = ent.get("G", 1)
                H.check(lib.swr_transpose_groups(H.ptr(src), Gq, src.shape[0] // Gq, src.shape[1], H.ptr(ent["buf"]), H.stream()),
                        "swr_transpose_groups")              # (was an ATen strided copy per layer and step)
            else:
                ent["buf"].copy_(src.t())
        ent["epoch"] = _side["epoch"]
    if needed:
        _side["extras_ev"] = torch.cuda.Event()
        _side["extras_ev"].record(torch.cuda.current_stream())
    else:
        # nothing the backward pass reads was launched here (the optimizer cleared the gradients it consumed, no W^T copies):
        # its first kernel takes no edge from this branch -- in a replayed graph such an edge parks that kernel in the branch's
        # queue, behind the sort
        _side["extras_ev"] = "none"


def join_side_extras():
    """Before the backward pass: wait for the one-shot jobs of the forward-time fork (zero_grad, W^T copies), NOT for the
    sort behind them -- that join is the embedding backward's own (join_side_streams(dw=False) there), where the main stream
    has work queued; here it is idle and a join on the sort's last kernel costs ~8 us of edge latency."""
    _flush_deferred()
    ev = _side.get("extras_ev")
    if ev is None:
        join_side_streams()
        return
    _side["extras_ev"] = None
    if ev != "none":
        torch.cuda.current_stream().wait_event(ev)


def _transposed_weight(W):
    """W^T (contiguous) for the dX product: the copy made by this step's forward-time fork when there is one."""
    key = (W.data_ptr(), W.shape[0], W.shape[1])
    ent = _side["wt"].get(key)
    if ent is not None and ent["epoch"] == _side["epoch"] and SIDE_STREAM:
        join_side_streams(dw=False)
        return ent["buf"]
    Wt = transpose_groups(W, 1)
    if SIDE_STREAM and len(_side["wt"]) < 64:
        _side["wt"][key] = {"src": W.detach(), "buf": torch.empty_like(Wt), "epoch": -1}
    return Wt


WT_GROUPS_FORK = os.environ.get("SWR_WT_GROUPS_FORK", "1") != "0"


def _transposed_groups(W, G, stable):
    """transpose_groups(W, G) for a grouped dX product, made by this step's forward-time fork when the step has one and the
    weights are `stable` (a zero-copy view of arena parameters: same address and live values every step -- NOT a tensor derived
    per step, e.g. STAR's effective weights): the copy leaves the backward pass's critical path (config 6: seven of them, ~15 us
    each at 8 x [128, 452] .. [32, 64])."""
    if not (stable and SIDE_STREAM and WT_GROUPS_FORK):
        return transpose_groups(W, G)
    key = (W.data_ptr(), W.shape[0], W.shape[1], "g", G)
    ent = _side["wt"].get(key)
    if ent is not None and ent["epoch"] == _side["epoch"]:
        join_side_streams(dw=False)
        return ent["buf"]
    Wt = transpose_groups(W, G)
    if len(_side["wt"]) < 64 and W.is_contiguous() and W.dtype == torch.float32:
        _side["wt"][key] = {"src": W.detach(), "buf": torch.empty_like(Wt), "epoch": -1, "G": G}
    return Wt


def transpose_groups(W, G):
    """[G * N, K] (G stacked [N, K] matrices) -> [G * K, N] (each transposed): one swr_transpose_groups launch."""
    N, K = W.shape[0] // G, W.shape[1]
    if not (W.is_contiguous() and W.dtype == torch.float32 and W.is_cuda):
        return W.reshape(G, N, K).transpose(1, 2).contiguous().reshape(G * K, N)
    Wt = torch.empty((G * K, N), dtype=torch.float32, device=W.device)
    H.check(lib.swr_transpose_groups(H.ptr(W), G, N, K, H.ptr(Wt), H.stream()), "swr_transpose_groups")
    return Wt


def _selected_wt(W, sel):
    """Rows `sel` of W^T (contiguous [len(sel), N]): the operand of a dX product restricted to the columns `sel` of x.
    Prepared by this step's forward-time fork when one is registered (like _transposed_weight)."""
    key = (W.data_ptr(), W.shape[0], W.shape[1], sel.data_ptr())
    ent = _side["wt"].get(key)
    if ent is not None and ent["epoch"] == _side["epoch"] and SIDE_STREAM:
        join_side_streams(dw=False)
        return ent["buf"]
    Wt = W.t().index_select(0, sel)
    if SIDE_STREAM and len(_side["wt"]) < 64:
        _side["wt"][key] = {"src": W.detach(), "buf": torch.empty_like(Wt), "epoch": -1, "sel": sel}
    return Wt


def _side_stream(dev):
    key = torch.device(dev).index or 0
    if key not in _side["streams"]:
        _side["streams"][key] = torch.cuda.Stream(device=dev)
    return _side["streams"][key]


def join_side_streams(dw=True):
    """Make the current stream wait for everything forked onto the side stream so far.  Cheap when nothing is
    pending.  A cross-stream edge costs ~10 us of latency when the waiting stream is otherwise ready, so callers
    join where the main stream still has work queued behind it (the trainer joins before `loss.backward()`).
    `dw=False` leaves the weight-gradient branch (a stream of its own, `_fork_dw`) running: the embedding backward
    joins only the sort it needs."""
    _flush_deferred()
    if _side["pending"]:
        for st in _side["streams"].values():
            torch.cuda.current_stream(st.device).wait_stream(st)
        _side["pending"] = 0
    if dw and _dw["pending"]:
        for st in _dw["streams"].values():
            torch.cuda.current_stream(st.device).wait_stream(st)
        _dw["pending"] = 0
    if dw:
        _side["keep"].clear()


# weight-gradient branch: the dW chain of the first layer (product + reduce + unfold + small-table gradients) has no
# consumer before the optimizer, and the dX -> K3 chain does not depend on it
_dw = {"streams": {}, "pending": 0, "riders": []}


def _ride_dw(fn, keep):
    """A SMALL weight-gradient product (the towers' grouped dW: 17 + 5 us at config 2) that nothing in the backward pass
    reads: instead of running where autograd reaches it -- on the critical path, ahead of the expert level's backward -- it
    waits for the first layer's weight-gradient branch (_fork_dw) and runs there, behind that product, at no extra fork /
    join edge.  Without such a branch in this backward it runs on the main stream when autograd finishes.  `keep`:
    tensors it reads, held until the join."""
    _dw["riders"].append(fn)
    _side["keep"].append(keep)
    if not _side["queued"]:
        _side["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_join_side)


def _in_backward():
    """True while the autograd engine is running a backward pass on this thread (queue_callback is only legal there)."""
    try:
        return torch._C._current_graph_task_id() >= 0
    except AttributeError:
        return False


def _run_dw_riders():
    riders, _dw["riders"] = _dw["riders"], []
    for fn in riders:
        fn()


def _fork_dw(dev, fn, keep):
    key = torch.device(dev).index or 0
    if key not in _dw["streams"]:
        _dw["streams"][key] = torch.cuda.Stream(device=dev)
    st = _dw["streams"][key]
    st.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(st):
        _skew(3)
        _stamp("d_begin")
        if RIDERS_FIRST:
            _run_dw_riders()
        fn()
        if _DELAY_DW_US:
            H.check(lib.swr_spin_us(_DELAY_DW_US, H.stream()), "swr_spin_us")
        _stamp("d_dw_end")
        _run_dw_riders()
        _stamp("d_end")
    _skew(4)
    _dw["pending"] += 1
    _side["keep"].append(keep)
    if not _side["queued"]:
        _side["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_join_side)


def _join_side():
    _side["queued"] = False
    _stamp("m_autograd_end")
    _run_dw_riders()              # (no weight-gradient branch was forked in this backward: they run here)
    join_side_streams()


def _fork_side(dev, fn, after_event=None):
    """Run `fn()` (kernel launches / allocations) on the side stream, ordered after `after_event` (or after
    everything enqueued so far on the current stream)."""
    side = _side_stream(dev)
    if isinstance(after_event, str):
        pass                               # the dependency was taken when the work was deferred (_defer_side)
    elif after_event is not None:
        side.wait_event(after_event)
    else:
        side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        _skew(1)
        fn()
    _skew(2)
    _side["pending"] += 1


def _defer_side(dev, fn):
    """Fork `fn` onto the side stream: the side stream takes its dependency on the main stream HERE (after the gather),
    but its kernels are enqueued only after the main stream's next kernel (`_flush_deferred()` is called by the GEMM
    launcher) -- launched at once they are dispatched ahead of that kernel and delay it by their launch latency; in a
    captured graph the main stream's kernel then comes first in capture order.  (Measured at config 2: this form 0.700
    ms, fork at once 0.709, edge and launches both later 0.711, event-record nodes stall the branch until the join.)"""
    _side_stream(dev).wait_stream(torch.cuda.current_stream(dev))
    _side["deferred"].append((dev, fn, "nowait"))


def _flush_deferred():
    if _side["deferred"]:
        todo, _side["deferred"] = _side["deferred"], []
        for dev, fn, ev in todo:
            _fork_side(dev, fn, ev)


def _on_side_stream(dev, fn, keep):
    """Backward-pass helper: run `fn()` on the side stream after everything enqueued so far; `keep`: tensors the
    side work reads, held until the join (queued for the end of the autograd pass)."""
    _fork_side(dev, fn)
    _side["keep"].append(keep)
    if not _side["queued"]:
        _side["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_join_side)


# ---- split backward (data-parallel step): work whose results only the optimizer needs -- the weight-gradient products
# and the small tables' gradients -- is held back until `run_late_jobs()`, so that the large tables' row lists are
# ready (and on their way to the other ranks) as early as possible and the held-back work overlaps the all-gather.
_late = {"on": False, "jobs": [], "rows_event": None, "rows_recorded": False}


def split_backward(on, rows_event=None):
    """Set by parallel.DataParallelStep around its forward+backward; whoever sets it must call run_late_jobs().
    `rows_event` (the one-graph step): recorded on the main stream behind the large tables' row lists, and the small tables'
    sums are NOT held back but forked onto the side stream at that point -- the caller's all-gather waits for the event alone
    (not for the weight-gradient branch or those sums), and nothing is left to serialise behind the row lists."""
    _late["on"] = bool(on)
    _late["rows_event"] = rows_event if on else None
    if on:
        _late["rows_recorded"] = False


def rows_event_recorded():
    """True when the last split backward recorded its `rows_event` (a model without large tables never reaches that point)."""
    return _late["rows_recorded"]


def run_late_jobs():
    jobs, _late["jobs"] = _late["jobs"], []
    for fn in jobs:
        fn()


def abort_step():
    """Drop every piece of per-step state an aborted forward / backward (a failed hipGraph capture) may have left behind:
    held-back jobs, deferred forks, riders, pending counters and the tensors kept alive for the side branches.  The caller
    synchronises the device first and zeroes the gradients afterwards; nothing queued here is run."""
    _late["jobs"] = []
    _late["on"], _late["rows_event"], _late["rows_recorded"] = False, None, False
    _side["deferred"] = []
    _side["jobs"] = []
    _side["pending"] = 0
    _side["queued"] = False
    _side["keep"].clear()
    _dw["pending"] = 0
    _dw["riders"] = []
    _rider["offer"] = _rider["fallback"] = None


def _split_like(flat, tensors):
    """Slices of `flat` (first-dim concatenation) shaped like each of `tensors`."""
    out, off = [], 0
    for t in tensors:
        n = t.shape[0] if t.dim() > 0 else 1
        out.append(flat[off:off + n].reshape(t.shape))
        off += n
    return out


# =========================================================================== embedding gather
class _GatherPlan:
    """Static description of one EmbeddingLayer lookup (built per call, cheap)."""
    __slots__ = ("sparse", "dense", "width", "ld", "dense_limit_bytes", "lazy", "want_grad", "bags", "onehot", "oh", "ctx", "fold", "wide", "fl")
    # onehot: the caller promises that ONE LinearBNAct consumes the lookup (see OneHotInfo); oh: the block laid out by forward
    # bags: SequenceFeature lookups, dicts(wpos, idx [B, L], vocab, dim, col, L, mode 0 sum / 1 mean / 2 concat, pad, seed)


ONEHOT = os.environ.get("SWR_ONEHOT", "1") != "0"
ONEHOT_MAX_VOCAB = int(os.environ.get("SWR_ONEHOT_MAX_VOCAB", "16"))
ONEHOT_MAX_WIDTH = int(os.environ.get("SWR_ONEHOT_MAX_WIDTH", "128"))
FOLD = os.environ.get("SWR_FOLD", "1") != "0"       # the lookup writes [E_big | dense | one-hot] only; the layer folds the rest
# fused lookup + first layer (csrc/first_layer.hip): the lookup writes NOTHING but keys / one-hot bits / piece offsets; the
# consuming layer's products fetch table rows through the keys ("0": the folded layout is written as before)
FUSED_LOOKUP = os.environ.get("SWR_FUSED_LOOKUP", "1") != "0"
_DELAY_DW_US = int(os.environ.get("SWR_DELAY_DW_US", "0"))         # measurement aids: which branch behind dX is the critical one?
_DELAY_EMBED_US = int(os.environ.get("SWR_DELAY_EMBED_US", "0"))
_SORT_DELAY_US = int(os.environ.get("SWR_SORT_DELAY_US", "0"))   # measurement aid: is the forward-time sort on the step's critical path?


class OneHotInfo(object):
    """One-hot block of the small tables behind an embedding concat (csrc/embed_fwd.hip, swr_embed_gather_fwd_onehot),
    handed from the lookup to the ONE layer that consumes it (LinearBNAct), whose backward then
      * takes the small tables' gradients from its own weight-gradient product: dZ^T [E | one-hot] gives dW and the
        per-row segment sums S of dZ in one launch; grad_t = S_t W_t (swr_onehot_table_grads) -- no K3 for them;
      * computes dX only for the columns of the other tables (`sel`), compactly, and passes it to the lookup's backward
        through `ctx` (autograd carries a zero-stride placeholder)."""
    __slots__ = ("ctx", "oh_col", "oh_width", "tables", "tables_p", "params", "sel", "n_sel", "compact",
                 "fold", "wide", "col0", "Kp", "src", "inv", "ohtab", "K", "fl")

    def materialize(self):
        """The written folded layout [B, ld] of a FUSED lookup (`fl`), produced on demand by the ordinary gather launch: for
        a consumer the fused product does not take (more than 160 output columns, SWR_GEMM=f32 / bf16)."""
        if self.wide is None:
            f = self.fl
            wide = torch.empty((f["B"], f["ld"]), dtype=torch.float32, device=f["dev"])
            sp, dn, oh_off = f["gather"]
            H.check(lib.swr_embed_gather_fwd_onehot(sp, f["ns"], dn, f["nd"], f["B"], H.ptr(wide), f["ld"], None, oh_off, f["pad_col"],
                                                    f["oh_col"], f["oh_width"], H.ptr(H.err_flag(f["dev"])), H.stream()),
                    "swr_embed_gather_fwd_onehot")
            self.wide = wide
        return self.wide


def _grad_slot_layout(plan, weights, n_grad_slots):
    """(live slots, uses per table, table ids): dense-gradient tables first, row-sparse (large) tables last."""
    live = plan.sparse[:n_grad_slots]
    uses = {}
    for s, (wpos, *_rest) in enumerate(live):
        uses.setdefault(wpos, []).append(s)
    order = sorted(uses, key=lambda w: (weights[w].numel() * 4 > plan.dense_limit_bytes, w))
    return live, uses, {w: i for i, w in enumerate(order)}


class EmbedGather(Function):
    """K1 forward / K3 backward (basic/layers.py:64-105)."""

    @staticmethod
    def forward(ctx, plan, *weights):
        # plan.sparse: list of (weight_pos, idx tensor, vocab, dim, out_col, hash_seed); plan.dense: (values, out_col)
        dev = weights[0].device if weights else plan.dense[0][0].device
        bags = getattr(plan, "bags", None) or []
        B = (plan.sparse[0][1] if plan.sparse else (bags[0]["idx"] if bags else plan.dense[0][0])).shape[0]
        if _late["jobs"]:
            raise H.SwrError("a forward pass with held-back gradient work of the previous backward pending "
                             "(split backward: run_late_jobs() was not called)")
        if _side["queued"] and not _in_backward():
            # a backward pass died between queueing its end-of-pass callback and running it (an exception in a kernel
            # launcher): without this the riders of every later pass would wait for a callback that is never queued again
            _side["queued"] = False
            _dw["riders"].clear()
        # slots whose table takes a gradient first: the backward reduces exactly that prefix
        plan.sparse = sorted(plan.sparse, key=lambda s: not weights[s[0]].requires_grad)
        ctx.n_grad_slots = sum(1 for s in plan.sparse if weights[s[0]].requires_grad)
        ns, nd = len(plan.sparse), len(plan.dense)
        # one-hot block of the small tables (see OneHotInfo): only when the caller vouches for a single consuming layer
        plan.oh, oh_off, oh_width, oh_col = None, None, 0, 0
        if (ONEHOT and getattr(plan, "onehot", False) and (getattr(plan, "want_grad", False) or plan.onehot == "layout")
                and not bags and 0 < ns <= 64):
            off, tables = 0, []
            for i, (wpos, idx, vocab, dim, col, seed) in enumerate(plan.sparse[:ctx.n_grad_slots]):
                w = weights[wpos]
                if (vocab <= ONEHOT_MAX_VOCAB and seed == 0 and off + vocab <= ONEHOT_MAX_WIDTH and dim % 4 == 0
                        and w.numel() * 4 <= plan.dense_limit_bytes and _grad_alias([w], 4) is not None):
                    tables.append((i, wpos, vocab, dim, off, col))
                    off += vocab
            if tables:
                # one-hot slots go behind the other gradient-taking slots: K3 then handles a PREFIX of the slots (and of
                # the keys, which are laid out in slot order)
                pos = {t[0] for t in tables}
                ng = ctx.n_grad_slots
                plan.sparse = ([sl for i, sl in enumerate(plan.sparse[:ng]) if i not in pos] +
                               [plan.sparse[t[0]] for t in tables] + plan.sparse[ng:])
                first = ng - len(tables)
                tables = [(first + j,) + t[1:] for j, t in enumerate(tables)]
                oh_width = (off + 3) // 4 * 4
                oh_off = (C.c_int32 * ns)(*([-1] * ns))
                for i, _w, _v, _d, o, _c in tables:
                    oh_off[i] = o
                plan.oh = tables
                pad_col, oh_col = plan.width, (plan.width + 3) // 4 * 4
                if FOLD:
                    # folded layout (include/swr.h "folded first layer"): behind the (unwritten) ordinary columns a compact
                    # block [embeddings of the other tables | dense features | 0-padding | one-hot]; the consuming layer
                    # multiplies THAT with its folded weights
                    oh_set = {t[0] for t in tables}
                    col0 = (plan.width + 3) // 4 * 4
                    cc, src, sp_col = col0, [], {}
                    for i, (_wpos, _idx, _vocab, dim, col, _seed) in enumerate(plan.sparse):
                        if i not in oh_set:
                            sp_col[i] = cc
                            src.extend(range(col, col + dim))
                            cc += dim
                    dn_col = []
                    for _vals, col in plan.dense:
                        dn_col.append(cc)
                        src.append(col)
                        cc += 1
                    # fused lookup (FUSED_LOOKUP, csrc/first_layer.hip): k axis in 8-column pieces, 16-column groups -- every
                    # table takes a gradient and has a multiple of 8 columns, at most 16 groups in front of the one-hot block
                    n_pieces = sum(sl[3] // 8 for i, sl in enumerate(plan.sparse) if i not in oh_set) + (len(plan.dense) + 7) // 8
                    # (the LAYOUT -- 16-column groups -- is taken whenever the lookup qualifies; FUSED_LOOKUP only chooses between the
                    # fused launches and the written block, so that the two can be compared bit for bit)
                    # (the limits of fl_build, csrc/first_layer.hip: at least one real piece, one-hot block <= 128 columns, 64 slots,
                    # the bf16 shadows of the small tables below 1 GiB; what depends on the batch is checked per call -- swr_fl_layout)
                    shadow_bytes = sum(sl[2] * (sl[3] // 8) * 48 for i, sl in enumerate(plan.sparse)
                                       if i not in oh_set and weights[sl[0]].numel() * 4 <= plan.dense_limit_bytes)
                    fl_ok = (ctx.n_grad_slots == ns and 1 <= n_pieces <= 32 and len(plan.dense) <= 32 and ns <= 64
                             and (off + 15) // 16 * 16 <= 128 and shadow_bytes < (1 << 30) - (1 << 20)
                             and all(sl[3] % 8 == 0 and weights[sl[0]].is_contiguous() and weights[sl[0]].dtype == torch.float32
                                     and weights[sl[0]].data_ptr() % 16 == 0
                                     for i, sl in enumerate(plan.sparse) if i not in oh_set)
                             and all(weights[sl[0]].numel() * 4 > plan.dense_limit_bytes or sl[2] < (1 << 24) for sl in plan.sparse)
                             and lib.swr_gemm_precision_mode() == 1)
                    Kp = (cc - col0 + 15) // 16 * 16 if fl_ok else (cc - col0 + 3) // 4 * 4
                    if fl_ok:
                        oh_width = (off + 15) // 16 * 16
                    src.extend([-1] * (Kp - (cc - col0)))
                    pad_col, oh_col = cc, col0 + Kp
                    plan.fold = {"col0": col0, "Kp": Kp, "src": tuple(src), "sp_col": sp_col, "dn_col": dn_col, "fl": fl_ok and FUSED_LOOKUP}
                plan.ld = oh_col + oh_width
        fold = getattr(plan, "fold", None) if plan.oh else None
        fl = fold is not None and fold.get("fl", False) and B > 0
        if fl and B * max(1, fold["Kp"] // 8) * 48 >= (1 << 32) - (1 << 24):
            fl = False          # (a lane's 32-bit piece offset cannot reach the batch's gathered pieces: the written folded layout)
        # (fused lookup: nothing of the concat is written -- autograd carries a zero-stride placeholder of its shape)
        out = _zero_scalar(dev).expand(B, plan.ld) if fl else torch.empty((B, plan.ld), dtype=torch.float32, device=dev)
        sp = (H.SparseSlot * max(ns, 1))()
        sp_fl = (H.SparseSlot * max(ns, 1))() if fl else None
        for i, (wpos, idx, vocab, dim, col, seed) in enumerate(plan.sparse):
            H.require_device(idx, weights[wpos])
            if fl:                          # the fused launches see the slot as it is: real width, its column in W
                sp_fl[i] = H.SparseSlot(weights[wpos].data_ptr(), idx.data_ptr(), vocab, dim, H.dtype_code(idx), col, seed)
            if fold is not None:            # compact column; a folded (one-hot) slot writes no embedding at all (dim 0)
                col, dim = (fold["sp_col"][i], dim) if i in fold["sp_col"] else (0, 0)
            sp[i] = H.SparseSlot(weights[wpos].data_ptr(), idx.data_ptr(), vocab, dim, H.dtype_code(idx), col, seed)
        dn = (H.DenseSlot * max(nd, 1))()
        for i, (vals, col) in enumerate(plan.dense):
            H.require_device(vals)
            dn[i] = H.DenseSlot(vals.data_ptr(), H.dtype_code(vals), fold["dn_col"][i] if fold is not None else col)
        # lazily updated tables (optim.LazyRows): bring the rows about to be read up to date first (exact replay)
        seen, behind = set(), []
        for wpos, idx, vocab, dim, col, seed in plan.sparse:
            lazy = getattr(plan, "lazy", {}).get(wpos)
            if lazy is not None and (wpos, idx.data_ptr()) not in seen:
                seen.add((wpos, idx.data_ptr()))
                behind.append((lazy, idx, seed))
        if len(behind) == 1:
            behind[0][0].catchup(behind[0][1], behind[0][2])
        elif behind:
            from .optim import catchup_many
            catchup_many(behind)               # one claim + one replay launch for all large tables of the lookup
        need_keys = ctx.n_grad_slots > 0
        keys = torch.empty(ns * B, dtype=torch.int32, device=dev) if (need_keys and ns and not fl) else None
        flag = H.err_flag(dev)
        plan.fl = None
        if fl:
            # keys, one-hot bits, piece offsets and the fp32-sourced pieces -- one launch; the products of the consuming layer
            # fetch the rows themselves (csrc/first_layer.hip)
            n_k3 = ns - len(plan.oh)
            fp = H.FlPlan()
            fp.sparse_host, fp.n_sparse = C.cast(sp_fl, C.c_void_p), ns
            fp.dense_host, fp.n_dense = C.cast(dn, C.c_void_p), nd
            fp.n_keys, fp.oh_width, fp.oh_off_host, fp.B, fp.N = n_k3, oh_width, C.cast(oh_off, C.c_void_p), B, 0
            q = 0
            for i, (wpos, idx, vocab, dim, col, seed) in enumerate(plan.sparse):
                if i not in fold["sp_col"]:
                    continue
                kind = H.FL_ROWS if weights[wpos].numel() * 4 > plan.dense_limit_bytes else H.FL_PLANES
                for o8 in range(0, dim, 8):
                    fp.piece[q] = H.FlPiece(kind, i, o8, 8, col + o8, 0)
                    q += 1
            for d0 in range(0, nd, 8):
                fp.piece[q] = H.FlPiece(H.FL_DENSE, d0, 0, min(8, nd - d0), plan.dense[d0][1], 0)
                q += 1
            fp.n_real_groups = fold["Kp"] // 16
            while q < 2 * fp.n_real_groups:
                fp.piece[q] = H.FlPiece(H.FL_ZERO, 0, 0, 0, 0, 0)
                q += 1
            offs = H.FlOffsets()
            H.check(lib.swr_fl_layout(C.byref(fp), C.byref(offs)), "swr_fl_layout")
            ws = torch.empty(offs.total, dtype=torch.uint8, device=dev)
            H.check(lib.swr_fl_keys(C.byref(fp), H.ptr(ws), H.ptr(flag), H.stream()), "swr_fl_keys")
            keys = ws[offs.keys:offs.keys + 4 * n_k3 * B].view(torch.int32) if n_k3 else None
            plan.fl = {"plan": fp, "ws": ws, "offs": offs, "keep": (sp_fl, dn, oh_off), "gather": (sp, dn, oh_off), "ns": ns, "nd": nd,
                       "B": B, "ld": plan.ld, "dev": dev, "pad_col": pad_col, "oh_col": oh_col, "oh_width": oh_width}
            plan.wide = None
        elif plan.oh:
            H.check(lib.swr_embed_gather_fwd_onehot(sp, ns, dn, nd, B, H.ptr(out), plan.ld, H.ptr(keys), oh_off, pad_col,
                                                    oh_col, oh_width, H.ptr(flag), H.stream()), "swr_embed_gather_fwd_onehot")
            plan.wide = out
        else:
            H.check(lib.swr_embed_gather_fwd(sp, ns, dn, nd, B, H.ptr(out), plan.ld, H.ptr(keys), H.ptr(flag), H.stream()),
                    "swr_embed_gather_fwd")
        # SequenceFeature columns: pooled lookups (csrc/embed_bag.hip), one launch each, into their columns of `out`
        ctx.bags = []
        for bag in bags:
            w, idx = weights[bag["wpos"]], bag["idx"]
            H.require_device(idx, w)
            idx = idx.contiguous()
            lazy = getattr(plan, "lazy", {}).get(bag["wpos"])
            if lazy is not None:
                lazy.catchup(idx.reshape(-1) if bag["mode"] == 2 or bag["pad"] is None else
                             torch.where(idx == bag["pad"], torch.full_like(idx, -1), idx).reshape(-1), bag["seed"])
                if bag["mode"] != 2:
                    # masked positions reach the backward as (row 0, weight 0): row 0 then sits in the row list the optimizer's
                    # row kernel marks current, so its pending decay-only steps must be replayed here too (un-hashed row id)
                    lazy.catchup(_zero_index(dev), 0)
            want = w.requires_grad and getattr(plan, "want_grad", False)
            bkeys = torch.empty(B * bag["L"], dtype=torch.int32, device=dev) if want else None
            bwts = torch.empty(B * bag["L"], dtype=torch.float32, device=dev) if want else None
            H.check(lib.swr_embed_bag_fwd(H.ptr(w), bag["vocab"], bag["dim"], H.ptr(idx), H.dtype_code(idx), B, bag["L"],
                                          bag["mode"], int(bag["pad"] is not None), int(bag["pad"] or 0), bag["seed"],
                                          H.ptr(out), plan.ld, bag["col"], H.ptr(bkeys), H.ptr(bwts), H.ptr(flag),
                                          H.stream()), "swr_embed_bag_fwd")
            if want:
                ctx.bags.append((bag, bkeys, bwts))
        ctx.plan, ctx.keys, ctx.B = plan, keys, B
        ctx.weights = weights            # identity / shapes only
        ctx.presorted = None
        ctx.fused_dx = None              # (compact dX, {position in plan.sparse: first compact column}) from the consuming layer
        ctx.n_k3_slots = ctx.n_grad_slots - (len(plan.oh) if plan.oh else 0)      # without the one-hot tables
        plan.ctx = ctx
        if _SIDE_MODE == "auto" and not _in_backward() and getattr(plan, "want_grad", False):
            # (only a lookup that a backward pass may follow decides: an evaluation forward between a training step's forward and its
            # backward must not flip the choice the backward-time forks read)
            global SIDE_STREAM
            # (the lookup opens the step: every fork decision of the step follows it.)  A replayed multi-stream graph costs the host
            # ~7 us per node, so the forks pay where they hide more than that: from SIDE_MIN_BATCH rows on for every model, and from
            # SIDE_MIN_BATCH_FUSED rows on for a step that has BOTH a fused first layer (few launches: ~28 nodes) and large tables
            # (their sort, the direct sums and the first layer's weight gradient are three independent chains behind dX) -- config 2
            # at 4 096 / 8 192 / 16 384 rows: 0.189 / 0.207 / 0.260 ms on one stream, 0.165 / 0.185 / 0.210 with the forks; SharedBottom
            # (no large table), PLE and STAR (generic layers, 38 - 113 nodes) are faster on one stream up to 16 384 rows
            big = any(weights[wp].numel() * 4 > plan.dense_limit_bytes for wp, *_r in plan.sparse) if plan.sparse else False
            fused = getattr(plan, "fl", None) is not None
            SIDE_STREAM = B >= SIDE_MIN_BATCH or (fused and big and B >= SIDE_MIN_BATCH_FUSED)
        if need_keys and ns and B > 0 and SIDE_STREAM and getattr(plan, "want_grad", False):   # a backward may follow
            # the grouping of the large tables' entries by row needs only the keys: run it NOW on the side stream,
            # hidden behind the rest of the forward and backward pass; the backward joins before it reduces
            live, _uses, table_id = _grad_slot_layout(plan, weights, ctx.n_k3_slots)
            proto = (H.EmbedGradSlot * max(1, len(live)))()
            pos = 0
            for s, (wpos, idx, vocab, dim, col, seed) in enumerate(live):
                mode = 1 if weights[wpos].numel() * 4 > plan.dense_limit_bytes else 0
                if plan.oh:
                    # the backward will see the consuming layer's COMPACT dX (OneHotInfo.compact: the K3 slots packed in
                    # order); the workspace layout depends on which lookups are adjacent (direct-sum groups -> slab count)
                    col, pos = pos, pos + dim
                proto[s] = H.EmbedGradSlot(vocab, dim, col, table_id[wpos], mode, None, None, None)
            nbytes = lib.swr_embed_bwd_workspace_bytes(proto, len(live), B) if live else 0
            if not live:
                _defer_side(dev, _fork_extras)         # nothing to sort: the fork still carries zero_grad and the W^T copies
            if nbytes:
                box = {"nbytes": nbytes}

                def sort_now(box=box, proto=proto, n=len(live), keys=keys, B=B, dev=dev):
                    # the one-shot jobs first (zero_grad, the optimizer's step counter, W^T copies): the main stream joins
                    # this branch right before the backward pass, and the dozen latency-bound sort launches are what it
                    # would otherwise wait for LAST (measured: 30 us of join stall at config 2 with the jobs behind the sort)
                    # (the sort on a branch of its own -- so that nothing that joins the one-shot jobs queues behind it -- was
                    # measured: 0.487 vs 0.463 ms per step, the runtime then starts it in the middle of the backward pass)
                    _stamp("s_begin")
                    _fork_extras()
                    _stamp("s_extras_end")
                    if _SORT_DELAY_US:
                        H.check(lib.swr_spin_us(_SORT_DELAY_US, H.stream()), "swr_spin_us")
                    box["ws"] = torch.empty(box["nbytes"], dtype=torch.uint8, device=dev)
                    H.check(lib.swr_embed_bwd_sort(proto, n, H.ptr(keys), B, H.ptr(box["ws"]), box["nbytes"], H.stream()),
                            "swr_embed_bwd_sort")
                    _stamp("s_sort_end")
                _defer_side(dev, sort_now)
                ctx.presorted = box
        return out[:, :plan.width] if plan.width != plan.ld else out

    @staticmethod
    @once_differentiable
    def backward(ctx, dE):
        plan, B, weights = ctx.plan, ctx.B, ctx.weights
        no_plain = ctx.keys is None or ctx.n_grad_slots == 0
        if B == 0 or (no_plain and not ctx.bags):
            return (None,) + tuple(None if not w.requires_grad else torch.zeros_like(w) for w in weights)
        fused = ctx.fused_dx
        ctx.fused_dx = None
        if fused is not None:
            # the consuming layer computed dX only for the columns K3 needs, compactly (OneHotInfo); `dE` is a placeholder
            dE, compact = fused
            if ctx.n_k3_slots == 0:
                return (None,) + (None,) * len(weights)
        else:
            if getattr(plan, "fold", None) and plan.oh:
                raise H.SwrError("a lookup made with onehot=True was not consumed by the ONE fused layer it was promised to "
                                 "(EmbeddingLayer.forward(..., onehot=True)): its ordinary columns were never written")
            dE, compact = H.f32c(dE), None
        dev = dE.device
        if no_plain:
            grads, sparse_out = [None] * len(weights), {}
            EmbedGather._bags_backward(ctx, dE, grads, sparse_out)
            for wpos, (urow, ugrad) in sparse_out.items():
                if getattr(weights[wpos], "_swr_sparse_grad", None) is not None:
                    raise H.SwrError("a row-sparse table gradient is already pending (backward() twice without zero_grad())")
                weights[wpos]._swr_sparse_grad = (urow, ugrad)
                weights[wpos]._swr_sparse_local = True
            return (None,) + tuple(None if isinstance(g, tuple) else g for g in grads)
        # tables: dense gradient when small, row-sparse entries when large; sparse tables take the largest ids
        live, uses, table_id = _grad_slot_layout(plan, weights, ctx.n_k3_slots if compact is not None else ctx.n_grad_slots)
        grads = [None] * len(weights)
        sparse_out = {}
        slots = (H.EmbedGradSlot * len(live))()
        for s, (wpos, idx, vocab, dim, col, seed) in enumerate(live):
            if compact is not None:
                col = compact[s]                       # this slot's first column in the compact dX
            w = weights[wpos]
            sparse_mode = w.numel() * 4 > plan.dense_limit_bytes
            if sparse_mode:
                if wpos not in sparse_out:
                    cnt = len(uses[wpos]) * B
                    pre = getattr(w, "_swr_sparse_out", None)      # caller-owned outputs (the data-parallel send buffer)
                    if pre is not None and tuple(pre[0].shape) == (cnt,) and tuple(pre[1].shape) == (cnt, dim):
                        sparse_out[wpos] = pre
                    else:
                        sparse_out[wpos] = (torch.empty(cnt, dtype=torch.int32, device=dev),
                                            torch.empty((cnt, dim), dtype=torch.float32, device=dev))
                urow, ugrad = sparse_out[wpos]
                slots[s] = H.EmbedGradSlot(vocab, dim, col, table_id[wpos], 1, None, urow.data_ptr(), ugrad.data_ptr())
            else:
                if grads[wpos] is None:
                    direct = _grad_alias([w], 4)
                    if direct is not None:
                        grads[wpos] = ("direct", direct)
                    else:
                        grads[wpos] = torch.empty_like(w, memory_format=torch.contiguous_format)
                if isinstance(grads[wpos], tuple):          # accumulate into the gradient arena (mode 2)
                    slots[s] = H.EmbedGradSlot(vocab, dim, col, table_id[wpos], 2, grads[wpos][1].data_ptr(), None, None)
                else:
                    slots[s] = H.EmbedGradSlot(vocab, dim, col, table_id[wpos], 0, grads[wpos].data_ptr(), None, None)
        ns = len(live)
        nbytes = lib.swr_embed_bwd_workspace_bytes(slots, ns, B)
        if nbytes == 0:
            raise H.SwrError("swr_embed_bwd: unsupported lookup shape (more than 40 lookup slots)")
        if ctx.presorted is not None and ctx.presorted["nbytes"] == nbytes:
            join_side_streams(dw=False)                               # the sort forked in forward() (no-op if joined)
            _stamp("m_embed_bwd_begin")
            if _DELAY_EMBED_US:
                H.check(lib.swr_spin_us(_DELAY_EMBED_US, H.stream()), "swr_spin_us")
            ws = ctx.presorted["ws"]
            all_direct = all(g is None or isinstance(g, tuple) for g in grads)      # dense gradients go to the arena
            if _late["on"] and sparse_out and all_direct:
                # split backward: the row lists of the large tables now, everything else when the caller says so
                def reduce_part(part, slots=slots, keys=ctx.keys, dE=dE, ws=ws):
                    H.check(lib.swr_embed_bwd_reduce_part(slots, ns, H.ptr(keys), H.ptr(dE), dE.stride(0), B, part, H.ptr(ws),
                                                          nbytes, H.ptr(H.err_flag(dev)), H.stream()), "swr_embed_bwd_reduce_part")
                if _late["rows_event"] is not None:
                    # one-graph step: nothing is held back -- the small tables' sums go to the side stream (free since the sort
                    # was joined) behind the sorted reduce; the event marks the row lists.  (Beside the sorted reduce -- legal when
                    # no dense table is sorted -- they slow it: world-1 0.424-0.429 against 0.413-0.421 ms.)
                    reduce_part(1)
                    _late["rows_event"].record(torch.cuda.current_stream())
                    _late["rows_recorded"] = True
                    _on_side_stream(dev, lambda: reduce_part(2), (dE, ctx.keys, ws))      # (operands held until the join)
                else:
                    reduce_part(1)
                    _late["jobs"].append(lambda: reduce_part(2))
            # (measured and dropped: the sorted reduce + row lists on the weight-gradient branch while the direct sums stay
            # here -- the two halves are independent when no dense table is sorted -- 0.524 vs 0.494 ms per step)
            else:
                H.check(lib.swr_embed_bwd_reduce(slots, ns, H.ptr(ctx.keys), H.ptr(dE), dE.stride(0), B, H.ptr(ws), nbytes,
                                                 H.ptr(H.err_flag(dev)), H.stream()), "swr_embed_bwd_reduce")
        else:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            H.check(lib.swr_embed_bwd(slots, ns, H.ptr(ctx.keys), H.ptr(dE), dE.stride(0), B, H.ptr(ws), nbytes,
                                      H.ptr(H.err_flag(dev)), H.stream()), "swr_embed_bwd")
        if ctx.bags:
            EmbedGather._bags_backward(ctx, dE, grads, sparse_out)
        for wpos, (urow, ugrad) in sparse_out.items():
            # row-sparse gradient of a large table: consumed by FusedAdam (optim.py); `.grad` stays None
            if getattr(weights[wpos], "_swr_sparse_grad", None) is not None:
                raise H.SwrError("a row-sparse table gradient is already pending (a second lookup of the same large table in "
                                 "one backward pass, or backward() twice without zero_grad()): row lists do not accumulate "
                                 "-- use one fused lookup per step, or raise dense_table_limit_bytes "
                                 "(SwrModule.set_dense_table_limit)")
            weights[wpos]._swr_sparse_grad = (urow, ugrad)
            weights[wpos]._swr_sparse_local = True     # every row listed was looked up (and caught up) by THIS forward
        for i, g in enumerate(grads):
            if isinstance(g, tuple):
                weights[i]._swr_touched = True
                weights[i]._swr_grad_clean = False
                grads[i] = None
        return (None,) + tuple(grads)

    @staticmethod
    def _bags_backward(ctx, dE, grads, sparse_out):
        """SequenceFeature lookups: one gradient row per looked-up position (scaled by the pooling weight, 0 where
        masked), then the ordinary K3 over B * L "samples" of one slot.  Runs after the plain slots' K3: a table shared
        with a plain lookup (`shared_with`) accumulates into the same dense gradient."""
        plan, B, weights = ctx.plan, ctx.B, ctx.weights
        dev = dE.device
        for bag, bkeys, bwts in ctx.bags:
            wpos, dim, L = bag["wpos"], bag["dim"], bag["L"]
            w = weights[wpos]
            n = B * L
            rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
            H.check(lib.swr_embed_bag_bwd_expand(H.ptr(dE), dE.stride(0), bag["col"], dim, L, int(bag["mode"] == 2), H.ptr(bwts),
                                                 B, H.ptr(rows), H.stream()), "swr_embed_bag_bwd_expand")
            slot = (H.EmbedGradSlot * 1)()
            merge_with = None
            if w.numel() * 4 > plan.dense_limit_bytes:
                urow = torch.empty(n, dtype=torch.int32, device=dev)
                ugrad = torch.empty((n, dim), dtype=torch.float32, device=dev)
                slot[0] = H.EmbedGradSlot(bag["vocab"], dim, 0, 0, 1, None, urow.data_ptr(), ugrad.data_ptr())
                merge_with = sparse_out.get(wpos)          # the table's row list from its plain lookups / an earlier bag
                sparse_out[wpos] = (urow, ugrad)
            else:
                g = grads[wpos]
                if g is None:
                    direct = _grad_alias([w], 4)
                    g = grads[wpos] = ("direct", direct) if direct is not None else torch.zeros_like(w, memory_format=torch.contiguous_format)
                target = g[1] if isinstance(g, tuple) else g
                if isinstance(g, tuple):
                    w._swr_touched = True
                    w._swr_grad_clean = False
                slot[0] = H.EmbedGradSlot(bag["vocab"], dim, 0, 0, 2, target.data_ptr(), None, None)
            nbytes = lib.swr_embed_bwd_workspace_bytes(slot, 1, n)
            if nbytes == 0:
                raise H.SwrError("swr_embed_bwd: unsupported SequenceFeature shape (batch x length above 2^24)")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            H.check(lib.swr_embed_bwd(slot, 1, H.ptr(bkeys), H.ptr(rows), dim, n, H.ptr(ws), nbytes, H.ptr(H.err_flag(dev)),
                                      H.stream()), "swr_embed_bwd(bag)")
            if merge_with is not None:
                # a row-sparse table with several lookups: the lists are merged into one with every row listed once
                # (the optimizer's row kernel applies one Adam update per listed row)
                from .parallel import hip_merge_rows
                urow, ugrad = sparse_out[wpos]
                sparse_out[wpos] = hip_merge_rows(torch.cat([merge_with[0], urow]), torch.cat([merge_with[1], ugrad]), bag["vocab"])


# =========================================================================== Linear (+BN) (+act)
def _bn_train_finalize(bn, gamma, beta, partials, n_tiles, M, Ntot, dev):
    """Merge the per-tile (mean, M2) pairs into batch statistics, update the running statistics in place and return
    (mean, rstd, scale, shift) with BN(z) = scale * z + shift."""
    mean = torch.empty(Ntot, dtype=torch.float32, device=dev)
    rstd = torch.empty(Ntot, dtype=torch.float32, device=dev)
    scale = torch.empty(Ntot, dtype=torch.float32, device=dev)
    shift = torch.empty(Ntot, dtype=torch.float32, device=dev)
    rm, rv, nbt = _cat_params(bn["running_mean"]), _cat_params(bn["running_var"]), _cat_params(bn["nbt"])
    copy_back = rm.data_ptr() != bn["running_mean"][0].data_ptr()
    H.check(lib.swr_bn_finalize(H.ptr(partials), n_tiles, M, Ntot, H.ptr(gamma), H.ptr(beta), bn["eps"],
                                bn["momentum"], H.ptr(rm), H.ptr(rv), H.ptr(nbt), nbt.numel(), H.ptr(mean),
                                H.ptr(rstd), H.ptr(scale), H.ptr(shift), H.stream()), "swr_bn_finalize")
    if copy_back:      # buffers were not adjacent: scatter the updated copies back
        for dst, src in zip(bn["running_mean"], _split_like(rm, bn["running_mean"])):
            dst.copy_(src)
        for dst, src in zip(bn["running_var"], _split_like(rv, bn["running_var"])):
            dst.copy_(src)
        for dst, src in zip(bn["nbt"], _split_like(nbt, bn["nbt"])):
            dst.copy_(src)
    return mean, rstd, scale, shift


class LinearBNAct(Function):
    """[Linear -> BatchNorm1d -> activation] of an MLP block (basic/layers.py:253-258) for one or more
    independent layers at once:

      groups == 1 : the weights of several layers that read the SAME input are stacked along N
                    (experts + gates of MMoE: one [K0, 148] product instead of nine);
      groups  > 1 : layer g reads columns [g*K, (g+1)*K) of x and writes [g*N, (g+1)*N) (per-domain towers).

    Training: batch statistics (biased variance) from the GEMM epilogue, running stats updated in place;
    eval: running statistics.  `bn = None` gives a plain Linear (+activation).
    """

    @staticmethod
    def forward(ctx, cfg, x, *params):
        # params = n_w weights [N_i, K] + n_w biases (or none) + (gammas + betas if bn)
        x_in = x
        nw = cfg["n_w"]
        Ws, rest = params[:nw], params[nw:]
        bs = rest[:nw] if cfg["has_bias"] else ()
        rest = rest[nw:] if cfg["has_bias"] else rest
        gammas, betas = (rest[:cfg["n_bn"]], rest[cfg["n_bn"]:]) if cfg["bn"] is not None else ((), ())
        H.require_device(x, Ws[0])
        oh_in = getattr(x_in, "_swr_onehot", None)
        if oh_in is None or oh_in.fl is None:
            x = H.f32c(x)                    # (a fused lookup hands over a zero-stride placeholder: nothing to make contiguous)
        W = _cat_params(Ws)
        b = _cat_params(bs) if bs else None
        G = cfg["groups"]
        M, K = x.shape[0], W.shape[1]
        Ntot = W.shape[0]
        N = Ntot // G
        dev = x.device
        training = cfg["training"] and cfg["bn"] is not None
        # fused lookup + gate-mix level: the level's [B, N] tensors (Z here; dY, dZ, the compact dX in the backward) get 128-byte
        # ALIGNED rows (pitch a multiple of 32 floats).  At a 592-byte pitch a cache line belongs to two 64-byte k-groups that a
        # product touches a chunk apart -- with ~15 MB of rows in flight per XCD the 4 MB L2 had dropped the line in between
        # (fl_dx: HBM reads 1.6 x, writes 1.3 x the tensors' bytes) -- and the epilogues' 128-byte row segments straddle lines
        pad_rows = (PAD_ROWS and oh_in is not None and oh_in.fold and oh_in.fl is not None and Ntot <= 160 and training
                    and cfg.get("mix") is not None and G == 1)
        ldz = (Ntot + 31) // 32 * 32 if pad_rows else Ntot
        Z = torch.empty((M, ldz), dtype=torch.float32, device=dev)
        if ldz != Ntot:
            Z = Z[:, :Ntot]
        n_tiles = (M + 31) // 32
        partials = torch.empty((n_tiles, Ntot, 2), dtype=torch.float32, device=dev) if training else None
        planes = planes_t = None
        ctx.wt_sel = None
        ctx.fl_fused = False
        # (weights pre-split into bf16 planes once per step -- ops.split_weights + gemm(B_split=...) -- were measured 16 us
        # per step SLOWER at config 2 than the split inside every workgroup; the entry points stay, the layers do not use them)
        if oh_in is not None and oh_in.fold:
            # the lookup wrote [E_big | dense | one-hot] only (OneHotInfo / include/swr.h "folded first layer"): multiply
            # that with the folded weights [W_big | W_dense | P], P_t[:, v] = W_t emb_t[v]
            if G != 1 or K != oh_in.K or not oh_tables(oh_in):
                raise H.SwrError("a lookup made with onehot=True must feed ONE ungrouped Linear over all its columns")
            Kf = oh_in.Kp + oh_in.oh_width
            tabs = (H.OnehotTable * len(oh_in.tables_p))()
            for j, (p_t, vocab, dim, off, col) in enumerate(oh_in.tables_p):
                tabs[j] = H.OnehotTable(p_t.data_ptr(), vocab, dim, off, col)
            # (the same launch transposes the columns of W the backward's dX product multiplies with: no W^T copy on the
            # forward-time fork, no cross-stream edge in front of dX)
            want_t = bool(ctx.needs_input_grad[1]) and oh_in.n_sel > 0 and Ntot % 4 == 0
            Wt_sel = torch.empty((oh_in.n_sel, Ntot), dtype=torch.float32, device=dev) if want_t else None
            ctx.wt_sel = Wt_sel
            if oh_in.fl is not None and Ntot <= 160 and lib.swr_gemm_precision_mode() == 1:
                # fused lookup (csrc/first_layer.hip): the product fetches its A operand through the row keys -- one
                # parameter-sized launch (folded weights in fragment order, the small tables' bf16-term shadows, W^T rows),
                # then the product
                f = oh_in.fl
                f["plan"].N = Ntot
                H.check(lib.swr_fl_prep(C.byref(f["plan"]), H.ptr(W), W.stride(0), K, H.ptr(oh_in.ohtab), tabs, len(oh_in.tables_p),
                                        H.ptr(oh_in.sel) if want_t else None, oh_in.n_sel if want_t else 0, H.ptr(Wt_sel), Ntot,
                                        H.ptr(f["ws"]), H.stream()), "swr_fl_prep")
                H.check(lib.swr_fl_fwd(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(b), H.ptr(Z), Z.stride(0), H.ptr(partials), H.stream()),
                        "swr_fl_fwd")
                _flush_deferred()      # the forward-time fork (the large tables' sort) is enqueued behind the product, as gemm() does
                ctx.fl_fused = True
            else:
                x = (oh_in.materialize() if oh_in.fl is not None else oh_in.wide)[:, oh_in.col0:oh_in.col0 + Kf]
                # (launching this fold on the side stream beside the lookup -- it depends on parameters only -- was measured: the
                # extra fork / join pair costs more than the 6 us it hides, 0.4988 vs 0.4937 ms per step)
                Wf = torch.empty((Ntot, Kf), dtype=torch.float32, device=dev)
                H.check(lib.swr_fold_first_layer_fwd(H.ptr(W), W.stride(0), Ntot, K, oh_in.Kp, oh_in.oh_width, H.ptr(oh_in.src),
                                                     H.ptr(oh_in.inv), H.ptr(oh_in.ohtab), tabs, len(oh_in.tables_p), H.ptr(Wf), Kf,
                                                     H.ptr(oh_in.sel) if want_t else None, oh_in.n_sel if want_t else 0,
                                                     H.ptr(Wt_sel), Ntot, H.stream()),
                        "swr_fold_first_layer_fwd")
                gemm("nt", x, Wf, Z, M, N, Kf, bias=b, stat_partials=partials, a_exact_from=oh_in.Kp)
            planes_t = None
            epi_act = 0
        else:
            # a layer without BatchNorm whose columns all take the same ReLU / sigmoid: applied while the product is stored
            na = _norm_acts(cfg["acts"], Ntot)
            epi_act = 0
            if (cfg["bn"] is None and na and na[0][2] in ("relu", "sigmoid")
                    and all(r[2] == na[0][2] for r in na)
                    and sorted((r[0], r[1]) for r in na)[0][0] == 0
                    and all(a_[1] == b_[0] for a_, b_ in zip(sorted((r[0], r[1]) for r in na), sorted((r[0], r[1]) for r in na)[1:]))
                    and max(r[1] for r in na) == Ntot):
                epi_act = 1 if na[0][2] == "relu" else 2
            gemm("nt", x, W, Z, M, N, K, bias=b, stat_partials=partials, groups=G,
                 gsA=(K if G > 1 else 0), gsB=N * K, gsC=N, gsBias=N, B_split=planes, c_act=epi_act)
        ctx.planes_t = planes_t
        acts, n_acts = H.act_ranges(cfg["acts"], Ntot)
        mean = rstd = scale = shift = None
        if cfg["bn"] is not None:
            bn = cfg["bn"]
            gamma, beta = _cat_params(gammas), _cat_params(betas)
            scale = torch.empty(Ntot, dtype=torch.float32, device=dev)
            shift = torch.empty(Ntot, dtype=torch.float32, device=dev)
            if training:
                if M < 2:
                    raise ValueError("Expected more than 1 value per channel when training")   # torch's message
                mean, rstd, scale, shift = _bn_train_finalize(bn, gamma, beta, partials, n_tiles, M, Ntot, dev)
            else:
                rm, rv = _cat_params(bn["running_mean"]), _cat_params(bn["running_var"])
                H.check(lib.swr_bn_eval_coeffs(H.ptr(gamma), H.ptr(beta), H.ptr(rm), H.ptr(rv), bn["eps"], Ntot,
                                               H.ptr(scale), H.ptr(shift), H.stream()), "swr_bn_eval_coeffs")
        identity = cfg["bn"] is None and all(a[2] in (None, "none") for a in _norm_acts(cfg["acts"], Ntot))
        mix = cfg.get("mix") if training else None
        if mix is not None:
            # BN + ReLU / softmax + gate mix in one pass (csrc/bnmix.hip): the activations Y are never materialised
            ne, Hm, D = mix
            Y = None
            out = torch.empty((M, D * Hm), dtype=torch.float32, device=dev)
            a = H.BnMixArgs()
            a.M, a.ne, a.H, a.D = M, ne, Hm, D
            a.Z, a.ldz, a.scale, a.shift = Z.data_ptr(), Z.stride(0), scale.data_ptr(), shift.data_ptr()
            a.P, a.ldp = out.data_ptr(), D * Hm
            gate_p = torch.empty((M, D * ne), dtype=torch.float32, device=dev)     # kept for the backward (tiny)
            a.G = gate_p.data_ptr()
            H.check(lib.swr_bnmix_fwd(C.byref(a), H.stream()), "swr_bnmix_fwd")
        elif identity or epi_act:
            out = Y = Z                      # (epi_act: Z already holds the activated values; the backward reads only Y)
        else:
            out = Y = torch.empty_like(Z)
            H.check(lib.swr_affine_act_fwd(H.ptr(Z), Z.stride(0), H.ptr(scale), H.ptr(shift), acts, n_acts, H.ptr(Y), Ntot, M,
                                           Ntot, H.stream()), "swr_affine_act_fwd")
        ctx.cfg, ctx.dims = cfg, (M, N, K, G, Ntot)
        ctx.grad_cols = getattr(x_in, "_swr_grad_cols", None)
        ctx.grad_dst = getattr(x_in, "_swr_grad_dst", None) if x.data_ptr() == x_in.data_ptr() else None
        oh = oh_in
        ctx.onehot = oh if (oh is not None and G == 1 and (oh.fold or (x.data_ptr() == x_in.data_ptr()
                                                                        and x.stride(0) == oh.oh_col + oh.oh_width))) else None
        ctx.params = params
        ctx.training_bn = training
        ctx.mix = mix
        ctx.save_for_backward(x, W, Z, Y, mean, rstd, scale, _cat_params(gammas) if gammas else None, shift if mix else None,
                              gate_p if mix else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dY):
        cfg = ctx.cfg
        M, N, K, G, Ntot = ctx.dims
        x, W, Z, Y, mean, rstd, scale, gamma, shift, gate_p = ctx.saved_tensors
        dev = x.device
        dY = H.f32c(dY)
        acts, n_acts = H.act_ranges(cfg["acts"], Ntot)
        nw = cfg["n_w"]
        p_W = ctx.params[:nw]
        p_b = ctx.params[nw:2 * nw] if cfg["has_bias"] else ()
        off = nw * (2 if cfg["has_bias"] else 1)
        p_g = ctx.params[off:off + cfg["n_bn"]] if cfg["bn"] is not None else ()
        p_be = ctx.params[off + cfg["n_bn"]:off + 2 * cfg["n_bn"]] if cfg["bn"] is not None else ()
        dgamma = dbeta = None
        direct_bn = False
        bn_dx = None
        ldp = Z.stride(0) if M > 1 else Ntot                # (the forward's row pitch: padded at a fused gate-mix level)
        dZ = torch.empty((M, ldp), dtype=torch.float32, device=dev)
        if ldp != Ntot:
            dZ = dZ[:, :Ntot]
        if ctx.training_bn:
            tile = lib.swr_bnmix_tile_rows() if ctx.mix is not None else 64
            nt = (M + tile - 1) // tile
            partials = torch.empty((nt, Ntot, 2), dtype=torch.float32, device=dev)
            if ctx.mix is not None:
                # incoming gradient is dP (w.r.t. the pooled outputs): one pass gives dL/d(BN output) + the statistics
                ne, Hm, D = ctx.mix
                dP = dY if dY.stride(0) % 4 == 0 and dY.data_ptr() % 16 == 0 else dY.contiguous()
                dY = torch.empty((M, ldp), dtype=torch.float32, device=dev)
                if ldp != Ntot:
                    dY = dY[:, :Ntot]
                a = H.BnMixArgs()
                a.M, a.ne, a.H, a.D = M, ne, Hm, D
                a.Z, a.ldz, a.scale, a.shift = Z.data_ptr(), ldp, scale.data_ptr(), shift.data_ptr()
                a.dP, a.lddp = dP.data_ptr(), dP.stride(0)
                a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
                a.dY, a.lddy, a.bn_partials = dY.data_ptr(), ldp, partials.data_ptr()
                a.G = gate_p.data_ptr()
                H.check(lib.swr_bnmix_bwd(C.byref(a), H.stream()), "swr_bnmix_bwd")
                acts, n_acts = H.act_ranges(None, Ntot)          # the activations are already differentiated
                Y = Z                                            # placeholder operand (no activation reads it)
            else:
                H.check(lib.swr_bn_act_bwd_stats(H.ptr(dY), dY.stride(0), H.ptr(Y), Ntot, H.ptr(Z), ldp, H.ptr(mean),
                                                 H.ptr(rstd), acts, n_acts, H.ptr(partials), M, Ntot, H.stream()),
                        "swr_bn_act_bwd_stats")
            dgamma, dbeta = _grad_alias(p_g, 2), _grad_alias(p_be, 2)
            direct_bn = dgamma is not None and dbeta is not None
            if not direct_bn:
                dgamma = torch.empty(Ntot, dtype=torch.float32, device=dev)
                dbeta = torch.empty(Ntot, dtype=torch.float32, device=dev)
            ca, cb, cc = (torch.empty(Ntot, dtype=torch.float32, device=dev) for _ in range(3))
            H.check(lib.swr_bn_bwd_finalize(H.ptr(partials), nt, M, Ntot, H.ptr(gamma), H.ptr(rstd), H.ptr(dgamma),
                                            H.ptr(dbeta), int(direct_bn), H.ptr(ca), H.ptr(cb), H.ptr(cc), H.stream()),
                    "swr_bn_bwd_finalize")
            # fused lookup, gate-mix level (no activation left to differentiate), the weight-gradient product forked / held back
            # behind dX: dZ is produced INSIDE the dX product (swr_bn_bwd_dx below) -- no pass of its own
            oh0 = ctx.onehot
            n_dw = 2.0 * M * Ntot * K
            fuse_dx = (FUSE_BN_DX and ctx.fl_fused and ctx.mix is not None and oh0 is not None and oh0.n_sel > 0
                       and ctx.needs_input_grad[1] and getattr(ctx, "wt_sel", None) is not None and dY.stride(0) % 4 == 0
                       and lib.swr_bn_bwd_dx_supported(Ntot, oh0.n_sel)
                       and ((SIDE_STREAM and SIDE_DW_MIN_FLOP <= n_dw < SIDE_DW_MAX_FLOP) or _late["on"] or FUSE_BN_DX_SINGLE))
            if fuse_dx:
                bn_dx = (dY, ca, cb, cc)
            else:
                H.check(lib.swr_act_bwd_apply(H.ptr(dY), dY.stride(0), H.ptr(Y), Y.stride(0) if M > 1 else Ntot, H.ptr(Z), ldp, H.ptr(ca),
                                              H.ptr(cb), H.ptr(cc), H.ptr(mean), acts, n_acts, H.ptr(dZ), ldp, M, Ntot, H.stream()),
                        "swr_act_bwd_apply")
        else:
            # eval-mode BN (a fixed affine) or no BN: dZ = scale * act'(Y) dY
            identity = scale is None and all(a[2] in (None, "none") for a in _norm_acts(cfg["acts"], Ntot))
            if identity:
                dZ = dY if (M <= 1 or dY.stride(0) == Ntot) else dY.contiguous()
            else:
                H.check(lib.swr_act_bwd_apply(H.ptr(dY), dY.stride(0), H.ptr(Y), Y.stride(0) if M > 1 else Ntot, H.ptr(Z), ldp, H.ptr(scale), None,
                                              None, None, acts, n_acts, H.ptr(dZ), ldp, M, Ntot, H.stream()),
                        "swr_act_bwd_apply")
        # parameter gradients: dW[g] = dZ_g^T x_g (+ db = column sums), dX = dZ W.  When the layer's parameters sit
        # in the arena the kernels accumulate into the gradient arena directly (it was zeroed by zero_grad).
        dW = _grad_alias(p_W)
        db = _grad_alias(p_b) if cfg["has_bias"] else None
        direct_w = dW is not None and (db is not None or not cfg["has_bias"])
        if not direct_w:
            dW = torch.empty((Ntot, K), dtype=torch.float32, device=dev)
            db = torch.empty(Ntot, dtype=torch.float32, device=dev) if cfg["has_bias"] else None
        oh = ctx.onehot if (ctx.onehot is not None and direct_w and ctx.needs_input_grad[1] and oh_ready(ctx.onehot)) else None
        if ctx.onehot is not None and ctx.onehot.fold and oh is None:
            raise H.SwrError("folded first layer: the layer's and the small tables' gradients must live in the gradient arena "
                             "(SwrModule.build_arena) and the lookup must take a gradient")

        # dZ of a fused BN-backward + dX launch is read by ONE consumer, the weight-gradient product: where that product can
        # recompute it from dY and Z while staging (swr_fl_dw_bn) it is never written -- the dX launch moves 122 MB instead of 164
        dz_free = bool(DZ_FREE and bn_dx is not None and oh is not None and oh.fold and ctx.fl_fused
                       and lib.swr_fl_dw_supported(C.byref(oh.fl["plan"]), dZ.stride(0))
                       and lib.swr_fl_dw_bn_supported(C.byref(oh.fl["plan"]), bn_dx[0].stride(0), ldp))

        def launch_dw():
            if oh is not None and oh.fold:
                # dWp = dZ^T [E_big | dense | one-hot]; unfolded into dW / db (arena), the small tables' gradients from S
                Kf = oh.Kp + oh.oh_width
                dWp = torch.empty((Ntot, Kf), dtype=torch.float32, device=dev)
                dbp = torch.empty(Ntot, dtype=torch.float32, device=dev) if cfg["has_bias"] else None
                if dz_free:
                    f = oh.fl
                    dYb, ca_, cb_, cc_ = bn_dx
                    nb = lib.swr_fl_dw_workspace_bytes(C.byref(f["plan"]))
                    wsd = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
                    H.check(lib.swr_fl_dw_bn(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(dYb), dYb.stride(0), H.ptr(Z), ldp, H.ptr(ca_),
                                             H.ptr(cb_), H.ptr(cc_), H.ptr(mean), H.ptr(dWp), Kf, H.ptr(dbp), H.ptr(wsd), nb, H.stream()),
                            "swr_fl_dw_bn")
                elif ctx.fl_fused and lib.swr_fl_dw_supported(C.byref(oh.fl["plan"]), dZ.stride(0)):
                    # the product's staging threads fetch the table rows through the keys: A' is never written
                    f = oh.fl
                    nb = lib.swr_fl_dw_workspace_bytes(C.byref(f["plan"]))
                    wsd = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
                    H.check(lib.swr_fl_dw(C.byref(f["plan"]), H.ptr(f["ws"]), H.ptr(dZ), dZ.stride(0), H.ptr(dWp), Kf, H.ptr(dbp),
                                          H.ptr(wsd), nb, H.stream()), "swr_fl_dw")
                else:
                    xa = x
                    if ctx.fl_fused:  # a small batch: the gather launch writes A' now, on this (the weight-gradient) branch
                        xa = oh.materialize()[:, oh.col0:oh.col0 + Kf]
                    gemm_tn(dZ, xa, dWp, M, N, Kf, colsum=dbp)
                tw = (H.OnehotTable * len(oh.tables_p))()
                tg = (H.OnehotTable * len(oh.tables))()
                for j, ((p_t, vocab, dim, off, col), (g_t, *_r)) in enumerate(zip(oh.tables_p, oh.tables)):
                    tw[j] = H.OnehotTable(p_t.data_ptr(), vocab, dim, off, col)
                    tg[j] = H.OnehotTable(g_t.data_ptr(), vocab, dim, oh.Kp + off, col)
                if FOLD_BWD_ONE_LAUNCH and 0 < len(oh.tables) <= 64:
                    H.check(lib.swr_fold_first_layer_bwd_tables(H.ptr(dWp), Kf, H.ptr(dbp), Ntot, K, oh.Kp, oh.oh_width, H.ptr(oh.src),
                                                                H.ptr(oh.inv), tw, len(oh.tables_p), H.ptr(dW), K, H.ptr(db), 1,
                                                                H.ptr(W), W.stride(0), tg, len(oh.tables), H.stream()),
                            "swr_fold_first_layer_bwd_tables")
                    return
                H.check(lib.swr_fold_first_layer_bwd(H.ptr(dWp), Kf, H.ptr(dbp), Ntot, K, oh.Kp, oh.oh_width, H.ptr(oh.src),
                                                     H.ptr(oh.inv), tw, len(oh.tables_p), H.ptr(dW), K, H.ptr(db), 1, H.stream()),
                        "swr_fold_first_layer_bwd")
                H.check(lib.swr_onehot_table_grads(H.ptr(dWp), Kf, H.ptr(W), W.stride(0), Ntot, tg, len(oh.tables), 1,
                                                   H.stream()), "swr_onehot_table_grads")
                return
            if oh is not None:
                # dZ^T [E | 0 | one-hot]: dW into the arena, the segment sums of dZ per small-table row into S, then the small
                # tables' gradients S_t W_t straight into the arena (OneHotInfo)
                K2 = oh.oh_col + oh.oh_width
                S = torch.empty((Ntot, K2 - K), dtype=torch.float32, device=dev)
                gemm_tn(dZ, x, dW, M, N, K2, colsum=db, accumulate=True, ldc=K, C2=S, c2_from=K)
                tabs = (H.OnehotTable * len(oh.tables))()
                for j, (g_t, vocab, dim, off, col) in enumerate(oh.tables):
                    tabs[j] = H.OnehotTable(g_t.data_ptr(), vocab, dim, oh.oh_col - K + off, col)
                H.check(lib.swr_onehot_table_grads(H.ptr(S), K2 - K, H.ptr(W), W.stride(0), Ntot, tabs, len(oh.tables), 1,
                                                   H.stream()), "swr_onehot_table_grads")
                return
            gemm_tn(dZ, x, dW, M, N, K, colsum=db, accumulate=direct_w, groups=G, gsA=N, gsB=(K if G > 1 else 0),
                    gsC=N * K, gsColsum=N, ldc=K)
        # (a fork / join pair costs ~10-20 us of edges: only a product worth several of those goes to its own stream --
        # measured: forking every small product LOSES 0.02 ms at configs 1, 3 and 4)
        side_dw = direct_w and SIDE_STREAM and SIDE_DW_MIN_FLOP <= 2.0 * M * Ntot * K < SIDE_DW_MAX_FLOP
        late_dw = direct_w and _late["on"] and ctx.needs_input_grad[1] and not side_dw
        # single-stream step (short batches) with the BatchNorm backward inside the dX product: dX first, the weight gradient --
        # which recomputes dZ from dY and Z, or reads the dZ that launch wrote -- right behind it on the same stream
        after_dx = bn_dx is not None and not side_dw and not late_dw
        if late_dw:
            _late["jobs"].append(launch_dw)       # split backward: dX first, this product after the row lists are out
        elif not side_dw and not after_dx:
            launch_dw()
        dx = None
        if oh is not None:
            # dX only for the columns of the tables that go through K3, compact; the lookup's backward picks it up from its
            # ctx, autograd carries a zero-stride placeholder of the right shape
            if oh.n_sel > 0:
                lds_ = (oh.n_sel + 31) // 32 * 32 if ldp != Ntot else oh.n_sel
                dsel = torch.empty((M, lds_), dtype=torch.float32, device=dev)
                if lds_ != oh.n_sel:
                    dsel = dsel[:, :oh.n_sel]
                wt = getattr(ctx, "wt_sel", None)
                if bn_dx is not None:
                    dYb, ca_, cb_, cc_ = bn_dx
                    H.check(lib.swr_bn_bwd_dx(C.byref(oh.fl["plan"]), H.ptr(oh.fl["ws"]), H.ptr(dYb), dYb.stride(0), H.ptr(Z), ldp,
                                              H.ptr(ca_), H.ptr(cb_), H.ptr(cc_), H.ptr(mean), oh.n_sel, None if dz_free else H.ptr(dZ), ldp,
                                              H.ptr(dsel), lds_, H.stream()), "swr_bn_bwd_dx")
                    _stamp("m_dx_end")
                    if _side["deferred"]:
                        _flush_deferred()
                else:
                    gemm("nt", dZ, wt if wt is not None else _selected_wt(W, oh.sel), dsel, M, oh.n_sel, Ntot)
            else:
                dsel = dZ
            oh.ctx.fused_dx = (dsel, oh.compact)
            _mark_touched(oh.params)
            dx = _zero_scalar(dev).expand(M, K)
        elif ctx.needs_input_grad[1]:
            if G > 1:
                dx = _grad_dst_view(ctx.grad_dst, M, G * K, dev)           # a block of a split_cols gradient, or
                if dx is None:
                    dx = torch.empty((M, G * K), dtype=torch.float32, device=dev)
                if M >= 4096 and N % 4 == 0 and K >= 32 and K % 4 == 0 and lib.swr_gemm_precision_mode() == 1:
                    # the groups' transposed weights (one small launch) make dX an "nt" product: the [N, K] layout the bf16-split
                    # kernel stages, 2.7 x the matrix rate of the f32-MFMA kernel the "nn" form falls to
                    gemm("nt", dZ, _transposed_groups(W, G, direct_w), dx, M, K, N, groups=G, gsA=N, gsB=N * K, gsC=K)
                else:
                    gemm("nn", dZ, W, dx, M, K, N, groups=G, gsA=N, gsB=N * K, gsC=K)
            else:
                dx = torch.empty((M, _pad4(K)), dtype=torch.float32, device=dev)
                if Ntot % 4 == 0 and K >= 32:
                    # dX = dZ @ W as an "nt" product against the transposed weights (a small copy): the [N, K]
                    # layout is the one the bf16-split MFMA kernel stages into LDS
                    # (forking this 5 us copy onto the side stream at forward time was measured: the extra
                    # cross-stream edge costs ~12 us of main-stream latency, more than the copy)
                    # x = the embedding concat: its trailing dense-feature columns take no gradient (EmbeddingLayer
                    # marks the tensor), so whole column tiles of dX past them are skipped and stored as zeros
                    nc = ctx.grad_cols if (ctx.grad_cols is not None and 0 < ctx.grad_cols < K) else 0
                    if ctx.planes_t is not None:
                        gemm("nt", dZ, W, dx, M, K, Ntot, ldb=Ntot, B_split=ctx.planes_t, n_compute=nc)    # W^T only through its planes
                    else:
                        gemm("nt", dZ, _transposed_weight(W), dx, M, K, Ntot, n_compute=nc)
                else:
                    gemm("nn", dZ, W, dx, M, K, Ntot)
                if dx.shape[1] != K:
                    dx = dx[:, :K]
        if after_dx:
            launch_dw()
        if side_dw:
            # forked AFTER the dX product is enqueued: dX is on the critical path and must not share the MFMA pipes
            # with dW; dW then overlaps whatever the main stream does next (the embedding backward, lower layers).
            # (Measured, config 2: forking BEFORE dX instead -- dW next to dX and K3 -- 0.523 vs 0.520 ms: that stretch of
            # the step is throughput-bound, not dependency-bound.)
            _fork_dw(dev, launch_dw, (dZ, x, dW, db) + ((tuple(bn_dx) + (Z, mean)) if dz_free else ()))
        if direct_w:
            _mark_touched(p_W + tuple(p_b))
            grads = [None] * (nw * (2 if cfg["has_bias"] else 1))
        else:
            grads = list(_split_like(dW, p_W))
            if cfg["has_bias"]:
                grads += _split_like(db, p_b)
        if cfg["bn"] is not None:
            if direct_bn:
                _mark_touched(tuple(p_g) + tuple(p_be))
                grads += [None] * (2 * cfg["n_bn"])
            elif dgamma is not None:
                grads += _split_like(dgamma, p_g) + _split_like(dbeta, p_g)
            else:
                grads += [None] * (2 * cfg["n_bn"])
        return (None, dx) + tuple(grads)


_ZEROS = {}
_ZERO_IDX = {}


def _zero_index(dev):
    z = _ZERO_IDX.get(str(dev))
    if z is None:
        z = _ZERO_IDX[str(dev)] = torch.zeros(1, dtype=torch.int64, device=dev)
    return z


def _zero_scalar(dev):
    z = _ZEROS.get(str(dev))
    if z is None:
        z = _ZEROS[str(dev)] = torch.zeros(1, dtype=torch.float32, device=dev)
    return z


def oh_tables(oh):
    """Folded layout: the tables must be current fp32 tensors on the layer's device (always true for arena members)."""
    return all(p.dtype == torch.float32 and p.is_contiguous() for p, *_r in oh.tables_p)


def oh_ready(oh):
    """The one-hot tables' gradients live in the gradient arena right now (zero_grad re-points them every step)."""
    tabs = []
    for p, vocab, dim, off, col in oh.tables_p:
        g = _grad_alias([p], 4)
        if g is None:
            return False
        tabs.append((g, vocab, dim, off, col))
    oh.tables = tabs
    return True


def _pad4(n):
    return (n + 3) // 4 * 4


def _norm_acts(acts, n):
    if acts is None or isinstance(acts, str):
        return [(0, n, acts, 1)]
    return acts


def bnmix_supported(ne, Hm, D):
    return bool(lib.swr_bnmix_supported(int(ne), int(Hm), int(D)))


def linear_bn_act(x, weights, biases, bn=None, acts=None, groups=1, training=True, mix=None):
    """Functional front-end of LinearBNAct.

    weights / biases: lists of Parameters stacked along the output dim (biases may be None);
    bn: None or dict(gamma=[...], beta=[...], running_mean=[...], running_var=[...], nbt=[...], eps, momentum)."""
    cfg = {"n_w": len(weights), "has_bias": biases is not None, "groups": groups, "acts": acts,
           "training": training, "bn": None, "n_bn": 0, "mix": mix}
    params = list(weights) + (list(biases) if biases is not None else [])
    if bn is not None:
        cfg["bn"] = {k: bn[k] for k in ("running_mean", "running_var", "nbt", "eps", "momentum")}
        cfg["n_bn"] = len(bn["gamma"])
        params += list(bn["gamma"]) + list(bn["beta"])
    return LinearBNAct.apply(cfg, x, *params)


def _tower_forward_linear(cfg, x, params):
    """First tower layer + its BatchNorm statistics (swr_tower_fwd_linear, swr_bn_finalize); returns the launch arguments
    with scale / shift / w2 / b2 filled in for the head kernel, and the tensors the backward needs."""
    G = cfg["groups"]
    W1s, b1s, gammas, betas, w2s, b2s = (params[i * G:(i + 1) * G] for i in range(6))
    H.require_device(x, W1s[0])
    x = H.f32c(x)
    M = x.shape[0]
    Hd, K = W1s[0].shape
    N = G * Hd
    dev = x.device
    if M < 2:
        raise ValueError("Expected more than 1 value per channel when training")   # torch's message
    W1, b1 = _cat_params(W1s), _cat_params(b1s)
    gamma, beta = _cat_params(gammas), _cat_params(betas)
    w2, b2 = _cat_params([w.reshape(-1) for w in w2s]), _cat_params(b2s)
    Z1 = torch.empty((M, N), dtype=torch.float32, device=dev)
    n_tiles = (M + 31) // 32
    partials = torch.empty((n_tiles, N, 2), dtype=torch.float32, device=dev)
    a = H.TowerArgs()
    a.M, a.G, a.K, a.H = M, G, K, Hd
    a.X, a.ldx = x.data_ptr(), x.stride(0)
    a.W1, a.b1 = W1.data_ptr(), b1.data_ptr()
    a.Z1, a.ldz = Z1.data_ptr(), N
    a.stat_partials = partials.data_ptr()
    H.check(lib.swr_tower_fwd_linear(C.byref(a), H.stream()), "swr_tower_fwd_linear")
    mean, rstd, scale, shift = _bn_train_finalize(cfg["bn"], gamma, beta, partials, n_tiles, M, N, dev)
    a.scale, a.shift = scale.data_ptr(), shift.data_ptr()
    a.w2, a.b2 = w2.data_ptr(), b2.data_ptr()
    return a, (M, G, K, Hd), (x, W1, Z1, mean, rstd, scale, shift, gamma, w2), b2


def _tower_backward(ctx, saved, dV=None, sel=None):
    """swr_tower_bwd + the grouped weight-gradient product of the first layer.  `dV` [M, G], or `sel` = (domain, y, p, dloss):
    the gradient implied by the fused select + BCE, computed inside the kernels (include/swr.h "selected mode")."""
    x, W1, Z1, mean, rstd, scale, shift, gamma, w2 = saved
    M, G, K, Hd = ctx.dims
    N = G * Hd
    dev = x.device
    p_W1, p_b1, p_g, p_be, p_w2, p_b2 = (ctx.params[i * G:(i + 1) * G] for i in range(6))
    dgamma, dbeta = _grad_alias(p_g, 2), _grad_alias(p_be, 2)
    dw2, db2 = _grad_alias(p_w2), _grad_alias(p_b2)
    direct = all(t is not None for t in (dgamma, dbeta, dw2, db2))
    if not direct:
        dgamma, dbeta, dw2 = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
        db2 = torch.empty(G, dtype=torch.float32, device=dev)
    ca, cb, cc = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
    dZ1 = torch.empty((M, N), dtype=torch.float32, device=dev)
    need_dx = ctx.needs_input_grad[1]
    dx = torch.empty((M, G * K), dtype=torch.float32, device=dev) if need_dx else None
    a = H.TowerArgs()
    a.M, a.G, a.K, a.H, a.accumulate = M, G, K, Hd, int(direct)
    a.W1 = W1.data_ptr()
    a.Z1, a.ldz = Z1.data_ptr(), N
    a.scale, a.shift, a.mean, a.rstd, a.gamma = (t.data_ptr() for t in (scale, shift, mean, rstd, gamma))
    a.w2 = w2.data_ptr()
    if dV is not None:
        a.dV, a.lddv = dV.data_ptr(), G
    else:
        domain, y, p, dloss = sel
        a.sel_domain, a.sel_dom_dtype = domain.data_ptr(), H.dtype_code(domain)
        a.sel_y, a.sel_y_dtype = y.data_ptr(), H.dtype_code(y)
        a.sel_p, a.sel_dloss = p.data_ptr(), dloss.data_ptr()
    a.ca, a.cb, a.cc = ca.data_ptr(), cb.data_ptr(), cc.data_ptr()
    a.dgamma, a.dbeta, a.dw2, a.db2 = dgamma.data_ptr(), dbeta.data_ptr(), dw2.data_ptr(), db2.data_ptr()
    a.dZ1, a.lddz = dZ1.data_ptr(), N
    a.dX, a.lddx = (dx.data_ptr(), G * K) if need_dx else (None, 0)
    nbytes = lib.swr_tower_bwd_workspace_bytes(M, G, Hd)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    H.check(lib.swr_tower_bwd(C.byref(a), H.ptr(ws), nbytes, H.stream()), "swr_tower_bwd")
    # first-layer weights: the ordinary grouped weight-gradient product on dZ1, x
    dW1, db1 = _grad_alias(p_W1), _grad_alias(p_b1)
    direct_w = dW1 is not None and db1 is not None
    if not direct_w:
        dW1 = torch.empty((N, K), dtype=torch.float32, device=dev)
        db1 = torch.empty(N, dtype=torch.float32, device=dev)
    # one pass over dZ1 and x for all towers (csrc/tower.hip tower_dw_kernel) where the shape is built, else the grouped product
    one_pass = (TOWER_DW and M > 0 and lib.swr_tower_dw_supported(K, Hd, G) and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0 and dW1.is_contiguous() and db1.is_contiguous())
    def launch_dw1():
        if one_pass:
            nb = lib.swr_tower_dw_workspace_bytes(M, K, Hd, G)
            wsd = torch.empty(nb, dtype=torch.uint8, device=dev)
            H.check(lib.swr_tower_dw(H.ptr(dZ1), N, H.ptr(x), x.stride(0), M, K, Hd, G, H.ptr(dW1), H.ptr(db1), int(direct_w),
                                     H.ptr(wsd), nb, H.stream()), "swr_tower_dw")
            return
        gemm_tn(dZ1, x, dW1, M, Hd, K, colsum=db1, accumulate=direct_w, groups=G, gsA=Hd, gsB=K, gsC=Hd * K, gsColsum=Hd,
                ldc=K)
    if direct_w and SIDE_STREAM and (not _late["on"] or DP_RIDE) and _in_backward():
        _ride_dw(launch_dw1, (dZ1, x, dW1, db1))
    elif direct_w and _late["on"]:
        # split backward (data-parallel step): nothing but the optimizer reads it -- with the other weight-gradient work,
        # behind the row lists (it sat on the critical path in front of the expert level's backward: 25 us at config 2)
        _late["jobs"].append(launch_dw1)
    else:
        launch_dw1()
    grads = []
    if direct_w:
        _mark_touched(p_W1 + p_b1)
        grads += [None] * (2 * G)
    else:
        grads += list(_split_like(dW1, p_W1)) + list(_split_like(db1, p_b1))
    if direct:
        _mark_touched(p_g + p_be + p_w2 + p_b2)
        grads += [None] * (4 * G)
    else:
        grads += list(_split_like(dgamma, p_g)) + list(_split_like(dbeta, p_be))
        grads += [t.reshape(p.shape) for t, p in zip(_split_like(dw2, [w.reshape(-1) for w in p_w2]), p_w2)]
        grads += list(_split_like(db2, p_b2))
    return dx, tuple(grads)


class TowerHead(Function):
    """G per-domain towers [Linear(K, H) -> BatchNorm1d(H) -> ReLU -> Linear(H, 1)] on their own K-column blocks of x, in
    training mode (batch statistics): mmoe.py:38-41,50-51 with tower_params = {"dims": [H]}.  Three launches forward,
    three + the grouped weight-gradient product backward (csrc/tower.hip); A1 = relu(bn(Z1)) is never stored.

    params = G first-layer weights [H, K] + G biases + G gammas + G betas + G output weights [1, H] + G output biases."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        a, dims, saved, _b2 = _tower_forward_linear(cfg, x, params)
        M, G = dims[0], dims[1]
        V = torch.empty((M, G), dtype=torch.float32, device=saved[0].device)
        a.V, a.ldv = V.data_ptr(), G
        H.check(lib.swr_tower_fwd_head(C.byref(a), H.stream()), "swr_tower_fwd_head")
        ctx.cfg, ctx.dims, ctx.params = cfg, dims, params
        ctx.save_for_backward(*saved)
        return V

    @staticmethod
    @once_differentiable
    def backward(ctx, dV):
        dx, grads = _tower_backward(ctx, ctx.saved_tensors, dV=H.f32c(dV).contiguous())
        return (None, dx) + grads


class TowerHeadSelectBCE(Function):
    """TowerHead + sigmoid + domain select (mmoe.py:51-55) + the trainer's mean BCE (ctr_trainer.py:56,70): returns
    (p [M], loss).  A row evaluates only the output layer of its own domain's tower (swr_tower_head_select_bce_fwd: one
    launch instead of head + select/BCE), and the backward kernels form dV from (p, y, domain) on the fly (no select/BCE
    backward launch, no dV).  Same bits as TowerHead -> SelectBCE.  A gradient arriving on p as well (someone used the
    probabilities) goes through the explicit dV path."""

    @staticmethod
    def forward(ctx, cfg, x, domain, y, *params):
        a, dims, saved, b2 = _tower_forward_linear(cfg, x, params)
        M, G, _K, Hd = dims
        dev = saved[0].device
        H.require_device(domain, y)
        domain, y = domain.contiguous(), y.contiguous()
        p = torch.empty(M, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        nbytes = lib.swr_bce_workspace_bytes(M)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        Z1, scale, shift, w2 = saved[2], saved[5], saved[6], saved[8]
        adv_hyper, adv_hist, adv_cap = take_loss_rider()      # the optimizer's step bookkeeping, if it asked for a ride
        H.check(lib.swr_tower_head_select_bce_fwd_adv(H.ptr(Z1), G * Hd, G, Hd, H.ptr(scale), H.ptr(shift), H.ptr(w2), H.ptr(b2),
                                                      H.ptr(domain), H.dtype_code(domain), H.ptr(y), H.dtype_code(y), M, H.ptr(p),
                                                      H.ptr(loss), H.ptr(ws), nbytes, H.ptr(H.ticket(dev)), H.ptr(adv_hyper),
                                                      H.ptr(adv_hist), adv_cap, H.stream()),
                "swr_tower_head_select_bce_fwd_adv")
        ctx.cfg, ctx.dims, ctx.params = cfg, dims, params
        ctx.save_for_backward(*saved, p, domain, y)
        ctx.set_materialize_grads(False)
        return p, loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dp, dloss):
        saved, (p, domain, y) = ctx.saved_tensors[:-3], ctx.saved_tensors[-3:]
        M, G = ctx.dims[0], ctx.dims[1]
        if dloss is None and dp is None:
            return (None,) * (4 + len(ctx.params))
        if dloss is None:
            dloss = torch.zeros((), dtype=torch.float32, device=p.device)
        dloss = dloss.float().contiguous()
        if dp is None:
            dx, grads = _tower_backward(ctx, saved, sel=(domain, y, p, dloss))
        else:
            dV = torch.empty((M, G), dtype=torch.float32, device=p.device)
            H.check(lib.swr_select_bce_bwd(H.ptr(p), H.ptr(y), H.dtype_code(y), G, H.ptr(domain), H.dtype_code(domain), M,
                                           H.ptr(dloss), H.ptr(dV), G, H.stream()), "swr_select_bce_bwd")
            extra = torch.empty((M, G), dtype=torch.float32, device=p.device)
            dp = H.f32c(dp).contiguous()
            H.check(lib.swr_select_bwd(H.ptr(dp), H.ptr(p), G, H.ptr(domain), H.dtype_code(domain), 1, 0, H.ptr(extra), G,
                                       None, M, H.stream()), "swr_select_bwd")
            dx, grads = _tower_backward(ctx, saved, dV=dV + extra)
        return (None, dx, None, None) + grads


def tower_head_supported(K, Hd):
    return bool(lib.swr_tower_supported(int(K), int(Hd)))


def tower_head(x, W1s, b1s, bn, w2s, b2s):
    """Functional front-end of TowerHead; `bn` as in linear_bn_act."""
    cfg = {"groups": len(W1s), "bn": {k: bn[k] for k in ("running_mean", "running_var", "nbt", "eps", "momentum")}}
    return TowerHead.apply(cfg, x, *W1s, *b1s, *bn["gamma"], *bn["beta"], *w2s, *b2s)


TOWER_SELECT = os.environ.get("SWR_TOWER_SELECT", "1") != "0"
# the small products that ride the weight-gradient branch (the towers' dW) run IN FRONT of the first layer's product: that product
# holds every CU for ~45 us, and the embedding backward on the main stream can only start beside the riders -- behind the product
# it waited for it (stamps: embedding backward begins 8 us after dX instead of 47; 0.4008 -> 0.3870 ms per step)
RIDERS_FIRST = os.environ.get("SWR_RIDERS_FIRST", "1") != "0"
DZ_FREE = os.environ.get("SWR_DZ_FREE", "1") != "0"      # dZ recomputed inside the weight-gradient product (swr_fl_dw_bn): never written
FOLD_BWD_ONE_LAUNCH = os.environ.get("SWR_FOLD_BWD_ONE_LAUNCH", "1") != "0"   # unfolding of dWp + the small tables' gradients in one launch
DP_RIDE = os.environ.get("SWR_DP_RIDE", "1") != "0"           # split backward: the towers' dW rides the weight-gradient branch too
TOWER_DW = os.environ.get("SWR_TOWER_DW", "1") != "0"         # the towers' first-layer weight gradients in one pass (swr_tower_dw)


def tower_head_select(x, W1s, b1s, bn, w2s, b2s, domain):
    """`domain_select(tower_head(...), domain)`.  Inside `fused_bce(y)` (the trainer's step) head + select + BCE are one
    launch (TowerHeadSelectBCE) and the selected probabilities come back with the loss parked on the request."""
    req = fused_bce._active
    if TOWER_SELECT and req is not None and req.p is None and req.y.numel() == x.shape[0] and x.shape[0] > 0:
        cfg = {"groups": len(W1s), "bn": {k: bn[k] for k in ("running_mean", "running_var", "nbt", "eps", "momentum")}}
        req.p, req.loss = TowerHeadSelectBCE.apply(cfg, x, domain, req.y.reshape(-1), *W1s, *b1s, *bn["gamma"], *bn["beta"],
                                                   *w2s, *b2s)
        return req.p
    return domain_select(tower_head(x, W1s, b1s, bn, w2s, b2s), domain, apply_sigmoid=True)


class BatchStandardize(Function):
    """y = gamma * (x - mean_batch) / sqrt(var_batch + eps) + beta, biased variance, no running statistics (gamma, beta
    optional): the shared part of STAR's partitioned normalisation (star.py:91-98) and HAMUR's domain norm
    (hamur.py:188-196).  Statistics by swr_col_moments + swr_bn_finalize, one affine pass; the backward is the
    BatchNorm backward (two passes)."""

    @staticmethod
    def forward(ctx, x, eps, gamma, beta):
        H.require_device(x)
        x = H.f32c(x)
        M, N = x.shape
        if M < 1:
            raise ValueError("empty batch")
        dev = x.device
        n_tiles = (M + 31) // 32
        partials = torch.empty((n_tiles, N, 2), dtype=torch.float32, device=dev)
        H.check(lib.swr_col_moments(H.ptr(x), x.stride(0) if M > 1 else N, M, N, H.ptr(partials), H.stream()), "swr_col_moments")
        mean, rstd, scale, shift = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(4))
        g = H.f32c(gamma).contiguous() if gamma is not None else None
        b = H.f32c(beta).contiguous() if beta is not None else None
        H.check(lib.swr_bn_finalize(H.ptr(partials), n_tiles, M, N, H.ptr(g), H.ptr(b), float(eps), 0.0, None, None, None, 0,
                                    H.ptr(mean), H.ptr(rstd), H.ptr(scale), H.ptr(shift), H.stream()), "swr_bn_finalize")
        ldy = (N + 3) // 4 * 4
        y = torch.empty((M, ldy), dtype=torch.float32, device=dev)
        acts, n_acts = H.act_ranges(None, N)
        H.check(lib.swr_affine_act_fwd(H.ptr(x), x.stride(0) if M > 1 else N, H.ptr(scale), H.ptr(shift), acts, n_acts,
                                       H.ptr(y), ldy, M, N, H.stream()), "swr_affine_act_fwd")
        ctx.save_for_backward(x, mean, rstd, g)
        ctx.has_affine = (gamma is not None, beta is not None)
        return y[:, :N] if ldy != N else y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, mean, rstd, g = ctx.saved_tensors
        M, N = x.shape
        dev = x.device
        dy = H.f32c(dy)
        ldx = x.stride(0) if M > 1 else N
        acts, n_acts = H.act_ranges(None, N)
        nt = (M + 63) // 64
        partials = torch.empty((nt, N, 2), dtype=torch.float32, device=dev)
        H.check(lib.swr_bn_act_bwd_stats(H.ptr(dy), dy.stride(0) if M > 1 else N, H.ptr(x), ldx, H.ptr(x), ldx, H.ptr(mean),
                                         H.ptr(rstd), acts, n_acts, H.ptr(partials), M, N, H.stream()), "swr_bn_act_bwd_stats")
        ca, cb, cc = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
        dgamma = torch.empty(N, dtype=torch.float32, device=dev) if ctx.has_affine[0] else None
        dbeta = torch.empty(N, dtype=torch.float32, device=dev) if ctx.has_affine[1] else None
        H.check(lib.swr_bn_bwd_finalize(H.ptr(partials), nt, M, N, H.ptr(g), H.ptr(rstd), H.ptr(dgamma), H.ptr(dbeta), 0,
                                        H.ptr(ca), H.ptr(cb), H.ptr(cc), H.stream()), "swr_bn_bwd_finalize")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, N), dtype=torch.float32, device=dev)
            H.check(lib.swr_act_bwd_apply(H.ptr(dy), dy.stride(0) if M > 1 else N, H.ptr(x), ldx, H.ptr(x), ldx, H.ptr(ca),
                                          H.ptr(cb), H.ptr(cc), H.ptr(mean), acts, n_acts, H.ptr(dx), N, M, N, H.stream()),
                    "swr_act_bwd_apply")
        return dx, None, dgamma, dbeta


_rm_pending = []          # per-sample factors whose gradient is still being collected in this backward pass


def _rm_check():
    """End of a backward pass: every shared factor must have handed its collected gradient to autograd."""
    left = [t for t in _rm_pending if getattr(t, "_swr_rm", None) is not None]
    del _rm_pending[:]
    for t in left:
        t._swr_rm = None
        t._swr_rm_uses = 0
    if left:
        raise H.SwrError("RowMat: a per-sample factor feeds several products but not all of them took part in this backward "
                         "pass -- its gradient was being collected across them (set SWR_ROWMAT_SHARE=0)")


ROWMAT_SHARE = os.environ.get("SWR_ROWMAT_SHARE", "1") != "0"


class RowMat(Function):
    """out[b, d, :] = T[b, d, :] @ Hm[b]: the per-sample k x k factor of HAMUR's adapter (hamur.py:175-186).

    The SAME Hm feeds both products of an adapter cell: their backward passes add into ONE gradient buffer
    (`accumulate_dhm`) and only the last of them hands it to autograd -- the engine would otherwise sum two [B, k, k]
    tensors with an extra pass (74 us for 160 MB at config 5).  The uses are counted on the Hm object in forward."""

    @staticmethod
    def forward(ctx, T, Hm):
        H.require_device(T, Hm)
        Hc = H.f32c(Hm).contiguous()
        T = H.f32c(T).contiguous()
        B, D, k = T.shape
        out = torch.empty_like(T)
        H.check(lib.swr_rowmat_fwd(H.ptr(T), H.ptr(Hc), H.ptr(out), B, D, k, H.stream()), "swr_rowmat_fwd")
        ctx.save_for_backward(T, Hc)
        ctx.owner = None
        if ROWMAT_SHARE and Hm.requires_grad:
            ctx.owner = Hm                                   # the caller's object: the uses are counted on it
            Hm._swr_rm_uses = getattr(Hm, "_swr_rm_uses", 0) + 1
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        T, Hm = ctx.saved_tensors
        B, D, k = T.shape
        dout = H.f32c(dout).contiguous()
        dT = torch.empty_like(T) if ctx.needs_input_grad[0] else None
        owner = ctx.owner
        shared = ctx.needs_input_grad[1] and owner is not None and getattr(owner, "_swr_rm_uses", 1) > 1
        dHm, acc, ret = None, 0, None
        if shared:
            st = getattr(owner, "_swr_rm", None)
            if st is None:
                st = owner._swr_rm = {"left": owner._swr_rm_uses, "buf": torch.empty_like(Hm)}
                _rm_pending.append(owner)
                if len(_rm_pending) == 1:
                    torch.autograd.Variable._execution_engine.queue_callback(_rm_check)
            else:
                acc = 1
            dHm = st["buf"]
            st["left"] -= 1
            if st["left"] == 0:
                ret, owner._swr_rm = dHm, None
                owner._swr_rm_uses = 0                       # a factor that outlives the step (a leaf) starts over
        elif ctx.needs_input_grad[1]:
            dHm = ret = torch.empty_like(Hm)
            if owner is not None:
                owner._swr_rm_uses = 0
        if dT is not None or dHm is not None:
            H.check(lib.swr_rowmat_bwd(H.ptr(dout), H.ptr(T), H.ptr(Hm), H.ptr(dT), H.ptr(dHm), acc, B, D, k, H.stream()),
                    "swr_rowmat_bwd")
        return dT, ret


def batch_standardize(x, eps, gamma=None, beta=None):
    return BatchStandardize.apply(x, eps, gamma, beta)


class MatmulIO(Function):
    """y = x @ W + b with W stored [in, out] -- STAR's factorised FCN layer (star.py:103-107)."""

    @staticmethod
    def forward(ctx, x, W, b, act=None):
        H.require_device(x, W)
        ctx.pW, ctx.pb = W, b                    # the caller's objects: parameters whose .grad may live in the gradient arena
        x, W = H.f32c(x), H.f32c(W)
        M, K, N = x.shape[0], W.shape[0], W.shape[1]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        gemm("nn", x, W, y, M, N, K, bias=b)
        ctx.has_b, ctx.act = b is not None, act
        if act is not None:                      # activation on the product (HAMUR's adapter: sigmoid, hamur.py:180,349)
            z, y = y, torch.empty_like(y)
            acts, n_acts = H.act_ranges(act, N)
            H.check(lib.swr_affine_act_fwd(H.ptr(z), N, None, None, acts, n_acts, H.ptr(y), N, M, N, H.stream()),
                    "swr_affine_act_fwd")
            ctx.save_for_backward(x, W, y, z)
        else:
            ctx.save_for_backward(x, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, W = ctx.saved_tensors[:2]
        dy = H.f32c(dy)
        M, K, N = x.shape[0], W.shape[0], W.shape[1]
        if ctx.act is not None:
            y, z = ctx.saved_tensors[2:]
            acts, n_acts = H.act_ranges(ctx.act, N)
            dz = torch.empty((M, N), dtype=torch.float32, device=x.device)
            H.check(lib.swr_act_bwd_apply(H.ptr(dy), dy.stride(0), H.ptr(y), N, H.ptr(z), N, None, None, None, None, acts, n_acts,
                                          H.ptr(dz), N, M, N, H.stream()), "swr_act_bwd_apply")
            dy = dz
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
            gemm("nt", dy, W, dx, M, K, N)                    # dx[m,k] = sum_n dy[m,n] W[k,n]
        # parameter gradients straight into the gradient arena where they live there (zeroed by the optimizer): autograd's
        # AccumulateGrad is one tiny ATen add per parameter and step otherwise (HAMUR's adapter: 6 of them)
        dW = db = None
        if ctx.needs_input_grad[1]:
            gW = _grad_alias([ctx.pW])
            if gW is not None and tuple(gW.shape) == (K, N):
                gemm_tn(x, dy, gW, M, K, N, accumulate=True)
                _mark_touched([ctx.pW])
            else:
                dW = torch.empty((K, N), dtype=torch.float32, device=x.device)
                gemm_tn(x, dy, dW, M, K, N)                   # dW[k,n] = sum_m x[m,k] dy[m,n]
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = _grad_alias([ctx.pb])
            if gb is not None and tuple(gb.shape) == (N,):
                colsum(dy, M, N, out=gb, accumulate=True)
                _mark_touched([ctx.pb])
            else:
                db = torch.empty(N, dtype=torch.float32, device=x.device)
                colsum(dy, M, N, out=db)
        return dx, dW, db, None


# =========================================================================== gate mixing
def make_mix_desc(n_out, n_sel, H_, x_col, g_col, g_stride, sel):
    d = H.MixDesc()
    d.n_out, d.n_sel, d.H, d.x_col, d.g_col, d.g_stride = n_out, n_sel, H_, x_col, g_col, g_stride
    for o in range(n_out):
        for j in range(n_sel):
            d.sel[o][j] = sel[o][j]
    # columns of Y whose gradient swr_moe_mix_bwd writes: experts 0..max(sel) and the n_sel gates of every output
    n_expert = max(max(row[:n_sel]) for row in sel[:n_out]) + 1
    d._n_expert = n_expert
    d._written = set(range(x_col, x_col + n_expert * H_))
    for o in range(n_out):
        d._written.update(range(g_col + o * g_stride, g_col + o * g_stride + n_sel))
    return d


class MoeMix(Function):
    """pooled[:, o*H:(o+1)*H] = sum_j gate_o[:, j] * expert_{sel[o][j]}  (mmoe.py:48-49, ple.py:121-133).
    `Y` holds the activated experts (from x_col) and gate probabilities (from g_col) side by side -- or, with `G`, the experts
    alone and the gate probabilities sit in G (desc.g_col counted in G): PLE's gates are columns of the first layer's output
    while the experts run a second layer (`ple.py:107-125`), and concatenating the two cost a copy pass each way."""

    @staticmethod
    def forward(ctx, Y, desc, width_in, G=None):
        H.require_device(Y)
        Y = H.f32c(Y)
        M = Y.shape[0]
        P = torch.empty((M, desc.n_out * desc.H), dtype=torch.float32, device=Y.device)
        ctx.g_dst = getattr(G, "_swr_grad_dst", None) if G is not None else None
        if G is not None:
            G = H.f32c(G)
        H.check(lib.swr_moe_mix_fwd(C.byref(desc), H.ptr(Y), Y.stride(0) if M > 1 else Y.shape[1], H.ptr(G),
                                    (G.stride(0) if M > 1 else G.shape[1]) if G is not None else 0, H.ptr(P), P.shape[1], M,
                                    H.stream()), "swr_moe_mix_fwd")
        ctx.desc, ctx.width_in, ctx.has_g = desc, width_in, G is not None
        if G is not None:
            ctx.save_for_backward(Y, G)
        else:
            ctx.save_for_backward(Y)
        return P

    @staticmethod
    @once_differentiable
    def backward(ctx, dP):
        Y = ctx.saved_tensors[0]
        G = ctx.saved_tensors[1] if ctx.has_g else None
        dP = H.f32c(dP)
        M = Y.shape[0]
        d = ctx.desc
        # the kernel writes every expert and gate column: a zero fill is only needed when the tensors have other columns
        if G is None:
            full = len(d._written) == ctx.width_in
            dY = (torch.empty if full else torch.zeros)((M, ctx.width_in), dtype=torch.float32, device=Y.device)
            dG = None
        else:
            full = d._n_expert * d.H == ctx.width_in and d.x_col == 0
            dY = (torch.empty if full else torch.zeros)((M, ctx.width_in), dtype=torch.float32, device=Y.device)
            wg = G.shape[1]
            g_full = d.g_col == 0 and d.g_stride == d.n_sel and d.n_out * d.n_sel == wg
            # a block of a split_cols tensor: its gradient goes straight into that tensor's gradient (no copy in SplitCols.backward)
            dG = _grad_dst_view(ctx.g_dst, M, wg, Y.device)
            if dG is None:
                dG = (torch.empty if g_full else torch.zeros)((M, wg), dtype=torch.float32, device=Y.device)
            elif not g_full:
                dG.zero_()
        H.check(lib.swr_moe_mix_bwd(C.byref(d), H.ptr(dP), dP.stride(0) if M > 1 else dP.shape[1], H.ptr(Y),
                                    Y.stride(0) if M > 1 else Y.shape[1], H.ptr(G),
                                    (G.stride(0) if M > 1 else G.shape[1]) if G is not None else 0, H.ptr(dY), ctx.width_in,
                                    H.ptr(dG), (dG.stride(0) if M > 1 else dG.shape[1]) if dG is not None else 0, 0, M, H.stream()),
                "swr_moe_mix_bwd")
        return dY, None, None, dG


def moe_mix_separate_ok(X, G, desc):
    """True when MoeMix can read the gate probabilities from a tensor of their own (the 16-byte path of swr_moe_mix_*)."""
    return (X.is_cuda and X.dtype == torch.float32 and G.dtype == torch.float32 and desc.H % 4 == 0 and desc.x_col % 4 == 0
            and X.stride(1) == 1 and G.stride(1) == 1 and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0
            and os.environ.get("SWR_MIX_SEPARATE", "1") != "0")


# =========================================================================== select / loss
class DomainSelect(Function):
    """final = 0; for d: final = where(domain == d, y_d, final) (mmoe.py:53-55); y_d = sigmoid(V[:, d]) when
    `apply_sigmoid`, and out = sigmoid(select + extra) when `extra` is given (star.py:117)."""

    @staticmethod
    def forward(ctx, V, domain, apply_sigmoid, extra):
        H.require_device(V, domain)
        V = H.f32c(V)
        M, D = V.shape
        out = torch.empty(M, dtype=torch.float32, device=V.device)
        ex = H.f32c(extra).reshape(-1) if extra is not None else None
        domain = domain.contiguous()
        H.check(lib.swr_select_fwd(H.ptr(V), V.stride(0) if M > 1 else D, D, H.ptr(domain), H.dtype_code(domain),
                                   int(apply_sigmoid), H.ptr(ex), H.ptr(out), M, H.stream()), "swr_select_fwd")
        ctx.save_for_backward(out, domain)
        ctx.meta = (M, D, apply_sigmoid, extra is not None, tuple(extra.shape) if extra is not None else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        out, domain = ctx.saved_tensors
        M, D, apply_sigmoid, has_extra, eshape = ctx.meta
        dout = H.f32c(dout).contiguous()
        dV = torch.empty((M, D), dtype=torch.float32, device=out.device)
        dextra = torch.empty(M, dtype=torch.float32, device=out.device) if has_extra else None
        H.check(lib.swr_select_bwd(H.ptr(dout), H.ptr(out), D, H.ptr(domain), H.dtype_code(domain), int(apply_sigmoid),
                                   int(has_extra), H.ptr(dV), D, H.ptr(dextra), M, H.stream()), "swr_select_bwd")
        return dV, None, None, (dextra.reshape(eshape) if has_extra else None)


class BCEMean(Function):
    """torch.nn.BCELoss(reduction='mean') on probabilities (ctr_trainer.py:56,70)."""

    @staticmethod
    def forward(ctx, p, y):
        H.require_device(p, y)
        p = H.f32c(p).contiguous()
        y = y.contiguous()
        M = p.numel()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        nbytes = lib.swr_bce_workspace_bytes(M)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
        H.check(lib.swr_bce_fwd(H.ptr(p), H.ptr(y), H.dtype_code(y), M, H.ptr(loss), H.ptr(ws), nbytes, H.stream()),
                "swr_bce_fwd")
        ctx.save_for_backward(p, y)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        p, y = ctx.saved_tensors
        dloss = dloss.float().contiguous()
        dp = torch.empty_like(p)
        H.check(lib.swr_bce_bwd(H.ptr(p), H.ptr(y), H.dtype_code(y), p.numel(), H.ptr(dloss), H.ptr(dp), H.stream()),
                "swr_bce_bwd")
        return dp, None


def bce_mean(p, y):
    return BCEMean.apply(p, y)


class SelectBCE(Function):
    """(p, loss) = (domain_select(sigmoid(V)), BCELoss(p, y)) in one launch each way: the model's last op feeding the
    trainer's criterion directly (mmoe.py:51-55 + ctr_trainer.py:70).  p stays a differentiable output: a gradient
    arriving on p (someone used the probabilities as well) is added through the plain select backward."""

    @staticmethod
    def forward(ctx, V, domain, y):
        H.require_device(V, domain, y)
        V = H.f32c(V)
        M, D = V.shape
        domain, y = domain.contiguous(), y.contiguous()
        dev = V.device
        p = torch.empty(M, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        nbytes = lib.swr_bce_workspace_bytes(M)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        adv_hyper, adv_hist, adv_cap = take_loss_rider()
        H.check(lib.swr_select_bce_fwd_adv(H.ptr(V), V.stride(0) if M > 1 else D, D, H.ptr(domain), H.dtype_code(domain),
                                           H.ptr(y), H.dtype_code(y), M, H.ptr(p), H.ptr(loss), H.ptr(ws), nbytes,
                                           H.ptr(H.ticket(dev)), H.ptr(adv_hyper), H.ptr(adv_hist), adv_cap, H.stream()),
                "swr_select_bce_fwd_adv")
        ctx.save_for_backward(p, domain, y)
        ctx.D = D
        ctx.set_materialize_grads(False)
        return p, loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dp, dloss):
        p, domain, y = ctx.saved_tensors
        M, D = p.numel(), ctx.D
        dV = None
        if dloss is not None:
            dV = torch.empty((M, D), dtype=torch.float32, device=p.device)
            dloss = dloss.float().contiguous()
            H.check(lib.swr_select_bce_bwd(H.ptr(p), H.ptr(y), H.dtype_code(y), D, H.ptr(domain), H.dtype_code(domain), M,
                                           H.ptr(dloss), H.ptr(dV), D, H.stream()), "swr_select_bce_bwd")
        if dp is not None:
            extra = torch.empty((M, D), dtype=torch.float32, device=p.device)
            dp = H.f32c(dp).contiguous()
            H.check(lib.swr_select_bwd(H.ptr(dp), H.ptr(p), D, H.ptr(domain), H.dtype_code(domain), 1, 0, H.ptr(extra), D,
                                       None, M, H.stream()), "swr_select_bwd")
            dV = extra if dV is None else dV + extra
        return dV, None, None


class fused_bce(object):
    """`with fused_bce(y) as f: y_pred = model(x)`: while active, the model's final `domain_select(V, domain)` (tower
    sigmoids, no extra term) also produces the mean BCE against `y` in the same launch.  Afterwards
    `f.loss_for(y_pred)` is that loss if `y_pred` IS the selected tensor (nothing was applied on top), else None --
    the caller then evaluates its criterion the ordinary way."""
    _active = None

    def __init__(self, y):
        self.y, self.p, self.loss = y, None, None

    def __enter__(self):
        self._prev, fused_bce._active = fused_bce._active, self
        return self

    def __exit__(self, *exc):
        fused_bce._active = self._prev
        return False

    def loss_for(self, y_pred):
        return self.loss if (self.p is not None and y_pred is self.p) else None


def domain_select(V, domain, apply_sigmoid=True, extra=None):
    req = fused_bce._active
    if (req is not None and req.p is None and apply_sigmoid and extra is None and V.dim() == 2 and
            req.y.numel() == V.shape[0] and V.shape[0] > 0):
        req.p, req.loss = SelectBCE.apply(V, domain, req.y.reshape(-1))
        return req.p
    return DomainSelect.apply(V, domain, apply_sigmoid, extra)


# =========================================================================== small helpers
class Mul(Function):
    """c = a * (scale * b) (PPNet's `hidden * gate_out`, ppnet.py:27; EPNet's `agn_x * gate_output`, epnet.py:30; `scale` =
    the GateNU's gamma when b is the bare sigmoid)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        H.require_device(a, b)
        a, b = H.f32c(a).contiguous(), H.f32c(b).contiguous()
        c = torch.empty_like(a)
        H.check(lib.swr_mul_scale_fwd(H.ptr(a), H.ptr(b), scale, H.ptr(c), a.numel(), H.stream()), "swr_mul_scale_fwd")
        ctx.save_for_backward(a, b)
        ctx.scale = scale
        return c

    @staticmethod
    @once_differentiable
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = H.f32c(dc).contiguous()
        da = db = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            da, db = torch.empty_like(a), torch.empty_like(b)
            H.check(lib.swr_mul_scale_bwd(H.ptr(dc), H.ptr(a), H.ptr(b), ctx.scale, H.ptr(da), H.ptr(db), a.numel(), H.stream()),
                    "swr_mul_scale_bwd")
            return da, db, None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            H.check(lib.swr_mul_scale_fwd(H.ptr(dc), H.ptr(b), ctx.scale, H.ptr(da), a.numel(), H.stream()), "swr_mul_scale_fwd")
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)
            H.check(lib.swr_mul_scale_fwd(H.ptr(dc), H.ptr(a), ctx.scale, H.ptr(db), a.numel(), H.stream()), "swr_mul_scale_fwd")
        return da, db, None


def mul(a, b, scale=1.0):
    return Mul.apply(a, b, float(scale))


class MulSigmoid(Function):
    """c = a * (scale * sigmoid(z)): the sigmoid of a GateNU's output layer and the gating product in one pass each way
    (ppnet.py:27: `hidden * gate_out`, layers.py:318-320: `gamma * sigmoid(.)`); the gate tensor is never stored."""

    @staticmethod
    def forward(ctx, a, z, scale):
        H.require_device(a, z)
        a, z = H.f32c(a).contiguous(), H.f32c(z).contiguous()
        c = torch.empty_like(a)
        H.check(lib.swr_mul_sigmoid_fwd(H.ptr(a), H.ptr(z), scale, H.ptr(c), a.numel(), H.stream()), "swr_mul_sigmoid_fwd")
        ctx.save_for_backward(a, z)
        ctx.scale = scale
        return c

    @staticmethod
    @once_differentiable
    def backward(ctx, dc):
        a, z = ctx.saved_tensors
        dc = H.f32c(dc).contiguous()
        da, dz = torch.empty_like(a), torch.empty_like(z)
        H.check(lib.swr_mul_sigmoid_bwd(H.ptr(dc), H.ptr(a), H.ptr(z), ctx.scale, H.ptr(da), H.ptr(dz), a.numel(), H.stream()),
                "swr_mul_sigmoid_bwd")
        return da, dz, None


def mul_sigmoid(a, z, scale=1.0):
    return MulSigmoid.apply(a, z, float(scale))


class Add(Function):
    """c = a + b of two equally shaped tensors (residual connections); the backward passes the gradient to both."""

    @staticmethod
    def forward(ctx, a, b):
        H.require_device(a, b)
        if a.shape != b.shape:
            raise ValueError("ops.add: shapes differ")
        a, b = H.f32c(a).contiguous(), H.f32c(b).contiguous()
        c = torch.empty_like(a)
        H.check(lib.swr_add_fwd(H.ptr(a), H.ptr(b), H.ptr(c), a.numel(), H.stream()), "swr_add_fwd")
        return c

    @staticmethod
    @once_differentiable
    def backward(ctx, dc):
        return (dc if ctx.needs_input_grad[0] else None), (dc if ctx.needs_input_grad[1] else None)


def add(a, b):
    return Add.apply(a, b)


class StopGradCols(Function):
    """Identity whose gradient is zero on columns [lo, hi): `torch.cat((a, b.detach()), dim=1)` of the
    reference (ppnet.py:54, epnet.py:27) without materialising the concatenation."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.span = (lo, hi)
        return x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = g.clone()
        g[:, ctx.span[0]:ctx.span[1]] = 0
        return g, None, None


class SplitCols(Function):
    """Consecutive column blocks of x as views: `x[:, 0:w0], x[:, w0:w0+w1], ...`.  The backward pass writes the blocks'
    gradients side by side into ONE tensor.  Plain slicing leaves that to autograd, which allocates a zero-filled
    full-width tensor per block, copies the block's gradient into it and adds the L tensors up: at PPNet's config 6
    (gate hidden layers of all D*L GateNUs in one [32 768, 1 792] tensor, three blocks) 3 fills + 3 copies + 2 adds of
    235 MB each, ~0.6 ms of a 4.6 ms step."""

    @staticmethod
    def forward(ctx, x, *widths):
        ctx.widths, ctx.meta = widths, (tuple(x.shape), x.dtype, x.device)
        ctx.set_materialize_grads(False)
        # a consuming layer that computes the whole gradient of its block (LinearBNAct) writes it straight into the shared
        # tensor (`_swr_grad_dst`, picked up through _grad_dst_view): no copy for that block in backward()
        ctx.box = box = {"buf": None, "shape": tuple(x.shape), "dtype": x.dtype, "device": x.device}
        outs, off = [], 0
        for w in widths:
            v = x[:, off:off + w]
            v._swr_grad_dst = (box, off, w)
            outs.append(v)
            off += w
        if off > x.shape[1]:
            raise ValueError("split_cols: the blocks are wider than the tensor")
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        shape, dtype, dev = ctx.meta
        if all(g is None for g in gs):
            return (None,) * (1 + len(ctx.widths))
        out, ctx.box["buf"] = ctx.box["buf"], None
        if out is None:
            out = torch.empty(shape, dtype=dtype, device=dev)
        off = 0
        for w, g in zip(ctx.widths, gs):
            if g is None:
                out[:, off:off + w].zero_()
            elif not (g.data_ptr() == out.data_ptr() + 4 * off and g.stride(0) == shape[1] and g.stride(1) == 1
                      and out.dtype == torch.float32):
                out[:, off:off + w].copy_(g)               # (else: the consumer wrote it in place)
            off += w
        if off < shape[1]:
            out[:, off:].zero_()
        return (out,) + (None,) * len(ctx.widths)


def _grad_dst_view(dst, M, width, dev):
    """The block of a split_cols gradient tensor that belongs to the view a layer consumed, or None."""
    if dst is None:
        return None
    box, off, w = dst
    if w != width or box["dtype"] != torch.float32 or box["shape"][0] != M or box["shape"][1] % 4 or off % 4:
        return None
    if box["buf"] is None:
        box["buf"] = torch.empty(box["shape"], dtype=torch.float32, device=dev)
    return box["buf"][:, off:off + w]


def split_cols(x, widths):
    return SplitCols.apply(x, *[int(w) for w in widths])


# =========================================================================== evaluation metrics
def eval_metrics(prob, label, domain, n_domains):
    """Per-domain and overall log-loss / ROC-AUC ingredients of a prediction run, computed on the device
    (csrc/metrics.hip) instead of `.tolist()` + sklearn on the host (`trainers/ctr_trainer.py:99-165`).
    prob [n] fp32, label [n] any value dtype, domain [n] any integer dtype.  Returns four python lists of length
    n_domains + 1 (last slot = all rows): rows, positives, 2U (integers; AUC = 2U / (2 P N)), log-loss sums (floats)."""
    H.require_device(prob, label, domain)
    prob = H.f32c(prob.reshape(-1))
    label = label.reshape(-1).contiguous()
    domain = domain.reshape(-1).contiguous()
    n = prob.numel()
    if label.numel() != n or domain.numel() != n:
        raise ValueError("eval_metrics: prob, label and domain must have the same length")
    D = int(n_domains)
    nbytes = lib.swr_eval_metrics_workspace_bytes(n, D)
    if nbytes == 0:
        raise H.SwrError("swr_eval_metrics: unsupported shape (n < 2^31, 1 <= n_domains <= 254)")
    dev = prob.device
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    counts = torch.empty((D + 1) * 3, dtype=torch.int64, device=dev)
    ll = torch.empty(D + 1, dtype=torch.float64, device=dev)
    H.check(lib.swr_eval_metrics(H.ptr(prob), H.ptr(label), H.dtype_code(label), H.ptr(domain), H.dtype_code(domain), n, D,
                                 H.ptr(counts), H.ptr(ll), H.ptr(ws), nbytes, H.stream()), "swr_eval_metrics")
    c = counts.cpu().view(D + 1, 3).tolist()
    return [r[0] for r in c], [r[1] for r in c], [r[2] for r in c], ll.cpu().tolist()


# =========================================================================== STAR factorised weights
def _star_fwd_args(first, D, ps):
    """swr_star_layer_args of one layer's forward + the stacked outputs it writes."""
    I, O = ps[0].shape
    dev = ps[0].device
    off = 4 if first else 2
    a = H.StarLayerArgs()
    a.D, a.in_dim, a.out_dim, a.first = D, I, O, int(first)
    a.Ws, a.bs = ps[0].data_ptr(), ps[1].data_ptr()
    if first:
        a.gamma_s, a.beta_s = ps[2].data_ptr(), ps[3].data_ptr()
    Wst = torch.empty((D * O, I), dtype=torch.float32, device=dev)
    bst = torch.empty(D * O, dtype=torch.float32, device=dev)
    for d in range(D):
        a.Wd[d], a.bd[d] = ps[off + d].data_ptr(), ps[off + D + d].data_ptr()
        if first:
            a.gamma_d[d], a.beta_d[d] = ps[off + 2 * D + d].data_ptr(), ps[off + 3 * D + d].data_ptr()
        a.W_eff[d] = Wst.data_ptr() + 4 * d * O * I
        a.b_eff[d] = bst.data_ptr() + 4 * d * O
    return a, tuple(Wst[d * O:(d + 1) * O] for d in range(D)) + tuple(bst[d * O:(d + 1) * O] for d in range(D))


def _star_bwd_args(first, D, params, ps, needs, grads):
    """swr_star_layer_args of one layer's backward; -> (args, gradient tensors, written straight into the arena?, keep-alive)."""
    I, O = params[0].shape
    off = 4 if first else 2
    # straight into the gradient arena when every parameter's .grad lives there (zero_grad zeroed it), else fresh tensors
    direct = [_grad_alias([p]) if needs[j] else None for j, p in enumerate(params)]
    all_direct = all(g is not None or not needs[j] for j, g in enumerate(direct))
    if all_direct:
        out = direct
    else:
        out = [torch.empty_like(p, memory_format=torch.contiguous_format) if needs[j] else None for j, p in enumerate(params)]
    a = H.StarLayerArgs()
    a.D, a.in_dim, a.out_dim, a.first, a.accumulate = D, I, O, int(first), int(all_direct)
    a.Ws, a.bs = ps[0].data_ptr(), ps[1].data_ptr()
    ptr = lambda t: t.data_ptr() if t is not None else None
    a.dWs, a.dbs = ptr(out[0]), ptr(out[1])
    if first:
        a.gamma_s, a.beta_s = ps[2].data_ptr(), ps[3].data_ptr()
        a.dgamma_s, a.dbeta_s = ptr(out[2]), ptr(out[3])
    keep = []
    for d in range(D):
        a.Wd[d], a.bd[d] = ps[off + d].data_ptr(), ps[off + D + d].data_ptr()
        a.dWd[d], a.dbd[d] = ptr(out[off + d]), ptr(out[off + D + d])
        if first:
            a.gamma_d[d], a.beta_d[d] = ps[off + 2 * D + d].data_ptr(), ps[off + 3 * D + d].data_ptr()
            a.dgamma_d[d], a.dbeta_d[d] = ptr(out[off + 2 * D + d]), ptr(out[off + 3 * D + d])
        gW, gb = grads[d], grads[D + d]
        if gW is not None:
            gW = H.f32c(gW)
            keep.append(gW)
            a.dW_eff[d] = gW.data_ptr()
        if gb is not None:
            gb = H.f32c(gb)
            keep.append(gb)
            a.db_eff[d] = gb.data_ptr()
    return a, out, all_direct, keep


class StarLayerWeights(Function):
    """Effective weights / biases of one STAR layer for all domains (reference `star.py:99-107`; the first layer also
    folds in the partitioned norm's domain affine, `star.py:91-100`): one launch each way (csrc/star.hip) instead of
    ~10 elementwise launches per (layer, domain) and three times as many in the backward pass.

    params = Ws [in, out], bs [out], (first: gamma_s [in], beta_s [in]), D x Wd, D x bd, (first: D x gamma_d, D x beta_d).
    Returns D weights in Linear layout [out, in] (adjacent views of one buffer: the stacked operand of the layer's
    product is a zero-copy view) followed by D biases [out]."""

    @staticmethod
    def forward(ctx, first, D, *params):
        H.require_device(*params)
        ps = [H.f32c(p.detach()) for p in params]
        a, outs = _star_fwd_args(first, D, ps)
        H.check(lib.swr_star_layer_fwd(C.byref(a), H.stream()), "swr_star_layer_fwd")
        ctx.first, ctx.D, ctx.params, ctx.keep = first, D, params, ps
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        first, D, params, ps = ctx.first, ctx.D, ctx.params, ctx.keep
        n = len(params)
        needs = ctx.needs_input_grad[2:]
        a, out, all_direct, _keep = _star_bwd_args(first, D, params, ps, needs, grads)
        H.check(lib.swr_star_layer_bwd(C.byref(a), H.stream()), "swr_star_layer_bwd")
        if all_direct:
            _mark_touched([p for j, p in enumerate(params) if needs[j]])
            return (None, None) + (None,) * n
        return (None, None) + tuple(out)


class StarStackWeights(Function):
    """StarLayerWeights for EVERY layer of the FCN stack in one call each way: all effective weights before the first
    product (`swr_star_layers_fwd`), all parameter gradients after the last weight-gradient product (`swr_star_layers_bwd`;
    autograd runs this backward once every layer's dW_eff exists).  Parameter-sized tensors: 14 latency-bound launches per
    step at config 3 become 4.  counts[l] = number of parameters of layer l (the first layer is layer 0); returns the
    layers' outputs (2 D tensors each) one after another."""

    @staticmethod
    def forward(ctx, D, counts, *params):
        H.require_device(*params)
        ps = [H.f32c(p.detach()) for p in params]
        args = (H.StarLayerArgs * len(counts))()
        outs, pos = (), 0
        for l, n in enumerate(counts):
            a, o = _star_fwd_args(l == 0, D, ps[pos:pos + n])
            args[l] = a
            outs += o
            pos += n
        H.check(lib.swr_star_layers_fwd(args, len(counts), H.stream()), "swr_star_layers_fwd")
        ctx.D, ctx.counts, ctx.params, ctx.keep = D, counts, params, ps
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        D, counts, params, ps = ctx.D, ctx.counts, ctx.params, ctx.keep
        args = (H.StarLayerArgs * len(counts))()
        result, touched, keep, pos = [], [], [], 0
        for l, n in enumerate(counts):
            needs = ctx.needs_input_grad[2 + pos:2 + pos + n]
            a, out, all_direct, k = _star_bwd_args(l == 0, D, params[pos:pos + n], ps[pos:pos + n], needs,
                                                   grads[2 * D * l:2 * D * (l + 1)])
            args[l] = a
            keep.append(k)
            if all_direct:
                touched += [p for j, p in enumerate(params[pos:pos + n]) if needs[j]]
                result += [None] * n
            else:
                result += list(out)
            pos += n
        H.check(lib.swr_star_layers_bwd(args, len(counts), H.stream()), "swr_star_layers_bwd")
        _mark_touched(touched)
        return (None, None) + tuple(result)


def star_layer_weights(first, D, *params):
    return StarLayerWeights.apply(bool(first), int(D), *params)


def star_stack_weights(D, layer_params):
    """layer_params[l] = the parameter list star_layer_weights takes for layer l (layer 0 = the first layer) ->
    [outputs of layer 0, outputs of layer 1, ...], each 2 D tensors (D weights [out, in], D biases)."""
    counts = tuple(len(ps) for ps in layer_params)
    flat = [p for ps in layer_params for p in ps]
    outs = StarStackWeights.apply(int(D), counts, *flat)
    return [outs[2 * D * l:2 * D * (l + 1)] for l in range(len(counts))]


# =========================================================================== routed inference
ROUTED_EVAL = os.environ.get("SWR_ROUTED_EVAL", "1") != "0"


def routed_eval_ok(x):
    """Inference only: BatchNorm is a fixed affine there, so a row needs nothing but its own domain's branch."""
    return ROUTED_EVAL and x.is_cuda and not torch.is_grad_enabled()


class DomainRouting(object):
    """Rows grouped by domain for inference: `rows(x)` = x with the rows of domain 0 first, then domain 1, ... (ids outside
    [0, D) last); `segment(xs, d)` = the contiguous rows of domain d; `scatter(parts)` puts the per-domain results back in
    batch order, zeros where the id is out of range (the reference's chain of `where` from zeros, mmoe.py:53-55).  The
    reference evaluates every domain's branch on the whole batch and selects; in eval mode this is the same numbers
    with 1/D of the tower work.  One host sync per batch (the segment bounds)."""

    def __init__(self, domain_id, D):
        d = domain_id.reshape(-1).long()
        key = torch.where((d >= 0) & (d < D), d, torch.full_like(d, D))
        self.perm = torch.argsort(key, stable=True)
        self.bounds = [0] + torch.cumsum(torch.bincount(key, minlength=D + 1), 0).tolist()
        self.D, self.B = D, d.numel()

    def rows(self, x):
        return x.index_select(0, self.perm)

    def segment(self, xs, d):
        return xs[self.bounds[d]:self.bounds[d + 1]]

    def count(self, d):
        return self.bounds[d + 1] - self.bounds[d]

    def scatter(self, parts):
        """parts[d]: [count(d)] or [count(d), 1] (None for an empty domain) -> [B]."""
        vals = [p.reshape(-1) for p in parts if p is not None and p.numel()]
        out = torch.zeros(self.B, dtype=torch.float32, device=self.perm.device)
        if vals:
            cat = torch.cat(vals)
            out.index_copy_(0, self.perm[:cat.numel()], cat)
        return out

    def valid(self):
        m = torch.zeros(self.B, dtype=torch.bool, device=self.perm.device)
        m[self.perm[:self.bounds[self.D]]] = True
        return m


def routed_probs(route, logit_parts):
    """sigmoid of the selected logits, exactly 0.0 for rows whose domain id is out of range."""
    sel = route.scatter(logit_parts)
    return torch.where(route.valid(), torch.sigmoid(sel), torch.zeros_like(sel))

def routed_mmoe_eval(y, n_expert, H_, towers_w1, towers_b1, towers_bn, towers_w2, towers_b2, domain_id):
    """Eval-mode MMoE head, routed: every row mixes the experts with its own domain's gate probabilities and runs its
    own domain's tower (csrc/routed.hip) instead of every domain's on the whole batch followed by the select
    (`mmoe.py:48-55`).  y [B, ne*H + D*ne] = expert outputs + gate softmax probabilities; returns probabilities [B]."""
    H.require_device(y, towers_w1[0], domain_id)
    y = H.f32c(y)
    D = len(towers_w1)
    T = towers_w1[0].shape[0]
    dev = y.device
    W1, b1 = _cat_params(towers_w1), _cat_params(towers_b1)
    w2, b2 = _cat_params([w.reshape(-1) for w in towers_w2]), _cat_params(towers_b2)
    gamma, beta = _cat_params(towers_bn["gamma"]), _cat_params(towers_bn["beta"])
    rm, rv = _cat_params(towers_bn["running_mean"]), _cat_params(towers_bn["running_var"])
    scale = torch.empty(D * T, dtype=torch.float32, device=dev)
    shift = torch.empty(D * T, dtype=torch.float32, device=dev)
    H.check(lib.swr_bn_eval_coeffs(H.ptr(gamma), H.ptr(beta), H.ptr(rm), H.ptr(rv), towers_bn["eps"], D * T, H.ptr(scale),
                                   H.ptr(shift), H.stream()), "swr_bn_eval_coeffs")
    dom = domain_id.reshape(-1).contiguous()
    out = torch.empty(y.shape[0], dtype=torch.float32, device=dev)
    H.check(lib.swr_routed_mmoe_eval(H.ptr(y), y.stride(0), y.shape[0], n_expert, H_, D, T, H.ptr(W1), H.ptr(b1), H.ptr(scale),
                                     H.ptr(shift), H.ptr(w2), H.ptr(b2), H.ptr(dom), H.dtype_code(dom), H.ptr(out), H.stream()),
            "swr_routed_mmoe_eval")
    return out


def routed_mmoe_eval_supported(n_expert, H_, D, T):
    return bool(lib.swr_routed_mmoe_eval_supported(int(n_expert), int(H_), int(D), int(T)))


# =========================================================================== LayerNorm (+ReLU), block select  (M3oE)
class LayerNormAct(Function):
    """torch.nn.LayerNorm(N, eps) (+ ReLU) over G side-by-side column groups of x [M, G*N], each with its own gamma / beta
    (reference `m3oe.py:49-67` Mlp_N blocks, `:121-128` towers); csrc/layernorm.hip.  params = G gammas + G betas."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        G, N, relu, eps = cfg["G"], cfg["N"], cfg["relu"], cfg["eps"]
        H.require_device(x, params[0])
        x = H.f32c(x)
        if x.stride(1) != 1:
            x = x.contiguous()
        M = x.shape[0]
        dev = x.device
        gamma, beta = _cat_params(params[:G]), _cat_params(params[G:])
        y = torch.empty((M, G * N), dtype=torch.float32, device=dev)
        want = any(ctx.needs_input_grad[1:])
        mean = torch.empty((M, G), dtype=torch.float32, device=dev) if want else None
        rstd = torch.empty((M, G), dtype=torch.float32, device=dev) if want else None
        a = H.LayerNormArgs()
        a.M, a.G, a.N, a.relu, a.eps = M, G, N, int(relu), eps
        a.X, a.ldx, a.gamma, a.beta = x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr()
        a.Y, a.ldy = y.data_ptr(), G * N
        a.mean, a.rstd = (mean.data_ptr(), rstd.data_ptr()) if want else (None, None)
        H.check(lib.swr_layernorm_fwd(C.byref(a), H.stream()), "swr_layernorm_fwd")
        ctx.cfg, ctx.params = cfg, params
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        cfg = ctx.cfg
        G, N = cfg["G"], cfg["N"]
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        M = x.shape[0]
        dev = x.device
        dy = H.f32c(dy)
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        p_g, p_b = ctx.params[:G], ctx.params[G:]
        dgamma, dbeta = _grad_alias(p_g, 2), _grad_alias(p_b, 2)
        direct = dgamma is not None and dbeta is not None
        if not direct:
            dgamma = torch.empty(G * N, dtype=torch.float32, device=dev)
            dbeta = torch.empty(G * N, dtype=torch.float32, device=dev)
        dx = torch.empty((M, G * N), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        a = H.LayerNormArgs()
        a.M, a.G, a.N, a.relu, a.eps, a.accumulate = M, G, N, int(cfg["relu"]), cfg["eps"], int(direct)
        a.X, a.ldx, a.gamma, a.beta = x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr()
        a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
        a.dY, a.lddy = dy.data_ptr(), dy.stride(0)
        a.dX, a.lddx = (dx.data_ptr(), G * N) if dx is not None else (None, 0)
        a.dgamma, a.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
        nbytes = lib.swr_layernorm_bwd_workspace_bytes(M, G, N)
        if nbytes == 0:
            raise H.SwrError("swr_layernorm: unsupported width")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        H.check(lib.swr_layernorm_bwd(C.byref(a), H.ptr(ws), nbytes, H.stream()), "swr_layernorm_bwd")
        if direct:
            _mark_touched(tuple(p_g) + tuple(p_b))
            return (None, dx) + (None,) * (2 * G)
        return (None, dx) + tuple(_split_like(dgamma, p_g)) + tuple(_split_like(dbeta, p_b))


def layer_norm_act(x, norms, relu=True):
    """`norms`: G nn.LayerNorm modules of equal width applied to the G column groups of x."""
    n0 = norms[0]
    cfg = {"G": len(norms), "N": int(n0.normalized_shape[0]), "relu": bool(relu), "eps": float(n0.eps)}
    return LayerNormAct.apply(cfg, x, *[n.weight for n in norms], *[n.bias for n in norms])


class BlockSelect(Function):
    """out[b] = V[b, d_b*H : (d_b+1)*H] for 0 <= d_b < D, else zeros (`m3oe.py:141-146`)."""

    @staticmethod
    def forward(ctx, V, domain, D, Hh):
        H.require_device(V, domain)
        V = H.f32c(V)
        if V.stride(1) != 1:
            V = V.contiguous()
        domain = domain.contiguous()
        M = V.shape[0]
        out = torch.empty((M, Hh), dtype=torch.float32, device=V.device)
        H.check(lib.swr_block_select_fwd(H.ptr(V), V.stride(0), H.ptr(domain), H.dtype_code(domain), D, Hh, M, H.ptr(out), Hh,
                                         H.stream()), "swr_block_select_fwd")
        ctx.dims = (M, D, Hh)
        ctx.save_for_backward(domain)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (domain,) = ctx.saved_tensors
        M, D, Hh = ctx.dims
        dout = H.f32c(dout)
        if dout.stride(1) != 1:
            dout = dout.contiguous()
        dV = torch.empty((M, D * Hh), dtype=torch.float32, device=dout.device)
        H.check(lib.swr_block_select_bwd(H.ptr(dout), dout.stride(0), H.ptr(domain), H.dtype_code(domain), D, Hh, M, H.ptr(dV),
                                         D * Hh, H.stream()), "swr_block_select_bwd")
        return dV, None, None, None


def block_select(V, domain, D, Hh):
    return BlockSelect.apply(V, domain, int(D), int(Hh))
