"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI).  Replaces the reference's single-process `torch.nn.DataParallel`
(`trainers/ctr_trainer.py:45-47`) and keeps ITS semantics:

  * the batch shards by row; every rank holds a full replica of all tables and parameters;
  * BatchNorm statistics are per shard (no SyncBN); rank 0's running statistics are the model's;
  * the loss is the mean over the GLOBAL batch: every rank back-propagates its local mean loss and the
    gradients are averaged (equal shards), which equals the sum of the reference's per-replica gradients;
  * one exchange step per training step, two all-gathers (xGMI is point-to-point; the per-rank messages are
    small, so gathering everything and reducing locally beats an all-reduce tree):
      1. rows: per large table the rank's row-reduced entries (row id, summed gradient) -- sent as soon as the
         backward has produced them, in flight while the weight-gradient products still compute;
      2. dense: the flat gradient arena (all non-table parameters and all small tables; 0.5 MB at the KuaiRand
         config -- latency-bound, so a single call, not one per tensor);
    then ONE launch (swr_dp_finish) adds the arenas in rank order and merges the row lists without a sort; every
    rank computes bit-identical gradients from the same gathered bytes, so replicas never drift (there is no
    parameter broadcast after step 0).
  * the optimizer runs replicated.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import ops



# The captured data-parallel step is ONE hipGraph with the RCCL collectives inside it wherever the process group runs over RCCL
# (world 1 over RCCL, same box: 0.470 -> 0.428 ms per step at config 2, i.e. 1.27 x -> 1.16 x the single-GPU step); gloo collectives
# are host-synchronous and cannot be captured: three graphs with eager collectives between them (the multi-process tests on one GPU)
ONE_STREAM = os.environ.get("SWR_DP_ONE_STREAM", "auto")      # "1" / "0" / auto = with the step (ops.SIDE_STREAM): the exchange on the step's one stream
EARLY_ROWS = os.environ.get("SWR_DP_EARLY_ROWS", "1") != "0"   # one-graph step: catch-up + step bookkeeping on the exchange's branch
ROWS_EVENT = os.environ.get("SWR_DP_ROWS_EVENT", "1") != "0"   # one-graph step: the rows all-gather waits for the row lists alone
ONE_GRAPH = os.environ.get("SWR_DP_ONE_GRAPH", "auto")      # "auto": one graph over RCCL (nccl backend), three with gloo; "0" / "1" force


def allreduce_min_bytes():
    """Gradient arenas above this size are ALL-REDUCED (SURVEY.md 8e / north star: "a single RCCL all-reduce over xGMI on
    the dense parameters") instead of all-gathered and summed locally: an all-gather delivers N x arena bytes to every
    rank -- fine for the 0.5 MB arena of the KuaiRand MMoE, where one latency-bound collective + a rank-ordered local sum
    is the cheapest exchange, wasteful for STAR's 4 MB (30 MB received per rank per step at 8 ranks).  Every rank receives
    the same reduced bytes from an all-reduce, so replicas stay bitwise equal either way."""
    return int(os.environ.get("SWR_DP_ALLREDUCE_BYTES", 1 << 20))


def hip_merge_rows(urow, ugrad, vocab):
    """Deterministic merge of gathered row entries on the device: (row id or -1, gradient) x n ->
    the same representation with every distinct row listed once (K3's reduction, swr_embed_bwd)."""
    from . import _hip as H
    from ._hip import lib
    n, dim = ugrad.shape
    keys = torch.where(urow < 0, torch.zeros_like(urow), urow).contiguous()     # -1 entries carry a zero gradient
    out_row = torch.empty(n, dtype=torch.int32, device=urow.device)
    out_grad = torch.empty((n, dim), dtype=torch.float32, device=urow.device)
    slot = (H.EmbedGradSlot * 1)()
    slot[0] = H.EmbedGradSlot(vocab, dim, 0, 0, 1, None, out_row.data_ptr(), out_grad.data_ptr())
    nbytes = lib.swr_embed_bwd_workspace_bytes(slot, 1, n)
    if nbytes == 0:
        raise H.SwrError("swr_embed_bwd: unsupported merge shape")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=urow.device)
    H.check(lib.swr_embed_bwd(slot, 1, H.ptr(keys), H.ptr(ugrad), dim, n, H.ptr(ws), nbytes,
                              H.ptr(H.err_flag(urow.device)), H.stream()), "swr_embed_bwd(merge)")
    return out_row, out_grad


def exchange_layout(dense_grad, sparse_grads):
    """Layout of one rank's message, in fp32 words: [gradient arena | per large table: row ids (int32 bits), gradients]."""
    A = dense_grad.numel() if dense_grad is not None else 0
    offs, pos = [], A
    for urow, ugrad, vocab in sparse_grads:
        n, dim = ugrad.shape
        offs.append((pos, pos + n, n, dim, vocab))
        pos += n + n * dim
    return A, offs, (pos + 3) // 4 * 4          # whole 16-byte units: every rank's message starts aligned in `recv`


def communicate(dense_grad, sparse_grads, world_size, group=None, buffers=None):
    """The collective of the exchange step, nothing else: ONE all-gather of every rank's packed message (the flat
    gradient arena, and per large table the row ids + row gradients of its reduced entries).  xGMI is point-to-point
    and a step has no other exchange, so one large collective beats an all-reduce plus two all-gathers per table.
    `buffers` = (send, recv) preallocated (static addresses for hipGraph use); tensors that already live inside
    `send` (DataParallelStep points the backward's outputs there) are not copied."""
    A, offs, total = exchange_layout(dense_grad, sparse_grads)
    ref = dense_grad if A else sparse_grads[0][1]
    if buffers is None:
        send = torch.empty(total, dtype=torch.float32, device=ref.device)
        recv = torch.empty(world_size * total, dtype=torch.float32, device=ref.device)
    else:
        send, recv = buffers
    base = send.data_ptr()
    if A and dense_grad.data_ptr() != base:
        send[:A].copy_(dense_grad.reshape(-1))
    for (r0, r1, n, dim, _v), (urow, ugrad, _vocab) in zip(offs, sparse_grads):
        if urow.data_ptr() != base + 4 * r0:
            send[r0:r1].copy_(urow.contiguous().view(torch.float32))        # bit copy of the int32 ids
        if ugrad.data_ptr() != base + 4 * r1:
            send[r1:r1 + n * dim].copy_(ugrad.reshape(-1))
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv, A, offs, total


def finish(dense_grad, gathered, world_size, merge_rows=hip_merge_rows):
    """Device-side remainder of the exchange: the arenas of all ranks are summed in rank order and averaged, the
    gathered row entries are merged deterministically (every rank computes bit-identical results)."""
    recv, A, offs, total = gathered
    if recv.is_cuda and merge_rows is hip_merge_rows and world_size <= 8 and len(offs) <= 16:
        return _hip_finish(dense_grad, recv, A, offs, total, world_size)
    R = recv.view(world_size, total)
    if A:
        dense_grad.reshape(-1).copy_(R[:, :A].sum(dim=0))
        dense_grad.mul_(1.0 / world_size)
    merged = []
    for r0, r1, n, dim, vocab in offs:
        rows = R[:, r0:r1].contiguous().view(torch.int32).reshape(-1)
        grads = R[:, r1:r1 + n * dim].reshape(world_size * n, dim)
        r, g = merge_rows(rows, grads.contiguous(), vocab)
        merged.append((r, g.mul_(1.0 / world_size)))
    return merged


def _hip_finish(dense_grad, recv, A, offs, total, world_size):
    """`finish` as ONE launch (swr_dp_finish, csrc/exchange.hip): rank-ordered mean of the gradient arenas and the
    sort-free merge of the large tables' row lists."""
    from . import _hip as H
    from ._hip import lib
    tabs = (H.DpTable * max(1, len(offs)))()
    merged = []
    for t, (r0, r1, n, dim, _vocab) in enumerate(offs):
        out_row = torch.empty(world_size * n, dtype=torch.int32, device=recv.device)
        out_grad = torch.empty((world_size * n, dim), dtype=torch.float32, device=recv.device)
        tabs[t] = H.DpTable(r0, r1, n, dim, 0, out_row.data_ptr(), out_grad.data_ptr())
        merged.append((out_row, out_grad))
    dense = dense_grad.reshape(-1) if A else None
    if A and dense.data_ptr() != dense_grad.data_ptr():
        raise H.SwrError("exchange: the gradient arena must be contiguous")
    H.check(lib.swr_dp_finish(H.ptr(recv), total, A, H.ptr(dense) if A else None, H.ptr(recv), total, tabs, len(offs),
                              world_size, 1.0 / world_size, H.stream()), "swr_dp_finish")
    return merged


def exchange_gradients(dense_grad, sparse_grads, world_size, group=None, merge_rows=hip_merge_rows):
    """The exchange step.  `dense_grad`: the flat gradient arena (averaged in place).
    `sparse_grads`: list of (urow int32 [n], ugrad fp32 [n, dim], vocab) per large table.
    Returns the merged, averaged (urow, ugrad) per table.  An arena above `allreduce_min_bytes()` is all-reduced; the
    row lists (and a small arena) travel in one all-gather."""
    if dense_grad is not None and dense_grad.numel() * 4 > allreduce_min_bytes():
        dist.all_reduce(dense_grad, group=group)
        dense_grad.mul_(1.0 / world_size)
        if not sparse_grads:
            return []
        return finish(None, communicate(None, sparse_grads, world_size, group), world_size, merge_rows)
    return finish(dense_grad, communicate(dense_grad, sparse_grads, world_size, group), world_size, merge_rows)


class DataParallelStep(object):
    """`CTRTrainer.train_step` with the gradient exchange between backward and the optimizer step.

    The backward pass is split (ops.split_backward): the large tables' row lists are produced first and leave for
    the other ranks at once (an asynchronous all-gather on RCCL's stream); the weight-gradient products and the small
    tables' gradients -- results only the optimizer needs -- are computed WHILE that all-gather is in flight; a second,
    small all-gather carries the gradient arena; one launch (swr_dp_finish) averages and merges.

    `train_step` launches everything eagerly.  `capture(x, y)` records the step as THREE hipGraphs (forward + backward
    up to the row lists | held-back gradient work | arena mean + optimizer) with the two collectives and the row merge
    issued eagerly in between, so a replayed step costs three graph launches and two collectives instead of ~60 launches."""

    def __init__(self, trainer, world_size=None, group=None):
        self.trainer = trainer
        self.group = group
        self.world_size = world_size if world_size is not None else dist.get_world_size(group)
        model = trainer.model
        if hasattr(model, "build_arena") and model.arena() is None:
            model.build_arena()
        # identical replicas at step 0: broadcast rank 0's parameters and buffers once
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0, group=group)
        self._graphs = None

    # ---- pieces ---------------------------------------------------------------------------------------
    def _forward_backward(self, x_dict, y, rows_event=None):
        """Forward + the part of the backward that ends with the large tables' row lists; the rest waits in
        ops.run_late_jobs()."""
        model = self.trainer.model
        ops.split_backward(hasattr(model, "arena") and model.arena() is not None, rows_event)
        try:
            return self.trainer.forward_backward(x_dict, y).detach()
        finally:
            ops.split_backward(False)

    def _sparse(self):
        arena = self.trainer.model.arena()
        big = [p for p in arena["big"] if getattr(p, "_swr_sparse_grad", None) is not None]
        return arena, big, [(p._swr_sparse_grad[0], p._swr_sparse_grad[1], p.shape[0]) for p in big]

    def _exchange_buffers(self, dense, big, sparse):
        """Static send / receive buffers of the two messages: rows = per large table [row ids | row gradients], and
        dense = the gradient arena.  The large tables' backward is pointed at its slots of the rows message
        (`_swr_sparse_out`, read by ops.EmbedGather.backward) so that from the next step on nothing is copied but the
        0.5 MB gradient arena."""
        if self.world_size > 8 or len(sparse) > 16:
            raise ops.H.SwrError("DataParallelStep: at most 8 ranks and 16 row-sparse tables")
        A = dense.numel()
        offs, pos = [], 0
        for urow, ugrad, vocab in sparse:
            n, dim = ugrad.shape
            offs.append((pos, pos + n, n, dim, vocab))
            pos += n + n * dim
        total, A4 = (pos + 3) // 4 * 4, (A + 3) // 4 * 4
        key = (A, tuple(offs))
        if getattr(self, "_xb_key", None) != key:
            self._xb_key = key
            dev, W = dense.device, self.world_size
            f32 = dict(dtype=torch.float32, device=dev)
            # the arena itself is the dense message when its length keeps every rank's copy 16-byte aligned in `recv_d`
            self._xb = {"A": A, "A4": A4, "offs": offs, "total": total, "in_place": A > 0 and A % 4 == 0 and dense.is_contiguous(),
                        "allreduce": A * 4 > allreduce_min_bytes() and dense.is_contiguous(),
                        "send_d": torch.zeros(max(A4, 4), **f32),                                                    # A4 == A when in place
                        "recv_d": torch.empty((1 if A * 4 > allreduce_min_bytes() and dense.is_contiguous() else W) * max(A4, 4), **f32),
                        "send_r": torch.zeros(max(total, 4), **f32), "recv_r": torch.empty(W * max(total, 4), **f32)}
            send = self._xb["send_r"]
            for p, (r0, r1, n, dim, _v) in zip(big, offs):
                p._swr_sparse_out = (send[r0:r1].view(torch.int32), send[r1:r1 + n * dim].view(n, dim))
            # merged row lists (static: the captured optimizer step reads them)
            H = ops.H
            tabs = (H.DpTable * max(1, len(offs)))()
            merged = []
            for t, (r0, r1, n, dim, _vocab) in enumerate(offs):
                out_row = torch.empty(W * n, dtype=torch.int32, device=dev)
                out_grad = torch.empty((W * n, dim), **f32)
                tabs[t] = H.DpTable(r0, r1, n, dim, 0, out_row.data_ptr(), out_grad.data_ptr())
                merged.append((out_row, out_grad))
            self._xb["tabs"], self._xb["merged"] = tabs, merged
        return self._xb

    def _send_rows(self, xb, sparse):
        """Start the all-gather of the row lists (asynchronous: it runs on RCCL's stream, ordered after everything
        enqueued so far); returns the work handle, or None without large tables."""
        if not xb["offs"]:
            return None
        send = xb["send_r"]
        base = send.data_ptr()
        for (r0, r1, n, dim, _v), (urow, ugrad, _vocab) in zip(xb["offs"], sparse):
            if urow.data_ptr() != base + 4 * r0:                             # first step only: not yet written in place
                send[r0:r1].copy_(urow.contiguous().view(torch.float32))    # bit copy of the int32 ids
            if ugrad.data_ptr() != base + 4 * r1:
                send[r1:r1 + n * dim].copy_(ugrad.reshape(-1))
        return dist.all_gather_into_tensor(xb["recv_r"], send, group=self.group, async_op=True)

    def _send_dense(self, xb, dense, pack):
        """All-gather of the gradient arena (sent in place when its length allows; else through a packed copy, made
        here when `pack`, or by the captured graph)."""
        if xb["allreduce"]:
            dist.all_reduce(dense.reshape(-1), group=self.group)          # in place: every rank ends with the same sum
            return
        if xb["in_place"]:
            dist.all_gather_into_tensor(xb["recv_d"], dense.reshape(-1), group=self.group)
            return
        if pack:
            xb["send_d"][:xb["A"]].copy_(dense.reshape(-1))
        dist.all_gather_into_tensor(xb["recv_d"], xb["send_d"], group=self.group)

    def _merge_rows(self, xb, big):
        """swr_dp_finish, row lists only (needs just the first all-gather); hands the merged lists to the optimizer."""
        H, W = ops.H, self.world_size
        if xb["offs"]:
            H.check(H.lib.swr_dp_finish(None, 0, 0, None, H.ptr(xb["recv_r"]), max(xb["total"], 4), xb["tabs"],
                                        len(xb["offs"]), W, 1.0 / W, H.stream()), "swr_dp_finish(rows)")
        for p, rg in zip(big, xb["merged"]):
            p._swr_sparse_grad = rg
            p._swr_sparse_local = False      # rows other ranks touched are in the list too

    def _mean_dense(self, xb, dense):
        """swr_dp_finish, gradient arenas only: rank-ordered mean, written back into the arena."""
        H, W = ops.H, self.world_size
        flat = dense.reshape(-1)
        if flat.data_ptr() != dense.data_ptr():
            raise H.SwrError("exchange: the gradient arena must be contiguous")
        if xb["A"] and xb["allreduce"]:
            # the arena holds the all-reduced SUM: one pass scales it to the mean (swr_dp_finish over one "rank", in place)
            H.check(H.lib.swr_dp_finish(H.ptr(flat), max(xb["A4"], 4), xb["A"], H.ptr(flat), None, 0, None, 0,
                                        1, 1.0 / W, H.stream()), "swr_dp_finish(scale)")
        elif xb["A"]:
            H.check(H.lib.swr_dp_finish(H.ptr(xb["recv_d"]), max(xb["A4"], 4), xb["A"], H.ptr(flat), None, 0, None, 0,
                                        W, 1.0 / W, H.stream()), "swr_dp_finish(dense)")

    def train_step(self, x_dict, y):
        tr = self.trainer
        model = tr.model
        loss = self._forward_backward(x_dict, y)
        arena = model.arena() if hasattr(model, "arena") else None
        if arena is None:
            # no arena (foreign module): per-tensor exchange
            ops.run_late_jobs()
            for p in model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad, group=self.group)
                    p.grad.mul_(1.0 / self.world_size)
        else:
            arena, big, sparse = self._sparse()
            xb = self._exchange_buffers(arena["g"], big, sparse)
            work = self._send_rows(xb, sparse)
            ops._skew(7)
            ops.run_late_jobs()                                   # overlaps the row lists' all-gather
            self._send_dense(xb, arena["g"], pack=True)
            if work is not None:
                work.wait()
            self._merge_rows(xb, big)
            self._mean_dense(xb, arena["g"])
        tr.optimizer.step()
        return loss

    # ---- captured variant -----------------------------------------------------------------------------
    def capture(self, x_dict, y, warmup=2):
        """Static input buffers + three captured graphs.  Feed batches with `load`, run with `replay`.  `warmup` eager
        steps on (x, y) first (>= 2 when no step of this shape has run yet); the capture itself executes nothing."""
        self.x = {k: v.clone() for k, v in x_dict.items()}
        self.y = y.clone()
        if warmup > 0:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self.train_step(self.x, self.y)      # results dropped at once (see trainers/graph.py)
            torch.cuda.current_stream().wait_stream(side)
        elif getattr(self, "_xb", None) is None:
            # warmup=0: the caller has run (two) ordinary steps of this shape already -- they established the exchange
            # buffers and pointed the large tables' backward at its slots of the rows message
            raise ops.H.SwrError("DataParallelStep.capture(warmup=0) needs earlier train_step() calls of the same batch shape")
        torch.cuda.synchronize()
        opt = self.trainer.optimizer
        if hasattr(opt, "hist_cap") and 2 * opt._since_flush >= opt.hist_cap:
            opt.materialize()
        snap = opt.host_counts() if hasattr(opt, "host_counts") else None
        one = ONE_GRAPH == "1" or (ONE_GRAPH == "auto" and dist.get_backend(self.group) == "nccl")
        if one:
            err = None
            try:
                self._capture_one(snap)
            except Exception as e:                  # noqa: BLE001  (a runtime that cannot capture its collectives)
                err = e
            # the choice is COLLECTIVE: one rank on three graphs beside seven on one would issue different collective
            # sequences and hang.  (A capture executes nothing, so every rank reaches this eager all-reduce.)
            bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=self.y.device)
            torch.cuda.synchronize()
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
            if int(bad.item()) == 0:
                return self
            if ONE_GRAPH == "1":
                raise ops.H.SwrError(f"SWR_DP_ONE_GRAPH=1: the one-graph capture failed on "
                                     f"{'this rank' if err is not None else 'another rank'}: {err!r}")
            import sys
            print(f"[parallel] one-graph capture failed ({'here: ' + repr(err) if err is not None else 'on another rank'}); "
                  "three graphs", file=sys.stderr)
            # whatever the aborted capture queued (held-back weight-gradient jobs that would otherwise run AGAIN in the second
            # graph below, pending side-branch counters, sparse row lists pointing into the released capture pool) is dropped
            self._graphs = None
            ops.abort_step()
            if snap is not None:
                opt.restore_host_counts(snap)
            for p in self.trainer.model.parameters():
                if getattr(p, "_swr_sparse_grad", None) is not None:
                    p._swr_sparse_grad = None          # a row list the aborted step's optimizer never consumed
            self.trainer.model.zero_grad()
            torch.cuda.synchronize()
            # an eager step re-establishes the exchange buffers and the in-place row-list slots the capture asserts on
            self.train_step(self.x, self.y)
            torch.cuda.synchronize()
            snap = opt.host_counts() if hasattr(opt, "host_counts") else None
        g1 = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread polls events while this thread captures; in the default (global) mode
        # that poll would invalidate the capture
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            loss = self._forward_backward(self.x, self.y)
            ops.join_side_streams()
            arena, big, sparse = self._sparse()
            xb = self._exchange_buffers(arena["g"], big, sparse)                # established by the warm-up steps
        base = xb["send_r"].data_ptr()
        for (urow, ugrad, _v), (r0, r1, n, dim, _v2) in zip(sparse, xb["offs"]):
            assert urow.data_ptr() == base + 4 * r0 and ugrad.data_ptr() == base + 4 * r1      # written in place
        g1b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1b, pool=g1.pool(), capture_error_mode="thread_local"):
            ops.run_late_jobs()
            ops.join_side_streams()
            if not xb["in_place"] and not xb["allreduce"]:
                xb["send_d"][:xb["A"]].copy_(arena["g"].reshape(-1))           # the only packing copy
        g2 = torch.cuda.CUDAGraph()
        for p, rg in zip(big, xb["merged"]):          # (the merge itself is launched eagerly, on the side stream, in replay)
            p._swr_sparse_grad = rg
            p._swr_sparse_local = False
        with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
            self._mean_dense(xb, arena["g"])
            self.trainer.optimizer.step()
        if snap is not None:
            opt.restore_host_counts(snap)      # the capture ran the optimizer's host code without executing a step
        self._graphs = (g1, g1b, g2, xb)
        self._big, self._arena_g = big, arena["g"]
        self._merge_stream = torch.cuda.Stream()
        self._rows_ready = torch.cuda.Event()
        self.loss = loss
        return self

    def _capture_one(self, snap):
        """The whole data-parallel step as ONE hipGraph, collectives included (SWR_DP_ONE_GRAPH=1): the rows all-gather is captured
        on a side stream (a parallel branch of the graph: it overlaps the held-back gradient work), the arena collective on the
        main stream behind that work, then merge + mean + optimizer.  One launch per step instead of three graph launches with
        eager collectives and cross-stream waits between them."""
        opt = self.trainer.optimizer
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            rows_ev = torch.cuda.Event() if ROWS_EVENT else None
            loss = self._forward_backward(self.x, self.y, rows_ev)
            ops.join_side_streams()
            arena, big, sparse = self._sparse()
            xb = self._exchange_buffers(arena["g"], big, sparse)                # established by the warm-up steps
            rows = bool(xb["offs"])
            # short batches (the step itself runs on ONE stream, ops.SIDE_MIN_BATCH): the exchange stays on that stream too.  A replayed
            # multi-stream graph costs the HOST ~8 us per node -- at the 8 192-row strong-scaling shard of config 2 the two-branch
            # step was issued in 0.23 ms per replay and ran in 0.26: host-bound -- and the branch only hides ~15 us of row exchange
            single = ONE_STREAM == "1" or (ONE_STREAM == "auto" and not ops.SIDE_STREAM)
            self._single_stream = single
            if rows and single:
                dist.all_gather_into_tensor(xb["recv_r"], xb["send_r"], group=self.group)
                self._merge_rows(xb, big)
            elif rows:
                if rows_ev is not None and ops.rows_event_recorded():
                    side.wait_event(rows_ev)            # the row lists alone: not the weight-gradient branch, not the small tables' sums
                else:
                    side.wait_stream(cur)
                with torch.cuda.stream(side):
                    dist.all_gather_into_tensor(xb["recv_r"], xb["send_r"], group=self.group)
                    self._merge_rows(xb, big)
                    # the optimizer's work on the merged lists that needs no dense gradient (catch-up of rows other ranks looked
                    # up, step bookkeeping) stays on this branch: the main stream's tail is collective -> mean -> update
                    if EARLY_ROWS and hasattr(opt, "early_rows"):
                        opt.early_rows(big)
            ops.run_late_jobs()
            ops.join_side_streams()
            self._send_dense(xb, arena["g"], pack=True)
            if rows and not single:
                cur.wait_stream(side)
            self._mean_dense(xb, arena["g"])
            opt.step()
        if snap is not None:
            opt.restore_host_counts(snap)
        self._graphs = (g, None, None, xb)
        self._big, self._arena_g = big, arena["g"]
        self.loss = loss
        return self

    def load(self, x_dict, y):
        from .trainers.graph import load_batch
        load_batch(self.x, self.y, x_dict, y)

    def replay(self):
        g1, g1b, g2, xb = self._graphs
        cur = torch.cuda.current_stream()
        if hasattr(self.trainer.optimizer, "note_replays"):
            self.trainer.optimizer.note_replays(1)
        model = self.trainer.model
        if getattr(self.trainer.optimizer, "clear_grads", False) and hasattr(model, "arena_dirty") and model.arena_dirty():
            model.zero_grad()              # (the captured step holds no fill: trainers/graph.py GraphedStep.replay)
        g1.replay()
        if g1b is None:                    # one graph holds the whole step (SWR_DP_ONE_GRAPH)
            return self.loss
        rows = bool(xb["offs"])
        if rows:
            # the row lists leave now, while the rest of the gradients is computed -- but the collective is ENQUEUED after the
            # second graph, from the merge stream, ordered behind an event recorded here: issued between the two graph
            # launches, its stream synchronisation kept the second graph's first kernel waiting ~30 us
            self._rows_ready.record(cur)
            g1b.replay()
            ops._skew(6)
            with torch.cuda.stream(self._merge_stream):
                self._merge_stream.wait_event(self._rows_ready)                 # (also: after the previous step's readers)
                ops._skew(5)
                work = dist.all_gather_into_tensor(xb["recv_r"], xb["send_r"], group=self.group, async_op=True)
                work.wait()
                self._merge_rows(xb, self._big)                                 # needs only this all-gather: hidden too
        else:
            g1b.replay()
        self._send_dense(xb, self._arena_g, pack=False)
        if rows:
            cur.wait_stream(self._merge_stream)
        g2.replay()
        return self.loss
