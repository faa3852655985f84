"""FusedAdam: torch.optim.Adam(lr, betas, eps, weight_decay) semantics (the reference trainer's
optimizer, `trainers/ctr_trainer.py:50-52,73`) as HBM-streaming HIP kernels.

  * parameters that live back to back in a parameter arena (basic/module.py) with their gradients at
    matching offsets are updated by ONE launch per contiguous run;
  * large embedding tables with row-sparse gradients (ops.EmbedGather) take a row kernel for the looked-up
    rows and a sweep for all other rows (g = weight_decay * p) -- together exactly the dense update the
    reference performs on every row of every table every step;
  * parameters that took no gradient in the last backward are skipped entirely, like torch
    (`grad is None`: no decay, no state), e.g. PPNet's agnostic tables;
  * step count and bias corrections live in device memory (a captured hipGraph replays correctly).
"""
import bisect
import ctypes as C
import os
import sys

import torch

from . import _hip as H
from ._hip import lib


HIST_CAP = 1 << 20          # steps of (step_size, inv_bc2_sqrt) history kept on the device (8 MB): a ring -- every
                            # lazily updated table is flushed (exactly) before a row could lag that many steps
_EARLY_ADVANCED = set()     # hyper buffers whose step counter was advanced ahead of step() (advance_early)


class LazyRows(object):
    """Optimizer state of one large table under the exact lazy update (csrc/adam.hip, "lazy (exact) row updates"):
    `last[row]` = the step the row is current for.  Attached to the parameter as `_swr_lazy`, so the forward lookup
    can bring the rows it is about to read up to date, and `state_dict()` can materialise the table."""

    def __init__(self, p, hist, hyper):
        self.p = p
        self.m = torch.zeros_like(p)
        self.v = torch.zeros_like(p)
        self.last = torch.zeros(p.shape[0], dtype=torch.int32, device=p.device)
        self.claim = torch.empty(p.shape[0], dtype=torch.int32, device=p.device)      # ticket scratch of the catch-up
        self.hist, self.hyper = hist, hyper

    def catchup(self, idx, hash_seed=0):
        """Replay the pending decay-only updates of the rows `idx` refers to (ids, any integer dtype; -1 = skip)."""
        if self.hyper.data_ptr() in _EARLY_ADVANCED:
            raise H.SwrError("a lookup of a lazily updated table after FusedAdam.advance_early(): the step counter already "
                             "counts the coming update (use advance_early only with one lookup per step)")
        idx = idx.contiguous()
        n = idx.numel()
        ws = torch.empty(max(n, 1) * 8, dtype=torch.uint8, device=idx.device)
        p = self.p
        H.check(lib.swr_adam_catchup_rows(H.ptr(p), H.ptr(self.m), H.ptr(self.v), H.ptr(self.last), H.ptr(self.claim), p.shape[0], p.shape[1],
                                          H.ptr(idx), H.dtype_code(idx), hash_seed, n, H.ptr(self.hist), H.ptr(self.hyper),
                                          H.ptr(ws), ws.numel(), H.stream()), "swr_adam_catchup_rows")

    def flush(self):
        """Materialise every row (checkpoints, direct reads of the table)."""
        p = self.p
        H.check(lib.swr_adam_flush(H.ptr(p), H.ptr(self.m), H.ptr(self.v), H.ptr(self.last), p.shape[0], p.shape[1],
                                   H.ptr(self.hist), H.ptr(self.hyper), H.stream()), "swr_adam_flush")


def catchup_many(items):
    """`LazyRows.catchup` for several tables: items = [(lazy, idx, hash_seed)].  Tables that share an optimizer group
    (one history, one set of step scalars) go through ONE claim launch and ONE replay launch (swr_adam_catchup_multi)
    instead of two launches per table."""
    groups = {}
    for lazy, idx, seed in items:
        groups.setdefault((lazy.hist.data_ptr(), lazy.hyper.data_ptr()), []).append((lazy, idx, seed))
    for members_all in groups.values():
        # a table looked up through several id columns (`shared_with`) enters the multi launch ONCE: two entries with the
        # same claim / last arrays could both win the ticket of a row (tickets are table-local positions) and replay its
        # pending steps twice.  The repeats run as ordinary catch-ups behind it (ordered by the stream: exact).
        members, repeats, seen = [], [], set()
        for mbr in members_all:
            (repeats if id(mbr[0]) in seen else members).append(mbr)
            seen.add(id(mbr[0]))
        if len(members) == 1:
            members[0][0].catchup(members[0][1], members[0][2])
            for lazy, idx, seed in repeats:
                lazy.catchup(idx, seed)
            continue
        lazy0 = members[0][0]
        if lazy0.hyper.data_ptr() in _EARLY_ADVANCED:
            raise H.SwrError("a lookup of a lazily updated table after FusedAdam.advance_early(): the step counter already "
                             "counts the coming update (use advance_early only with one lookup per step)")
        for c0 in range(0, len(members), H.ADAM_MAX_TABLES):
            chunk = members[c0:c0 + H.ADAM_MAX_TABLES]
            idxs = [idx.contiguous() for _l, idx, _s in chunk]
            total = sum(max(i.numel(), 1) for i in idxs)
            ws = torch.empty(total * 2, dtype=torch.int32, device=idxs[0].device)
            tabs = (H.AdamTable * len(chunk))()
            off = 0
            for j, ((lazy, _i, seed), idx) in enumerate(zip(chunk, idxs)):
                n, p = idx.numel(), lazy.p
                tabs[j] = H.AdamTable(p.data_ptr(), lazy.m.data_ptr(), lazy.v.data_ptr(), lazy.last.data_ptr(), lazy.claim.data_ptr(),
                                      p.shape[0], p.shape[1], H.dtype_code(idx), idx.data_ptr(), seed, 0, n,
                                      ws.data_ptr() + 4 * off, ws.data_ptr() + 4 * (off + max(n, 1)), None, None)
                off += 2 * max(n, 1)
            H.check(lib.swr_adam_catchup_multi(tabs, len(chunk), H.ptr(lazy0.hist), H.ptr(lazy0.hyper), H.stream()),
                    "swr_adam_catchup_multi")
        for lazy, idx, seed in repeats:
            lazy.catchup(idx, seed)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, lazy_rows=True, hist_cap=HIST_CAP):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.lazy_rows = lazy_rows
        # True: the dense update zeroes each gradient it consumed (SwrModule.zero_grad then skips its fill).  For loops that
        # call zero_grad -> backward -> step and never read `.grad` after the step (CTRTrainer sets it); the default keeps
        # torch's behaviour: gradients stay readable until the next zero_grad
        self.clear_grads = False
        if hist_cap < 4 or hist_cap & (hist_cap - 1):
            raise ValueError("hist_cap must be a power of two >= 4")
        self.hist_cap = int(hist_cap)
        self._since_flush = 0   # optimizer steps (eager or replayed) since every lazy table was last fully current
        self._hyper = {}        # group index -> [device swr_adam_hyper, uploaded host copy, hist, host step count]
        self._mv = {}           # storage ptr -> (m_flat, v_flat) shadowing a parameter storage
        self._big = {}          # id(param) -> (m, v, bitmap)   (sweep mode)  |  LazyRows  (lazy mode)

    # ---- device-side hyper-parameters ---------------------------------------------------------------
    def _hyper_dev(self, gi, group, device):
        want = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                float(group["weight_decay"]))
        ent = self._hyper.get(gi)
        if ent is None:
            h = H.AdamHyper(*want, 0)
            buf = torch.frombuffer(bytearray(bytes(h)), dtype=torch.uint8).to(device)
            hist = torch.zeros((self.hist_cap, 2), dtype=torch.float32, device=device) if self.lazy_rows else None
            self._hyper[gi] = [buf, want, hist, 0]
        elif ent[1] != want:
            if self.lazy_rows and ent[1][1:] != want[1:]:
                # betas / eps / weight_decay changed: pending lazy updates belong to the old values
                self.materialize()
            # lr (scheduler) or another hyper-parameter changed on the host: refresh the five doubles, keep the step
            host = torch.tensor(want, dtype=torch.float64).view(torch.uint8)
            ent[0][:40].copy_(host.to(device))
            ent[1] = want
        return self._hyper[gi][0]

    def _state_for(self, p):
        """(m, v) views shadowing p inside per-storage flat state buffers (so adjacency carries over)."""
        st = p.untyped_storage()
        key = st.data_ptr()
        if key not in self._mv:
            n = st.nbytes() // 4
            self._mv[key] = (torch.zeros(n, dtype=torch.float32, device=p.device),
                             torch.zeros(n, dtype=torch.float32, device=p.device))
        off = (p.data_ptr() - key) // 4
        m, v = self._mv[key]
        return m[off:off + p.numel()], v[off:off + p.numel()], key

    def _lazy_state(self, p, hist, hyper):
        st = self._big.get(id(p))
        if st is None:
            st = self._big[id(p)] = LazyRows(p, hist, hyper)
            p._swr_lazy = st
        return st

    @torch.no_grad()
    def advance_early(self):
        """Advance the step counter / bias corrections NOW (a 1-thread launch) so that `step()` need not: the trainer runs
        this on the side stream during the forward pass, after the forward's own catch-up of the rows it reads.  Only
        with one parameter group whose device scalars exist (i.e. from the second step on); `step()` notices."""
        if len(self.param_groups) != 1 or 0 not in self._hyper or getattr(self, "_advanced", False):
            return
        ent = self._hyper[0]
        if self.lazy_rows and self._since_flush + 3 >= self.hist_cap:
            return                                             # step() flushes the lazy tables first, then advances
        hyper = self._hyper_dev(0, self.param_groups[0], ent[0].device)
        hist = ent[2]
        H.check(lib.swr_adam_advance(H.ptr(hyper), H.ptr(hist), self.hist_cap if hist is not None else 0, H.stream()),
                "swr_adam_advance")
        ent[3] += 1
        self._since_flush += 1
        self._advanced = True
        _EARLY_ADVANCED.add(hyper.data_ptr())

    @torch.no_grad()
    def early_rows(self, params):
        """Data-parallel step, on the exchange's side branch right behind the row merge: what `step()` does in FRONT of its update
        launches -- the catch-up of the merged row lists' rows that only other ranks looked up, then the step bookkeeping -- so that
        the main stream's tail behind the last gradient is collective -> mean -> update.  `params`: the row-sparse tables whose merged
        lists (`_swr_sparse_grad`, `_swr_sparse_local = False`) are in place.  Same launches, same order between them; `step()`
        notices (`_advanced`, `_behind_done`).  No-op where advance_early() would not advance either."""
        if len(self.param_groups) != 1 or 0 not in self._hyper or getattr(self, "_advanced", False):
            return False
        ent = self._hyper[0]
        if self.lazy_rows and self._since_flush + 3 >= self.hist_cap:
            return False
        group = self.param_groups[0]
        hyper = self._hyper_dev(0, group, ent[0].device)
        hist = ent[2]
        if self.lazy_rows:
            behind = []
            for p in params:
                sg = getattr(p, "_swr_sparse_grad", None)
                if sg is None:
                    continue
                st = self._lazy_state(p, hist, hyper)
                if not getattr(p, "_swr_sparse_local", False):
                    behind.append((st, sg[0], 0))
            if behind:
                catchup_many(behind)
        H.check(lib.swr_adam_advance(H.ptr(hyper), H.ptr(hist), self.hist_cap if hist is not None else 0, H.stream()),
                "swr_adam_advance")
        ent[3] += 1
        self._since_flush += 1
        self._advanced = True
        self._behind_done = True
        return True

    @torch.no_grad()
    def advance_rider(self):
        """`advance_early` without its launch: the same bookkeeping on the host, and the device arguments for the launch that will
        carry the advance as a rider (the step's fused loss launch, ops.take_loss_rider) -- or None where advance_early would
        not advance either."""
        if len(self.param_groups) != 1 or 0 not in self._hyper or getattr(self, "_advanced", False):
            return None
        ent = self._hyper[0]
        if self.lazy_rows and self._since_flush + 3 >= self.hist_cap:
            return None
        hyper = self._hyper_dev(0, self.param_groups[0], ent[0].device)
        hist = ent[2]
        ent[3] += 1
        self._since_flush += 1
        self._advanced = True
        _EARLY_ADVANCED.add(hyper.data_ptr())
        return hyper, hist, (self.hist_cap if hist is not None else 0)

    @torch.no_grad()
    def materialize(self):
        """Bring every lazily updated table fully up to date (exact); cheap no-op when nothing is pending."""
        for st in self._big.values():
            if isinstance(st, LazyRows):
                st.flush()
        self._since_flush = 0

    @torch.no_grad()
    def note_replays(self, n=1):
        """Called by whoever replays a captured step (trainers/graph.py, parallel.DataParallelStep) BEFORE the replay:
        the device-side step counter advances inside the graph where the host cannot see it.  Keeps the host's step
        count right (state_dict) and flushes the lazily updated tables in-stream before the history ring wraps."""
        if self.lazy_rows and self._since_flush + n + 2 >= self.hist_cap:
            self.materialize()
        self._since_flush += n
        for gi, ent in self._hyper.items():
            ent[3] += n
            self._hyper_dev(gi, self.param_groups[gi], ent[0].device)      # a scheduler's new lr reaches the device scalars

    def host_counts(self):
        """Host-side step bookkeeping (a capture runs step() on the host without executing it: snapshot / restore)."""
        return (self._since_flush, {gi: ent[3] for gi, ent in self._hyper.items()}, getattr(self, "_advanced", False),
                getattr(self, "_behind_done", False))

    def restore_host_counts(self, snap):
        self._since_flush = snap[0]
        for gi, n in snap[1].items():
            self._hyper[gi][3] = n
        self._advanced = snap[2]
        self._behind_done = snap[3] if len(snap) > 3 else False

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = H.stream()
        for gi, group in enumerate(self.param_groups):
            dense, sparse = [], []
            for p in group["params"]:
                sg = getattr(p, "_swr_sparse_grad", None)
                if sg is not None:
                    sparse.append((p, sg))
                elif p.grad is not None and getattr(p, "_swr_touched", True):
                    dense.append(p)
            if not dense and not sparse:
                continue
            dev = (dense[0] if dense else sparse[0][0]).device
            H.require_device(*(dense[:1] + [s[0] for s in sparse[:1]]))
            hyper = self._hyper_dev(gi, group, dev)
            ent = self._hyper[gi]
            hist = ent[2]
            if self.lazy_rows:
                behind = []
                done = gi == 0 and getattr(self, "_behind_done", False)      # early_rows() caught them up already
                for p, (urow, ugrad) in sparse:
                    # rows that take a gradient must be current BEFORE the step advances: those looked up by this
                    # rank were caught up by the forward lookup, those that only other ranks touched are caught up here
                    st = self._lazy_state(p, hist, hyper)        # (state must exist before the row kernel)
                    if not getattr(p, "_swr_sparse_local", False) and not done:
                        behind.append((st, urow, 0))
                if behind:
                    catchup_many(behind)
            if gi == 0:
                self._behind_done = False
            if gi == 0 and getattr(self, "_advanced", False):
                self._advanced = False                         # advance_early() / early_rows() already did it for this step
                _EARLY_ADVANCED.discard(hyper.data_ptr())
            else:
                if self.lazy_rows and gi == 0 and self._since_flush + 3 >= self.hist_cap:
                    self.materialize()                         # the history is a ring: no row may lag hist_cap - 1 steps
                H.check(lib.swr_adam_advance(H.ptr(hyper), H.ptr(hist), self.hist_cap if hist is not None else 0, stream),
                        "swr_adam_advance")
                ent[3] += 1
                if gi == 0:
                    self._since_flush += 1
            # contiguous runs: parameter, gradient and state addresses all advance together
            items = []
            clear_g = 1 if self.clear_grads else 0
            for p in dense:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise H.SwrError("FusedAdam needs contiguous fp32 parameters")
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                m, v, key = self._state_for(p)
                items.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), key, g))
                if clear_g and g is p.grad:
                    p._swr_grad_clean = True          # zeroed behind the update (SwrModule.zero_grad skips its fill)
            items.sort(key=lambda t: t[0])
            # parameters of this optimizer that are NOT updated this step (no gradient / untouched / frozen): a run must
            # not sweep over them (torch skips them entirely: no decay, no state)
            updated = {it[0] for it in items}
            blockers = sorted(q.data_ptr() for g2 in self.param_groups for q in g2["params"] if q.data_ptr() not in updated)
            runs = []
            for it in items:
                if runs:
                    r = runs[-1]
                    end = r[0] + 4 * r[4]
                    gap = it[0] - end
                    j = bisect.bisect_left(blockers, end)
                    clear = j >= len(blockers) or blockers[j] >= it[0]
                    # merge across the arena's alignment padding only (< 16 bytes of zeros with zero gradient and zero
                    # state: the update is a no-op there) and never across another parameter
                    if (0 <= gap < 16 and clear and it[5] == r[5] and it[1] - r[1] == it[0] - r[0] and it[2] - r[2] == it[0] - r[0]):
                        r[4] = (it[0] - r[0]) // 4 + it[4]
                        r[6].append(it[6])
                        continue
                runs.append([it[0], it[1], it[2], it[3], it[4], it[5], [it[6]]])
            if self.lazy_rows and len(runs) == 1 and len(sparse) == 1 and runs[0][4] > 0 and sparse[0][1][0].numel() > 0:
                # one arena + one large table (config 2): both updates in one launch
                p_ptr, g_ptr, m_ptr, v_ptr, n, _key, _keep = runs[0]
                p, (urow, ugrad) = sparse[0]
                st = self._lazy_state(p, hist, hyper)
                H.check(lib.swr_adam_dense_rows(C.c_void_p(p_ptr), C.c_void_p(g_ptr), C.c_void_p(m_ptr), C.c_void_p(v_ptr), n, clear_g,
                                                H.ptr(p), H.ptr(st.m), H.ptr(st.v), p.shape[0], p.shape[1], H.ptr(urow),
                                                H.ptr(ugrad), urow.numel(), H.ptr(st.last), H.ptr(hyper), stream),
                        "swr_adam_dense_rows")
                continue
            for p_ptr, g_ptr, m_ptr, v_ptr, n, _key, _keep in runs:
                H.check(lib.swr_adam_dense(C.c_void_p(p_ptr), C.c_void_p(g_ptr), C.c_void_p(m_ptr), C.c_void_p(v_ptr),
                                           n, clear_g, H.ptr(hyper), stream), "swr_adam_dense")
            if self.lazy_rows and len(sparse) > 1:
                # several large tables: one row-update launch for all of them (swr_adam_rows_multi)
                for c0 in range(0, len(sparse), H.ADAM_MAX_TABLES):
                    chunk = sparse[c0:c0 + H.ADAM_MAX_TABLES]
                    tabs = (H.AdamTable * len(chunk))()
                    for j, (p, (urow, ugrad)) in enumerate(chunk):
                        st = self._lazy_state(p, hist, hyper)
                        tabs[j] = H.AdamTable(p.data_ptr(), st.m.data_ptr(), st.v.data_ptr(), st.last.data_ptr(), None, p.shape[0],
                                              p.shape[1], 0, None, 0, 0, urow.numel(), None, None, urow.data_ptr(), ugrad.data_ptr())
                    H.check(lib.swr_adam_rows_multi(tabs, len(chunk), H.ptr(hyper), stream), "swr_adam_rows_multi")
                sparse = []
            for p, (urow, ugrad) in sparse:
                if self.lazy_rows:
                    st = self._lazy_state(p, hist, hyper)
                    H.check(lib.swr_adam_rows(H.ptr(p), H.ptr(st.m), H.ptr(st.v), p.shape[0], p.shape[1], H.ptr(urow),
                                              H.ptr(ugrad), urow.numel(), None, H.ptr(st.last), H.ptr(hyper), stream),
                            "swr_adam_rows")
                    continue
                if id(p) not in self._big:
                    self._big[id(p)] = (torch.zeros_like(p), torch.zeros_like(p),
                                        torch.zeros((p.shape[0] + 31) // 32, dtype=torch.int32, device=p.device))
                m, v, bitmap = self._big[id(p)]
                H.check(lib.swr_adam_rows(H.ptr(p), H.ptr(m), H.ptr(v), p.shape[0], p.shape[1], H.ptr(urow), H.ptr(ugrad),
                                          urow.numel(), H.ptr(bitmap), None, H.ptr(hyper), stream), "swr_adam_rows")
                H.check(lib.swr_adam_sweep_untouched(H.ptr(p), H.ptr(m), H.ptr(v), p.shape[0], p.shape[1], H.ptr(bitmap), 1,
                                                     H.ptr(hyper), stream), "swr_adam_sweep_untouched")
        return loss

    # ---- checkpointing: torch.optim.Adam's format (state[i] = {step, exp_avg, exp_avg_sq}) ----------------------
    def _state_views(self, p):
        """(m, v) of parameter p if it has optimizer state, else None."""
        st = self._big.get(id(p))
        if isinstance(st, LazyRows):
            return st.m, st.v
        if st is not None:
            return st[0], st[1]
        key = p.untyped_storage().data_ptr()
        if key in self._mv:
            m, v, _ = self._state_for(p)
            return m.view(p.shape), v.view(p.shape)
        return None

    @torch.no_grad()
    def state_dict(self):
        """The layout of `torch.optim.Adam.state_dict()`: interchangeable with the reference's optimizer.  Lazily updated
        tables are materialised first (exact).  The step count is per parameter group here (parameters that never took
        a gradient have no entry, like torch)."""
        self.materialize()
        sd = super().state_dict()          # param_groups with packed indices; `state` is empty (kept outside self.state)
        state, i = {}, 0
        for gi, group in enumerate(self.param_groups):
            steps = self._hyper[gi][3] if gi in self._hyper else 0
            for p in group["params"]:
                mv = self._state_views(p) if steps else None
                if mv is not None:
                    state[i] = {"step": torch.tensor(float(steps)), "exp_avg": mv[0].detach().clone(),
                                "exp_avg_sq": mv[1].detach().clone()}
                i += 1
        sd["state"] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(g["params"]) != len(h["params"]) for g, h in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has different parameter groups")
        self.materialize()
        i = 0
        for gi, (src, group) in enumerate(zip(groups, self.param_groups)):
            for k, val in src.items():
                if k != "params":
                    group[k] = val
            steps = set()
            for p in group["params"]:
                ent = sd["state"].get(i, sd["state"].get(str(i)))
                i += 1
                if ent is None:
                    continue
                steps.add(int(float(ent["step"])))
                big = getattr(p, "_swr_row_sparse", False) or getattr(p, "_swr_lazy", None) is not None
                if big and self.lazy_rows:
                    hyper = self._hyper_dev(gi, group, p.device)
                    st = self._lazy_state(p, self._hyper[gi][2], hyper)
                    m, v = st.m, st.v
                elif big:
                    if id(p) not in self._big:
                        self._big[id(p)] = (torch.zeros_like(p), torch.zeros_like(p),
                                            torch.zeros((p.shape[0] + 31) // 32, dtype=torch.int32, device=p.device))
                    m, v = self._big[id(p)][:2]
                else:
                    m, v, _ = self._state_for(p)
                    m, v = m.view(p.shape), v.view(p.shape)
                m.copy_(ent["exp_avg"].to(p.device, torch.float32))
                v.copy_(ent["exp_avg_sq"].to(p.device, torch.float32))
            if len(steps) > 1:
                raise H.SwrError("FusedAdam keeps one step count per parameter group; the loaded state has several: %s" % sorted(steps))
            if steps:
                n = steps.pop()
                dev = next(p.device for p in group["params"])
                hyper = self._hyper_dev(gi, group, dev)
                hyper[40:48].copy_(torch.tensor([n], dtype=torch.int64).view(torch.uint8).to(dev))     # swr_adam_hyper.step
                self._hyper[gi][3] = n
                for st in self._big.values():
                    if isinstance(st, LazyRows) and st.hyper.data_ptr() == hyper.data_ptr():
                        st.last.fill_(n)            # every row is current as of the loaded step
        self._since_flush = 0
        self._advanced = False
