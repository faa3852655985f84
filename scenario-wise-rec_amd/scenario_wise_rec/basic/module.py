"""Parameter arena: the HBM layout of a model's dense state.

Every fp32 parameter of a model (small embedding tables included) lives in ONE flat device buffer,
its gradient at the same offset of a second flat buffer, and the BatchNorm buffers in a third.  The
nn.Parameters of the reference-compatible module tree are views into those buffers, so
`state_dict()` keys and shapes are unchanged while

  * layers that are evaluated together (all experts + gates of an MMoE, all towers ...) have their
    weights back to back, so one GEMM reads them as a single stacked matrix without a concat copy;
  * the optimizer step is one streaming kernel over the arena (optim.FusedAdam);
  * the data-parallel gradient exchange is ONE all-reduce of the gradient arena (parallel.py).

Embedding tables above `SwrModule.dense_table_limit_bytes` stay outside: their gradients are
row-sparse (`ops.EmbedGather`), their update is a row kernel + an untouched-row sweep.
"""
import os

import torch
import torch.nn as nn

_ALIGN = 4      # elements: every arena member group starts 16-byte aligned


class SwrModule(nn.Module):
    # tables above this size take row-sparse gradients (row lists + exact lazy Adam) instead of a dense [V, E] gradient
    # that is zeroed, swept and Adam-stepped in full every step; SWR_DENSE_TABLE_LIMIT (bytes) overrides
    dense_table_limit_bytes = int(os.environ.get("SWR_DENSE_TABLE_LIMIT", 1 << 20))
    _apply_depth = 0

    # ---- layout ------------------------------------------------------------------------------
    def _fused_groups(self):
        """Lists of tensors (parameters or buffers) that must be adjacent in the arena, in order."""
        return []

    def _apply(self, fn, *args, **kwargs):
        SwrModule._apply_depth += 1
        try:
            super()._apply(fn, *args, **kwargs)
        finally:
            SwrModule._apply_depth -= 1
        if SwrModule._apply_depth == 0:
            self.build_arena()
        return self

    def build_arena(self):
        """(Re)home all dense state of this module tree into flat buffers.  Idempotent."""
        params = [p for p in self.parameters() if p.dtype == torch.float32]
        if not params:
            return self
        dev = params[0].device
        groups = []
        for m in self.modules():
            if isinstance(m, SwrModule):
                groups.extend(m._fused_groups())
        is_table = {id(m.weight) for m in self.modules() if isinstance(m, nn.Embedding)}
        big = {id(p) for p in params if id(p) in is_table and p.numel() * 4 > self.dense_table_limit_bytes}

        def layout(tensors, pre_groups, dtype):
            seen, order = set(), []
            for g in pre_groups:
                g = [t for t in g if t.dtype == dtype and id(t) not in seen and any(t is u for u in tensors)]
                if g:
                    seen.update(id(t) for t in g)
                    order.append(g)
            order.extend([t] for t in tensors if id(t) not in seen)
            total, spans = 0, []
            for g in order:
                total = (total + _ALIGN - 1) // _ALIGN * _ALIGN
                for t in g:
                    spans.append((t, total, t.numel()))
                    total += t.numel()
            flat = torch.zeros(max(total, 1), dtype=dtype, device=dev)
            for t, off, n in spans:
                view = flat[off:off + n].view(t.shape)
                view.copy_(t.detach())
                t.data = view
            return flat, spans

        dense = [p for p in params if id(p) not in big and p.device == dev]
        p_flat, p_spans = layout(dense, groups, torch.float32)
        g_flat = torch.zeros_like(p_flat)
        for p, off, n in p_spans:
            p.grad = g_flat[off:off + n].view(p.shape) if p.requires_grad else None
        fbufs = [b for b in self.buffers() if b.dtype == torch.float32 and b.device == dev]
        ibufs = [b for b in self.buffers() if b.dtype == torch.int64 and b.device == dev]
        b_flat, _ = layout(fbufs, groups, torch.float32)
        i_flat, _ = layout(ibufs, groups, torch.int64)
        self._swr_arena = {"p": p_flat, "g": g_flat, "spans": p_spans, "b": b_flat, "i": i_flat,
                           "big": [p for p in params if id(p) in big]}
        for p in params:
            p._swr_row_sparse = id(p) in big       # (optim.FusedAdam.load_state_dict: which state layout the table takes)
        self._install_touch_hooks(params)
        return self

    def _install_touch_hooks(self, params):
        """Record which parameters took a gradient in the last backward: Adam must skip the others
        entirely (torch leaves their `.grad` None: PPNet's agn tables, ppnet.py:54)."""
        for p in params:
            if p.requires_grad and not hasattr(p, "_swr_hooked"):
                p._swr_hooked = True
                p._swr_touched = False
                p.register_post_accumulate_grad_hook(_mark_touched)

    def arena(self):
        a = getattr(self, "_swr_arena", None)
        if a is None:
            return None
        for p, off, n in a["spans"]:           # still consistent? (tables shared with another model may have moved)
            if p.data_ptr() != a["p"].data_ptr() + 4 * off:
                return None
        return a

    def materialize(self):
        """Bring lazily updated tables (optim.LazyRows) fully up to date; exact, no-op when nothing is pending."""
        for p in self.parameters():
            lazy = getattr(p, "_swr_lazy", None)
            if lazy is not None:
                lazy.flush()
        return self

    def state_dict(self, *args, **kwargs):
        self.materialize()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        # rows of a lazily updated table that are behind still owe their pending decay-only steps: apply those to the OLD
        # values first (exact), so that afterwards every row is current and the loaded values are what the next step sees
        # -- the reference's behaviour when weights are loaded under a live optimizer
        self.materialize()
        return super().load_state_dict(state_dict, *args, **kwargs)

    def set_dense_table_limit(self, nbytes):
        """Per-model override of `dense_table_limit_bytes` (tables above it take row-sparse gradients that only
        optim.FusedAdam consumes); re-homes the arena."""
        self.materialize()
        for m in self.modules():
            if isinstance(m, SwrModule):
                m.dense_table_limit_bytes = int(nbytes)
        for p in self.parameters():
            if getattr(p, "_swr_lazy", None) is not None:
                raise RuntimeError("set_dense_table_limit after the optimizer created lazy-row state for a table")
        return self.build_arena()

    def arena_dirty(self):
        """True when a parameter of the gradient arena took a gradient that no optimizer step consumed and zeroed
        (optim.FusedAdam.clear_grads); None without an arena."""
        a = self.arena()
        if a is None:
            return None
        return any(getattr(p, "_swr_touched", True) and not getattr(p, "_swr_grad_clean", False)
                   for p, _off, _n in a["spans"] if p.requires_grad)

    def zero_grad(self, set_to_none=True):
        a = self.arena()
        if a is None:
            return super().zero_grad(set_to_none)
        g = a["g"]
        # nothing to fill when every parameter that took a gradient since the last call had it consumed AND zeroed by the
        # optimizer (optim.FusedAdam.clear_grads): the arena was all zeros before that backward pass and is again
        dirty = self.arena_dirty()
        if not dirty:
            pass
        elif g.is_cuda and g.data_ptr() % 16 == 0:
            from .. import _hip as H
            H.check(H.lib.swr_zero(H.ptr(g), g.numel() * 4, H.stream()), "swr_zero")
        else:
            g.zero_()
        for p, off, n in a["spans"]:
            if p.requires_grad:
                if p.grad is None or p.grad.data_ptr() != a["g"].data_ptr() + 4 * off:
                    p.grad = a["g"][off:off + n].view(p.shape)
                p._swr_grad_clean = False
                if hasattr(p, "_swr_hooked"):
                    p._swr_touched = False
        for p in a["big"]:
            p.grad = None
            p._swr_sparse_grad = None
            p._swr_touched = False
        return dirty           # (ops.add_side_job: False = no launch, nothing for the backward pass to wait for)


def _mark_touched(p):
    p._swr_touched = True
    p._swr_grad_clean = False
