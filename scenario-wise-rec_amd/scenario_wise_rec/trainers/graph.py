"""Whole-step hipGraph capture: one training step (forward, BCE, backward, optimizer) is ~60 kernel launches
of 5-400 us each; replaying them as ONE captured graph removes the per-launch host cost.

Capture rules learnt the hard way on ROCm 7.2 / torch 2.10 (see DESIGN.md "hipGraph"):
  * nothing from the warm-up steps may stay alive across the capture -- a live warm-up loss keeps its
    autograd graph and therefore the AccumulateGrad nodes of the leaves alive; the capture then reuses
    them on the (foreign) warm-up stream and the replayed gradient accumulation races with the rest of
    the step (wrong gradients, silently);
  * parameter gradients are written by the backward kernels straight into the gradient arena
    (ops._grad_alias), so the captured step contains no AccumulateGrad launches at all for the fused models;
  * inputs are static device tensors: feed new batches with `load(x, y)` (device-side copies).
"""
import torch

from .. import ops


def batch_signature(x_dict, y):
    """What a captured step is specific to: column names, shapes and dtypes of a batch."""
    return (tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in x_dict.items())), tuple(y.shape), y.dtype)


def load_batch(dst_x, dst_y, x_dict, y):
    """Batch -> static input buffers.  Device-resident 1-D columns (utils.data.DeviceDataLoader's row views) go in ONE
    launch for all columns and the label (swr_take_rows without a permutation); anything else column by column."""
    from .. import _hip as H
    pairs = [(x_dict[k], dst_x[k]) for k in dst_x] + [(y, dst_y)]
    n = dst_y.numel()
    if (n > 0 and len(pairs) <= 96 and all(s.is_cuda and s.dim() == 1 and s.is_contiguous() and s.numel() == n
                                          and s.dtype == d.dtype and s.element_size() in (1, 2, 4, 8) and d.numel() == n
                                          for s, d in pairs)):
        tab = (H.TakeColumn * len(pairs))()
        for j, (s, d) in enumerate(pairs):
            tab[j] = H.TakeColumn(s.data_ptr(), d.data_ptr(), s.element_size(), 0)
        H.check(H.lib.swr_take_rows(tab, len(pairs), None, n, n, H.ptr(H.err_flag(dst_y.device)), H.stream()),
                "swr_take_rows(copy)")
        return
    for s, d in pairs:
        d.copy_(s, non_blocking=True)


class GraphedStep(object):
    """`warmup` eager steps on (x, y) and then the capture -- or, with `warmup=0`, the capture alone: the caller has
    already run (at least two) ordinary steps of this shape, e.g. the first batches of CTRTrainer.train_one_epoch, and
    every batch is to be applied exactly once.  The capture itself executes nothing."""

    def __init__(self, trainer, x_dict, y, warmup=2, step_fn=None):
        self.trainer = trainer
        dev = y.device if y.is_cuda else trainer.device
        self.x = {k: v.to(dev, copy=True) for k, v in x_dict.items()}
        self.y = y.to(dev, copy=True)
        self.signature = batch_signature(x_dict, y)
        fn = step_fn if step_fn is not None else trainer.train_step
        if warmup > 0:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    fn(self.x, self.y)                  # result dropped at once: no autograd graph survives
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        opt = getattr(trainer, "optimizer", None)
        if hasattr(opt, "hist_cap") and 2 * opt._since_flush >= opt.hist_cap:
            opt.materialize()              # (a history-ring flush must not end up INSIDE the captured step)
        snap = opt.host_counts() if hasattr(opt, "host_counts") else None
        # with FusedAdam.clear_grads the captured step holds NO zero_grad fill (the update zeroes what it consumed): gradients
        # that an eager backward pass leaves behind between two replays would be added to -- replay() wipes them first
        model = getattr(trainer, "model", None)
        self._guard = model if (getattr(opt, "clear_grads", False) and hasattr(model, "arena_dirty")) else None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            loss = fn(self.x, self.y)
            ops.join_side_streams()        # nothing forked may outlive the capture (no-op after a full step)
        if snap is not None:
            opt.restore_host_counts(snap)  # the capture ran the optimizer's host code without executing a step
        self.loss = loss.detach()
        del loss

    def load(self, x_dict, y):
        """Copy a new batch into the captured input buffers (same shapes and dtypes)."""
        load_batch(self.x, self.y, x_dict, y)

    def replay(self):
        opt = getattr(self.trainer, "optimizer", None)
        if hasattr(opt, "note_replays"):
            opt.note_replays(1)            # host step count, scheduler lr, history-ring flush (optim.FusedAdam)
        if self._guard is not None and self._guard.arena_dirty():
            self._guard.zero_grad()        # (what the step's own zero_grad does when the loop is launched eagerly)
        self.graph.replay()
        return self.loss                   # the STATIC buffer: the next replay overwrites it (keep a `.clone()`, not this)
