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


class GraphedStep(object):
    def __init__(self, trainer, x_dict, y, warmup=2, step_fn=None):
        self.trainer = trainer
        self.x = {k: v.clone() for k, v in x_dict.items()}
        self.y = y.clone()
        fn = step_fn if step_fn is not None else trainer.train_step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn(self.x, self.y)                      # result dropped at once: no autograd graph survives
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            loss = fn(self.x, self.y)
            ops.join_side_streams()        # nothing forked may outlive the capture (no-op after a full step)
        self.loss = loss.detach()
        del loss

    def load(self, x_dict, y):
        """Copy a new batch into the captured input buffers (same shapes and dtypes)."""
        for k, v in x_dict.items():
            self.x[k].copy_(v, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def replay(self):
        self.graph.replay()
        return self.loss
