"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI).  Replaces the reference's single-process `torch.nn.DataParallel`
(`trainers/ctr_trainer.py:45-47`) and keeps ITS semantics:

  * the batch shards by row; every rank holds a full replica of all tables and parameters;
  * BatchNorm statistics are per shard (no SyncBN); rank 0's running statistics are the model's;
  * the loss is the mean over the GLOBAL batch: every rank back-propagates its local mean loss and the
    gradients are averaged (equal shards), which equals the sum of the reference's per-replica gradients;
  * one exchange step per training step, two collectives:
      1. dense: ONE all-reduce of the flat gradient arena (all non-table parameters and all small tables;
         0.5 MB at the KuaiRand config -- latency-bound, so a single call, not one per tensor);
      2. sparse: for each large table, all-gather of the per-rank row-reduced entries (row id, summed
         gradient), then the same deterministic segmented reduction as the local backward over the
         gathered entries -- every rank computes bit-identical row gradients, so replicas never drift
         (there is no parameter broadcast after step 0).
  * the optimizer runs replicated.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import ops


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


def communicate(dense_grad, sparse_grads, world_size, group=None, buffers=None):
    """The collectives of the exchange step, nothing else: one all-reduce of the flat gradient arena and, per large
    table, all-gathers of the per-rank (row id, gradient) entries.  Returns the gathered (rows, grads, vocab) list.
    `buffers` (optional) are preallocated gather targets [(rows, grads), ...] -- static addresses for hipGraph use."""
    if dense_grad is not None and dense_grad.numel():
        dist.all_reduce(dense_grad, op=dist.ReduceOp.SUM, group=group)
    gathered = []
    for t, (urow, ugrad, vocab) in enumerate(sparse_grads):
        n, dim = ugrad.shape
        if buffers is not None:
            rows, grads = buffers[t]
        else:
            rows = torch.empty(world_size * n, dtype=urow.dtype, device=urow.device)
            grads = torch.empty((world_size * n, dim), dtype=ugrad.dtype, device=ugrad.device)
        dist.all_gather_into_tensor(rows, urow.contiguous(), group=group)
        dist.all_gather_into_tensor(grads, ugrad.contiguous(), group=group)
        gathered.append((rows, grads, vocab))
    return gathered


def finish(dense_grad, gathered, world_size, merge_rows=hip_merge_rows):
    """Device-side remainder of the exchange: gradient averaging and the deterministic merge of the gathered row
    entries (every rank computes bit-identical results)."""
    if dense_grad is not None and dense_grad.numel():
        dense_grad.mul_(1.0 / world_size)
    merged = []
    for rows, grads, vocab in gathered:
        r, g = merge_rows(rows, grads, vocab)
        merged.append((r, g.mul_(1.0 / world_size)))
    return merged


def exchange_gradients(dense_grad, sparse_grads, world_size, group=None, merge_rows=hip_merge_rows):
    """The exchange step.  `dense_grad`: the flat gradient arena (averaged in place).
    `sparse_grads`: list of (urow int32 [n], ugrad fp32 [n, dim], vocab) per large table.
    Returns the merged, averaged (urow, ugrad) per table."""
    return finish(dense_grad, communicate(dense_grad, sparse_grads, world_size, group), world_size, merge_rows)


class DataParallelStep(object):
    """`CTRTrainer.train_step` with the gradient exchange between backward and the optimizer step.

    `train_step` launches everything eagerly.  `capture(x, y)` records the step as TWO hipGraphs -- forward +
    backward, and merge + optimizer -- with the RCCL collectives issued eagerly in between (3 calls at the
    KuaiRand config), so a replayed step costs two graph launches and the collectives instead of ~60 launches."""

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
    def _forward_backward(self, x_dict, y):
        return self.trainer.forward_backward(x_dict, y).detach()

    def _sparse(self):
        arena = self.trainer.model.arena()
        big = [p for p in arena["big"] if getattr(p, "_swr_sparse_grad", None) is not None]
        return arena, big, [(p._swr_sparse_grad[0], p._swr_sparse_grad[1], p.shape[0]) for p in big]

    def train_step(self, x_dict, y):
        tr = self.trainer
        model = tr.model
        loss = self._forward_backward(x_dict, y)
        arena = model.arena() if hasattr(model, "arena") else None
        if arena is None:
            # no arena (foreign module): per-tensor exchange
            for p in model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad, group=self.group)
                    p.grad.mul_(1.0 / self.world_size)
        else:
            arena, big, sparse = self._sparse()
            merged = exchange_gradients(arena["g"], sparse, self.world_size, self.group)
            for p, rg in zip(big, merged):
                p._swr_sparse_grad = rg
        tr.optimizer.step()
        return loss

    # ---- captured variant -----------------------------------------------------------------------------
    def capture(self, x_dict, y, warmup=2):
        """Static input buffers + two captured graphs.  Feed batches with `load`, run with `replay`."""
        self.x = {k: v.clone() for k, v in x_dict.items()}
        self.y = y.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.train_step(self.x, self.y)          # results dropped at once (see trainers/graph.py)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            loss = self._forward_backward(self.x, self.y)
            ops.join_side_streams()
        arena, big, sparse = self._sparse()
        buffers = [(torch.empty(self.world_size * r.numel(), dtype=r.dtype, device=r.device),
                    torch.empty((self.world_size * g.shape[0], g.shape[1]), dtype=g.dtype, device=g.device))
                   for r, g, _ in sparse]
        gathered = [(rows, grads, v) for (rows, grads), (_, _, v) in zip(buffers, sparse)]
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=g1.pool()):
            merged = finish(arena["g"], gathered, self.world_size)
            for p, rg in zip(big, merged):
                p._swr_sparse_grad = rg
            self.trainer.optimizer.step()
        self._graphs = (g1, g2, arena["g"], sparse, buffers)
        self.loss = loss
        return self

    def load(self, x_dict, y):
        for k, v in x_dict.items():
            self.x[k].copy_(v, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def replay(self):
        g1, g2, dense, sparse, buffers = self._graphs
        g1.replay()
        communicate(dense, sparse, self.world_size, self.group, buffers)
        g2.replay()
        return self.loss
