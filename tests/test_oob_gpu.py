"""No kernel may write outside the tensors it is given.

Every `torch.empty / zeros / empty_like / zeros_like` of the product's Python layer (ops, layers, optim, parallel, module,
graph) is replaced by an allocation with 64 KiB guard bands of a known byte pattern in front of and behind the payload;
after forward + backward + optimizer steps every guard band must be intact.  An out-of-range store (tile edges at
small batches, padded leading dimensions, workspace sizing) that normally lands in the slack of a caching-allocator
block -- and only rarely in another tensor or in a loaded code object -- fails here deterministically."""
import math

import numpy as np
import pytest
import torch

from _golden import Case, build_product_model, case_names, to_device

pytestmark = pytest.mark.gpu
GUARD = 64 * 1024
PATTERN = 0xA5


class GuardedTorch(object):
    """Stands in for the `torch` module inside the product's modules: allocation functions add guard bands."""

    def __init__(self, real):
        self._real = real
        self.guards = []            # (buffer uint8, payload bytes, description)

    def __getattr__(self, name):
        return getattr(self._real, name)

    def _alloc(self, shape, dtype, device, fill_zero):
        real = self._real
        dtype = dtype or real.get_default_dtype()
        dev = real.device(device) if device is not None else real.device("cpu")
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(int(s) for s in shape)
        n = math.prod(shape)
        if dev.type != "cuda":
            return (real.zeros if fill_zero else real.empty)(shape, dtype=dtype, device=dev)
        es = real.empty((), dtype=dtype).element_size()
        nbytes = (n * es + 255) // 256 * 256
        buf = real.empty(nbytes + 2 * GUARD, dtype=real.uint8, device=dev)
        buf.fill_(PATTERN)
        payload = buf[GUARD:GUARD + n * es]
        if fill_zero:
            payload.zero_()
        self.guards.append((buf, n * es, f"{shape} {dtype}"))
        return payload.view(dtype).view(shape)

    @staticmethod
    def _shape(size):
        return size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size

    def empty(self, *size, dtype=None, device=None, **kw):
        return self._alloc(self._shape(size), dtype, device, False)

    def zeros(self, *size, dtype=None, device=None, **kw):
        return self._alloc(self._shape(size), dtype, device, True)

    def empty_like(self, t, dtype=None, device=None, **kw):
        return self._alloc(tuple(t.shape), dtype or t.dtype, device or t.device, False)

    def zeros_like(self, t, dtype=None, device=None, **kw):
        return self._alloc(tuple(t.shape), dtype or t.dtype, device or t.device, True)

    def check(self):
        self._real.cuda.synchronize()
        bad = []
        for buf, nbytes, what in self.guards:
            head, tail = buf[:GUARD], buf[GUARD + nbytes:]
            for name, band in (("before", head), ("behind", tail)):
                if not bool((band == PATTERN).all()):
                    idx = int((band != PATTERN).nonzero()[0])
                    bad.append(f"write {name} a {what} tensor ({nbytes} bytes), first at guard offset {idx}")
        return bad


def _guard(monkeypatch):
    import scenario_wise_rec.basic.layers as layers
    import scenario_wise_rec.basic.module as module
    import scenario_wise_rec.ops as ops
    import scenario_wise_rec.optim as optim
    import scenario_wise_rec.parallel as parallel
    import scenario_wise_rec.trainers.graph as graph
    g = GuardedTorch(torch)
    for m in (ops, layers, module, optim, parallel, graph):
        monkeypatch.setattr(m, "torch", g)
    return g


SINGLE = [n for n in case_names() if "_dp" not in n]


@pytest.mark.parametrize("rows", [32, 33, 250], ids=["b32", "b33", "b250"])
@pytest.mark.parametrize("limit", [None, 2048], ids=["dense", "rows"])
@pytest.mark.parametrize("name", SINGLE)
def test_no_kernel_writes_outside_its_tensors(name, limit, rows, monkeypatch):
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.trainers import CTRTrainer
    if limit is not None:
        monkeypatch.setattr(SwrModule, "dense_table_limit_bytes", limit)
    g = _guard(monkeypatch)
    c = Case(name)
    x, y = c.batch(0)
    rows = min(rows, len(y))
    x = {k: v[:rows] for k, v in x.items()}
    y = y[:rows]
    model = build_product_model(c)
    tr = CTRTrainer(model, "guard", optimizer_params={"lr": c.meta["lr"], "weight_decay": c.meta["weight_decay"]}, device="cuda")
    model.train()
    xd, yd = to_device(x), torch.from_numpy(y).cuda()
    tr.train_step(xd, yd)
    # the data-parallel step's split backward (row lists first, weight gradients and small tables as late jobs)
    ops.split_backward(True)
    try:
        tr.forward_backward(xd, yd)
    finally:
        ops.split_backward(False)
    ops.run_late_jobs()
    tr.optimizer.step()
    model.eval()
    with torch.no_grad():
        model(xd)
    H.check_errors()
    bad = g.check()
    assert not bad, f"{name}: " + "; ".join(bad[:5])
    assert len(g.guards) > 20
