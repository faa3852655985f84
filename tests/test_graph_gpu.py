"""hipGraph capture of the whole training step (what bench.py times) must replay to exactly the state the
eager launches reach: same kernels, same order, same buffers -> bitwise identical parameters."""
import numpy as np
import pytest
import torch

from _golden import Case, build_product_model, to_device

pytestmark = pytest.mark.gpu


def _run(case, n_steps, use_graph, limit=None):
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.trainers import CTRTrainer
    old = SwrModule.dense_table_limit_bytes
    if limit is not None:
        SwrModule.dense_table_limit_bytes = limit
    try:
        model = build_product_model(case)
        tr = CTRTrainer(model, "g", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
        model.train()
        x, y = case.batch(0)
        xd, yd = to_device(x), torch.from_numpy(y).cuda()
        done = 0
        if use_graph:
            from scenario_wise_rec.trainers.graph import GraphedStep
            g = GraphedStep(tr, xd, yd, warmup=2)
            for _ in range(n_steps - 2):
                g.replay()
        else:
            for _ in range(n_steps):
                tr.train_step(xd, yd)
        torch.cuda.synchronize()
        return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    finally:
        SwrModule.dense_table_limit_bytes = old


@pytest.mark.parametrize("side", ["forced", "auto"])
@pytest.mark.parametrize("name,limit", [("mmoe", None), ("mmoe", 2048), ("ple", None), ("ppnet", None), ("star", None)])
def test_graph_replay_matches_eager(name, limit, side, monkeypatch):
    """`side`: the suite forces the side-stream forks on (tests/conftest.py); "auto" is the product default -- the lookup that
    opens a step decides by batch size (these batches: ONE stream, the fused BatchNorm-backward + dX in front of the weight
    gradient, the optimizer's step bookkeeping riding the loss launch)."""
    from scenario_wise_rec import ops
    if side == "auto":
        monkeypatch.setattr(ops, "_SIDE_MODE", "auto")
    c = Case(name)
    a = _run(c, 6, use_graph=False, limit=limit)
    b = _run(c, 6, use_graph=True, limit=limit)
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{k}: max diff {np.abs(a[k].astype(np.float64) - b[k]).max()}"


@pytest.mark.parametrize("name", ["mmoe", "star"])
def test_replay_after_a_backward_pass_nobody_consumed(name):
    """With FusedAdam.clear_grads the captured step holds no zero_grad fill.  An eager forward + backward WITHOUT an
    optimizer step between two replays leaves gradients in the arena; the next step's zero_grad must wipe them
    (`ctr_trainer.py:71`) -- GraphedStep.replay does, so the replayed loop ends where the eager one does, bit for bit."""
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.trainers.graph import GraphedStep
    c = Case(name)
    res = {}
    for use_graph in (False, True):
        model = build_product_model(c)
        tr = CTRTrainer(model, "g", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
        assert tr.optimizer.clear_grads
        model.train()
        x, y = c.batch(0)
        xd, yd = to_device(x), torch.from_numpy(y).cuda()
        x1, y1 = c.batch(1)
        x1d, y1d = to_device(x1), torch.from_numpy(y1).cuda()
        if use_graph:
            g = GraphedStep(tr, xd, yd, warmup=2)
            g.replay()
            tr.forward_backward(x1d, y1d)           # gradients nobody consumes
            assert model.arena_dirty()
            g.replay()
            g.replay()
        else:
            for _ in range(3):
                tr.train_step(xd, yd)
            tr.forward_backward(x1d, y1d)
            tr.train_step(xd, yd)
            tr.train_step(xd, yd)
        torch.cuda.synchronize()
        res[use_graph] = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    for k in res[False]:
        assert np.array_equal(res[False][k], res[True][k]), k


@pytest.mark.parametrize("name", ["mmoe", "star"])
def test_side_streams_by_batch_size(name, monkeypatch):
    """ops.SIDE_STREAM in its automatic mode (the product's default; the test suite forces the forks on, conftest.py): the
    lookup that opens a step turns the side streams on for batches of SIDE_MIN_BATCH rows and more, off below -- and the
    single-stream step (captured and replayed) lands bitwise where the multi-stream step lands: the forks only move launches
    between streams, never change a sum."""
    from scenario_wise_rec import ops
    c = Case(name)
    monkeypatch.setattr(ops, "_SIDE_MODE", "1")
    monkeypatch.setattr(ops, "SIDE_STREAM", True)
    multi = _run(c, 5, use_graph=True)
    monkeypatch.setattr(ops, "_SIDE_MODE", "auto")
    monkeypatch.setattr(ops, "SIDE_MIN_BATCH", 1 << 30)
    single = _run(c, 5, use_graph=True)
    assert ops.SIDE_STREAM is False                       # decided by the lookup: the fixture's batch is below the threshold
    monkeypatch.setattr(ops, "SIDE_MIN_BATCH", 1)
    auto_on = _run(c, 5, use_graph=True)
    assert ops.SIDE_STREAM is True
    for k in multi:
        assert np.array_equal(multi[k], single[k]), f"{k}: one stream differs from the forked step"
        assert np.array_equal(multi[k], auto_on[k]), k


def test_fork_policy_looks_at_what_the_step_can_overlap(monkeypatch):
    """Automatic side-stream mode (ops.SIDE_MIN_BATCH_FUSED): a training step with a fused first layer AND a row-sparse table forks
    from 4 096 rows on (three independent chains behind dX); the same model below that, or without a large table, stays on one
    stream; an evaluation forward never changes the decision a training step's backward will read."""
    from scenario_wise_rec import ops
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.models.multi_domain import MMOE
    from scenario_wise_rec.trainers import CTRTrainer
    monkeypatch.setattr(ops, "_SIDE_MODE", "auto")
    monkeypatch.setattr(SwrModule, "dense_table_limit_bytes", 64 * 1024)
    rng = np.random.default_rng(1)

    def step(vocab_big, B, train=True):
        torch.manual_seed(0)
        feats = [SparseFeature("a", vocab_big, 16), SparseFeature("b", 40, 16), SparseFeature("c", 3, 16), DenseFeature("d0")]
        model = MMOE(feats, domain_num=3, n_expert=2, expert_params={"dims": [16]}, tower_params={"dims": [8]})
        tr = CTRTrainer(model, "policy", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
        tr.use_graph = False
        x = {"a": rng.integers(0, vocab_big, size=B), "b": rng.integers(0, 40, size=B), "c": rng.integers(0, 3, size=B),
             "d0": rng.random(B).astype(np.float32), "domain_indicator": rng.integers(0, 3, size=B)}
        xd = {k: torch.from_numpy(v).cuda() for k, v in x.items()}
        y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).cuda()
        if train:
            model.train()
            tr.train_step(xd, y)
        else:
            model.eval()
            with torch.no_grad():
                model(xd)
        torch.cuda.synchronize()
        return ops.SIDE_STREAM

    assert step(50000, 4096) is True            # fused first layer + a 3.2 MB (row-sparse) table
    assert step(50000, 2048) is False           # below SIDE_MIN_BATCH_FUSED
    assert step(500, 4096) is False             # no large table: nothing to overlap but the host cost
    assert step(50000, 4096) is True
    assert step(500, 4096, train=False) is True  # an evaluation forward leaves the training step's decision alone
