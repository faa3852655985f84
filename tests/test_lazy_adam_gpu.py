"""Exact lazy row updates (optim.LazyRows) against the per-step sweep over the whole table: the reference's dense
Adam moves EVERY row of every table every step (g = weight_decay * p on rows without data gradient).
Replaying those decay-only updates when a row is next read, or when the table is materialised, must give the same
bits as sweeping the table every step."""
import numpy as np
import pytest
import torch

from _golden import Case, build_product_model, to_device

pytestmark = pytest.mark.gpu


def _train(case, lazy, n_steps, limit=1024):
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.trainers import CTRTrainer
    old = SwrModule.dense_table_limit_bytes
    SwrModule.dense_table_limit_bytes = limit          # tables above 16 rows x 16 take the row-sparse path
    try:
        model = build_product_model(case)
        tr = CTRTrainer(model, "lazy", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5, "lazy_rows": lazy}, device="cuda")
        model.train()
        rng = np.random.default_rng(0)
        mid_eval = None
        for s in range(n_steps):
            x, y = case.batch(s % 3)
            # a different random subset of rows every step, so rows go untouched for several steps
            keep = rng.choice(len(y), size=len(y) // 4, replace=False)
            xs = {k: v[keep] for k, v in x.items()}
            tr.train_step(to_device(xs), torch.from_numpy(y[keep]).cuda())
            if s == n_steps // 2:
                model.eval()
                with torch.no_grad():
                    mid_eval = model(to_device(case.batch(0)[0])).cpu().numpy()       # lookups in eval mode catch up too
                model.train()
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in model.state_dict().items()}, mid_eval
    finally:
        SwrModule.dense_table_limit_bytes = old


@pytest.mark.parametrize("name", ["mmoe", "ppnet"])
def test_lazy_rows_equal_dense_sweep_bitwise(name):
    c = Case(name)
    lazy, ev_l = _train(c, True, 9)
    dense, ev_d = _train(c, False, 9)
    assert np.array_equal(ev_l, ev_d)
    for k in dense:
        assert np.array_equal(lazy[k], dense[k]), f"{k}: max diff {np.abs(lazy[k].astype(np.float64) - dense[k]).max()}"


@pytest.mark.parametrize("dim", [16, 12], ids=["dim16-elected-in-one-launch", "dim12-claim-then-replay"])
def test_shared_lazy_table_is_caught_up_once_per_row(dim):
    """Two large (row-sparse, lazily updated) tables, one of them looked up through TWO id columns (`shared_with`) that hold
    the same id at the same batch position: the batched catch-up (optim.catchup_many) must replay a row's pending
    decay-only steps once -- lazy == per-step sweep, bitwise, over steps that leave rows untouched for a while."""
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.models.multi_domain import MMOE
    from scenario_wise_rec.trainers import CTRTrainer

    def train(lazy):
        old = SwrModule.dense_table_limit_bytes
        SwrModule.dense_table_limit_bytes = 1024
        try:
            torch.manual_seed(11)
            # (dim 16: a row's lanes sit in one wavefront and elect its replayer by compare-and-swap inside ONE launch,
            # csrc/adam.hip adam_catchup_elect; dim 12: the claim launch + the replay launch)
            feats = [SparseFeature("a", 3000, dim), SparseFeature("b", 2000, dim), SparseFeature("a2", 3000, dim, shared_with="a"),
                     SparseFeature("small", 7, dim), DenseFeature("d0")]
            model = MMOE(feats, domain_num=3, n_expert=2, expert_params={"dims": [16]}, tower_params={"dims": [8]})
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if "embed_dict" in n:
                        p.normal_(0, 0.3)
            tr = CTRTrainer(model, "shared-lazy", optimizer_params={"lr": 1e-3, "weight_decay": 1e-2, "lazy_rows": lazy}, device="cuda")
            model.train()
            rng = np.random.default_rng(3)
            for step in range(7):
                B = 256
                a = rng.integers(0, 3000 if step % 3 else 300, size=B)       # every third step revisits a small set of rows
                a2 = a.copy()
                a2[B // 2:] = rng.integers(0, 3000, size=B - B // 2)          # first half: the same id at the same position
                x = {"a": a, "a2": a2, "b": rng.integers(0, 2000, size=B), "small": rng.integers(0, 7, size=B),
                     "d0": rng.random(B).astype(np.float32), "domain_indicator": rng.integers(0, 3, size=B)}
                y = (rng.random(B) < 0.3).astype(np.float32)
                tr.train_step({k: torch.from_numpy(v).cuda() for k, v in x.items()}, torch.from_numpy(y).cuda())
            torch.cuda.synchronize()
            return {k: v.cpu().numpy() for k, v in model.state_dict().items()}
        finally:
            SwrModule.dense_table_limit_bytes = old
    lazy, dense = train(True), train(False)
    for k in dense:
        assert np.array_equal(lazy[k], dense[k]), f"{k}: max diff {np.abs(lazy[k].astype(np.float64) - dense[k]).max()}"


@pytest.mark.parametrize("early", [False, True])
def test_early_rows_is_the_steps_own_catchup_and_bookkeeping(early):
    """FusedAdam.early_rows (the data-parallel one-graph step runs it on the exchange's branch, parallel.py): the catch-up of the
    merged row lists' rows that THIS rank did not look up + the step bookkeeping, done ahead of `step()`, must leave exactly the state
    `step()` alone leaves -- checked on a row list that holds rows lagging several steps (what other ranks' lookups bring in at
    world > 1; no multi-GPU box needed) against the dense per-step sweep, bitwise."""
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.models.multi_domain import MMOE
    from scenario_wise_rec.trainers import CTRTrainer

    def train(lazy):
        old = SwrModule.dense_table_limit_bytes
        SwrModule.dense_table_limit_bytes = 1024
        try:
            torch.manual_seed(5)
            feats = [SparseFeature("a", 4000, 16), SparseFeature("small", 7, 16), DenseFeature("d0")]
            model = MMOE(feats, domain_num=3, n_expert=2, expert_params={"dims": [16]}, tower_params={"dims": [8]})
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if "embed_dict" in n:
                        p.normal_(0, 0.3)
            tr = CTRTrainer(model, "early-rows", optimizer_params={"lr": 1e-3, "weight_decay": 1e-2, "lazy_rows": lazy}, device="cuda")
            tr.use_graph = False
            model.train()
            table = dict(model.named_parameters())["embedding.embed_dict.a.weight"]
            rng = np.random.default_rng(8)
            for step in range(6):
                B = 256
                x = {"a": rng.integers(0, 400, size=B), "small": rng.integers(0, 7, size=B), "d0": rng.random(B).astype(np.float32),
                     "domain_indicator": rng.integers(0, 3, size=B)}
                y = (rng.random(B) < 0.3).astype(np.float32)
                loss = tr.forward_backward({k: torch.from_numpy(v).cuda() for k, v in x.items()}, torch.from_numpy(y).cuda())
                del loss
                if step >= 2:
                    # what the exchange hands the optimizer at world > 1: the merged list, with rows nobody on this rank looked up
                    # (ids >= 400: they have been lagging since step 0) -- appended behind this rank's own, ascending like a merge
                    urow, ugrad = table._swr_sparse_grad
                    extra = torch.from_numpy(np.sort(rng.choice(np.arange(400, 4000), size=64, replace=False)).astype(np.int32)).cuda()
                    g_extra = torch.from_numpy((rng.standard_normal((64, 16)) * 1e-3).astype(np.float32)).cuda()
                    table._swr_sparse_grad = (torch.cat([urow, extra]), torch.cat([ugrad, g_extra]))
                    table._swr_sparse_local = False
                    if early and lazy:
                        assert tr.optimizer.early_rows([table])
                tr.optimizer.step()
            torch.cuda.synchronize()
            if hasattr(model, "materialize"):
                model.materialize()
            return {k: v.cpu().numpy() for k, v in model.state_dict().items()}
        finally:
            SwrModule.dense_table_limit_bytes = old
    lazy, dense = train(True), train(False)
    for k in dense:
        assert np.array_equal(lazy[k], dense[k]), f"{k}: max diff {np.abs(lazy[k].astype(np.float64) - dense[k]).max()}"
