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
