"""Model-level parity on the GPU: the HIP path (product models, through the C ABI) against the golden
vectors the reference produced (tests/golden/*.npz) -- forward in eval and train mode, every parameter
gradient, the state after 1 and 3 optimisation steps driven by CTRTrainer, and the loss sequence.

Tolerance: the north star's 1e-4 on fp32 logits; rows zeroed by the domain select must be exactly 0.0
(bit-exact routing).  Gradients: 2e-4 of the largest entry of each tensor (+3e-7 noise floor, see
tests/_golden.state_atol)."""
import numpy as np
import pytest
import torch

from _golden import Case, assert_probs_close, build_product_model, case_names, assert_state_close, to_device

pytestmark = pytest.mark.gpu
SINGLE = [n for n in case_names() if "_dp" not in n]


@pytest.mark.parametrize("name", SINGLE)
def test_eval_forward(name):
    from scenario_wise_rec import _hip as H
    c = Case(name)
    model = build_product_model(c).eval()
    x, _ = c.batch(0)
    with torch.no_grad():
        p = model(to_device(x)).cpu().numpy()
    H.check_errors()
    assert_probs_close(p, c.z["eval_probs"], tol=1e-4)


@pytest.mark.parametrize("name", SINGLE)
def test_train_three_steps(name):
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    c = Case(name)
    model = build_product_model(c)
    trainer = CTRTrainer(model, "golden", optimizer_params={"lr": c.meta["lr"], "weight_decay": c.meta["weight_decay"]},
                         device="cuda")
    model.train()
    losses = []
    for s in range(3):
        x, y = c.batch(s)
        xd, yd = to_device(x), torch.from_numpy(y).cuda()
        if s == 0:
            # the step, opened up to look at the gradients (ctr_trainer.py:69-73)
            p = model(xd)
            loss = trainer.criterion(p, yd)
            model.zero_grad()
            loss.backward()
            assert_probs_close(p.detach().cpu().numpy(), c.z["train_probs"], tol=1e-4)
            want = c.group("grad")
            named = dict(model.named_parameters())
            for k, g in want.items():
                prm = named[k]
                got = prm.grad
                sg = getattr(prm, "_swr_sparse_grad", None)
                assert got is not None or sg is not None, f"{k}: no gradient"
                scale = max(1e-6, float(np.abs(g).max()))
                np.testing.assert_allclose(got.cpu().numpy(), g, rtol=0, atol=2e-4 * scale + 3e-7, err_msg=k)
            for k, prm in named.items():
                if k not in want:          # reference grad None (PPNet agn tables): must stay untouched
                    assert not getattr(prm, "_swr_touched", False), k
            trainer.optimizer.step()
        else:
            loss = trainer.train_step(xd, yd)
        losses.append(float(loss))
        if s in (0, 2):
            want = c.group("state1" if s == 0 else "state3")
            got = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
            assert set(got) == set(want)
            for k, v in want.items():
                if k.endswith("num_batches_tracked"):
                    assert int(got[k]) == int(v), k
                else:
                    assert_state_close(got[k], v, c, k, s + 1)
    H.check_errors()
    np.testing.assert_allclose(losses, c.z["losses"], rtol=5e-5)
    model.eval()
    with torch.no_grad():
        p = model(to_device(c.batch(0)[0])).cpu().numpy()
    assert_probs_close(p, c.z["eval3_probs"], tol=2e-4)


# ---- small local batches (the per-rank shard of an 8-way data-parallel step: 32 rows, less than one wavefront) -------
def _grads_of(model):
    out = {}
    for k, p in model.named_parameters():
        sg = getattr(p, "_swr_sparse_grad", None)
        if sg is not None:
            r, g = sg[0].cpu().numpy(), sg[1].cpu().numpy()
            full = np.zeros(tuple(p.shape), np.float64)
            np.add.at(full, r[r >= 0], g[r >= 0].astype(np.float64))
            out[k] = full
        elif p.grad is not None:
            out[k] = p.grad.cpu().numpy()
    return out


@pytest.mark.parametrize("split", [False, True], ids=["whole", "split"])
@pytest.mark.parametrize("limit", [None, 2048], ids=["dense", "rows"])
@pytest.mark.parametrize("rank", range(8))
def test_shard_of_32_rows_vs_oracle(rank, limit, split, monkeypatch):
    """One rank's share of tests/golden/mmoe_dp8 (32 rows) through the HIP path, forward and every gradient against
    the fp64 oracle on the same rows; `split` = the data-parallel step's split backward (row lists first, weight
    gradients and small tables as late jobs), `rows` = tables above 2 KiB take row-sparse gradients."""
    from _golden import make_oracle
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.trainers import CTRTrainer
    if limit is not None:
        monkeypatch.setattr(SwrModule, "dense_table_limit_bytes", limit)
    c = Case("mmoe_dp8")
    x, y = c.batch(0)
    sh = len(y) // 8
    x = {k: v[rank * sh:(rank + 1) * sh] for k, v in x.items()}
    y = y[rank * sh:(rank + 1) * sh]
    model = build_product_model(c)
    trainer = CTRTrainer(model, "shard", optimizer_params={"lr": c.meta["lr"], "weight_decay": c.meta["weight_decay"]},
                         device="cuda")
    model.train()
    xd, yd = to_device(x), torch.from_numpy(y).cuda()
    if split:
        ops.split_backward(True)
        try:
            loss = trainer.forward_backward(xd, yd)
        finally:
            ops.split_backward(False)
        ops.run_late_jobs()
    else:
        loss = trainer.forward_backward(xd, yd)
    torch.cuda.synchronize()
    H.check_errors()
    op, oloss, ograds = make_oracle(c).loss_and_grads(x, y)
    assert abs(float(loss) - oloss) < 2e-6 * max(1.0, abs(oloss))
    got = _grads_of(model)
    assert set(got) >= set(ograds)
    for k, g in ograds.items():
        scale = max(1e-6, float(np.abs(g).max()))
        np.testing.assert_allclose(got[k], g, rtol=0, atol=2e-4 * scale + 3e-7, err_msg=k)


# ---- results must not depend on what freed device memory holds (reads of uninitialised workspace / padding) -----------
def _poison_allocator(value_bits):
    """Fill the caching allocator's pools with a bit pattern and free them again: later torch.empty() calls get
    blocks holding that pattern."""
    keep = []
    for nbytes in [512, 2048, 8192, 32768, 131072, 524288] * 24 + [8 << 20] * 8 + [64 << 20] * 2:
        t = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda")
        t.fill_(value_bits)
        keep.append(t)
    torch.cuda.synchronize()
    del keep
    # the poison must be what new tensors see (a few blocks may come from other cached memory: most must show it)
    want = value_bits if value_bits < 2 ** 31 else value_bits - 2 ** 32
    probes = [torch.empty(n // 4, dtype=torch.int32, device="cuda") for n in (512, 2048, 8192, 32768, 131072, 524288) * 2]
    assert sum(int(t[0]) == want for t in probes) >= len(probes) // 2
    del probes


def _one_step_bits(c, rows, poison):
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    x, y = c.batch(0)
    x = {k: v[:rows] for k, v in x.items()}
    y = y[:rows]
    torch.cuda.empty_cache()
    _poison_allocator(poison)
    model = build_product_model(c)
    trainer = CTRTrainer(model, "poison", optimizer_params={"lr": c.meta["lr"], "weight_decay": c.meta["weight_decay"]},
                         device="cuda")
    model.train()
    xd, yd = to_device(x), torch.from_numpy(y).cuda()
    _poison_allocator(poison)
    out = {}
    for s in range(2):
        loss = trainer.forward_backward(xd, yd)
        out[f"loss{s}"] = loss.detach().cpu().numpy()
        for k, g in _grads_of(model).items():
            out[f"g{s}/{k}"] = np.asarray(g)
        trainer.optimizer.step()
    torch.cuda.synchronize()
    H.check_errors()
    for k, v in model.state_dict().items():
        out["s/" + k] = v.cpu().numpy()
    return out


@pytest.mark.parametrize("rows", [32, 250], ids=["b32", "b250"])
@pytest.mark.parametrize("name", SINGLE)
def test_results_do_not_depend_on_stale_memory(name, rows):
    """Two training steps with the allocator's free blocks pre-filled with zeros, NaNs and 1e30: bitwise equal results.
    A kernel that reads workspace it did not write, padding columns, or rows past the batch (tile edges at small
    batches) fails here deterministically instead of flaking."""
    c = Case(name)
    rows = min(rows, len(c.batch(0)[1]))
    ref = _one_step_bits(c, rows, 0)
    for bits in (0x7FC00000, 0x7149F2CA):          # NaN, 1e30
        got = _one_step_bits(c, rows, bits)
        for k, v in ref.items():
            assert np.array_equal(got[k], v, equal_nan=True), f"{name}: {k} depends on stale memory (pattern {bits:#x})"


@pytest.mark.parametrize("name", ["sharedbottom", "ple", "ple_2level", "star", "ppnet"])
def test_routed_eval_equals_the_all_domains_eval(name, monkeypatch):
    """Inference routes every row through its own domain's branch only; the reference evaluates all
    branches on the whole batch and selects.  Same probabilities (BatchNorm is a fixed affine in eval mode), including
    ids outside [0, D) -> exactly 0.0 (sigmoid(aux) for STAR) and an empty domain."""
    from scenario_wise_rec import ops
    c = Case(name)
    model = build_product_model(c).eval()
    x, _ = c.batch(0)
    x = {k: v.copy() for k, v in x.items()}
    D = c.meta["domain_num"]
    dom = x["domain_indicator"]
    dom[dom == D - 1] = 0                    # an empty domain
    dom[::9] = D                             # out-of-range ids
    dom[4::13] = -1
    xd = to_device(x)
    with torch.no_grad():
        monkeypatch.setattr(ops, "ROUTED_EVAL", True)
        routed = model(xd).cpu().numpy()
        monkeypatch.setattr(ops, "ROUTED_EVAL", False)
        dense = model(xd).cpu().numpy()
    bad = (dom < 0) | (dom >= D)
    if name != "star":
        assert np.array_equal(routed[bad], np.zeros(bad.sum(), np.float32))
    np.testing.assert_allclose(routed, dense, rtol=0, atol=2e-7)
