"""The drop-in loop (CTRTrainer.fit / train_one_epoch, reference trainers/ctr_trainer.py:62-97) on the HIP path:
hipGraph capture inside the loop, the optimizer's history ring, checkpoints, optimizers other than Adam."""
import numpy as np
import pytest
import torch

from _golden import Case, build_product_model, make_oracle, to_device

pytestmark = pytest.mark.gpu
LR, WD = 1e-3, 1e-5


def _batches(c, batch, tail, device=None):
    """Rows of the three golden batches, cut into `batch`-row batches + one ragged `tail`-row batch per epoch."""
    xs = [c.batch(s) for s in range(3)]
    x = {k: np.concatenate([b[0][k] for b in xs]) for k in xs[0][0]}
    y = np.concatenate([b[1] for b in xs])
    n_full = (len(y) - tail) // batch
    out, lo = [], 0
    for n in [batch] * n_full + ([tail] if tail else []):
        xb = {k: torch.from_numpy(np.ascontiguousarray(v[lo:lo + n])) for k, v in x.items()}
        yb = torch.from_numpy(y[lo:lo + n])
        if device is not None:
            xb, yb = {k: v.to(device) for k, v in xb.items()}, yb.to(device)
        out.append((xb, yb))
        lo += n
    return out


def _state(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def _limit(monkeypatch, nbytes):
    from scenario_wise_rec.basic.module import SwrModule
    if nbytes is not None:
        monkeypatch.setattr(SwrModule, "dense_table_limit_bytes", nbytes)


@pytest.mark.parametrize("name,limit,on_device", [("mmoe", None, False), ("mmoe", 2048, True), ("star", None, True),
                                                  ("hamur_small", 2048, False)])
def test_fit_with_graph_capture_is_bitwise_the_eager_loop(name, limit, on_device, monkeypatch, tmp_path):
    """2 epochs of 6 full batches + a ragged tail: eager, eager, CAPTURE + replay, replay ..., tail eager, next epoch
    replays again.  Same parameters, buffers and optimizer trajectory as launching every step eagerly."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    _limit(monkeypatch, limit)
    c = Case(name)
    res = {}
    for use_graph in (False, True):
        model = build_product_model(c)
        tr = CTRTrainer(model, "loop", optimizer_params={"lr": LR, "weight_decay": WD}, n_epoch=2, device="cuda",
                        model_path=str(tmp_path))
        tr.use_graph = use_graph
        tr.fit(_batches(c, 100, 50, device="cuda" if on_device else None))
        torch.cuda.synchronize()
        H.check_errors()
        assert (tr._graph is not None) == use_graph
        res[use_graph] = _state(model)
    for k, v in res[False].items():
        assert np.array_equal(res[True][k], v), f"{k}: max diff {np.abs(res[True][k].astype(np.float64) - v).max()}"


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "replayed"])
def test_history_ring_wraps_without_changing_a_bit(graph, monkeypatch):
    """FusedAdam(hist_cap=8): the lazy tables are flushed before the 8-step ring of per-step scalars wraps; 40 steps
    on rotating sub-batches (rows idle for many steps) == the same run with the default 2^20-step history."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.trainers.graph import GraphedStep
    _limit(monkeypatch, 1024)
    c = Case("mmoe")
    res = {}
    for cap in (1 << 20, 8):
        model = build_product_model(c)
        tr = CTRTrainer(model, "ring", optimizer_params={"lr": LR, "weight_decay": WD, "hist_cap": cap}, device="cuda")
        model.train()
        rng = np.random.default_rng(1)
        g = None
        for s in range(40):
            x, y = c.batch(s % 3)
            keep = rng.choice(len(y), size=60, replace=False)
            xd, yd = to_device({k: v[keep] for k, v in x.items()}), torch.from_numpy(y[keep]).cuda()
            if graph and s >= 2:
                if g is None:
                    g = GraphedStep(tr, xd, yd, warmup=0)
                g.load(xd, yd)
                g.replay()
            else:
                tr.train_step(xd, yd)
        torch.cuda.synchronize()
        H.check_errors()
        assert tr.optimizer._hyper[0][3] == 40                 # host step count follows the replays
        res[cap] = _state(model)
    for k, v in res[1 << 20].items():
        assert np.array_equal(res[8][k], v), k


def test_checkpoint_of_model_and_optimizer_resumes_bitwise(monkeypatch):
    """state_dict() of the model (lazy rows materialised) and of FusedAdam (torch.optim.Adam's layout) -> fresh objects ->
    the continued run equals the uninterrupted one; loading weights under a live optimizer with rows behind is exact."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    _limit(monkeypatch, 1024)
    c = Case("mmoe")
    rng = np.random.default_rng(2)
    subs = []
    for s in range(6):
        x, y = c.batch(s % 3)
        keep = rng.choice(len(y), size=80, replace=False)
        subs.append(({k: v[keep] for k, v in x.items()}, y[keep]))

    def make():
        m = build_product_model(c)
        t = CTRTrainer(m, "ckpt", optimizer_params={"lr": LR, "weight_decay": WD}, device="cuda")
        m.train()
        return m, t

    def run(tr, rng_):
        for x, y in rng_:
            tr.train_step(to_device(x), torch.from_numpy(y).cuda())

    m_a, t_a = make()
    run(t_a, subs)
    m_b, t_b = make()
    run(t_b, subs[:3])
    msd = {k: v.clone() for k, v in m_b.state_dict().items()}
    osd = t_b.optimizer.state_dict()
    # the layout is torch.optim.Adam's: a stock Adam over same-shaped parameters accepts it
    ref_params = [torch.nn.Parameter(torch.zeros_like(p, device="cpu")) for p in m_b.parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=LR, weight_decay=WD)
    ref_opt.load_state_dict({"state": {i: {k: v.cpu() for k, v in st.items()} for i, st in osd["state"].items()},
                             "param_groups": osd["param_groups"]})
    assert all(int(st["step"]) == 3 for st in osd["state"].values()) and len(osd["state"]) > 0
    m_c, t_c = make()
    m_c.load_state_dict(msd)
    t_c.optimizer.load_state_dict(osd)
    run(t_c, subs[3:])
    torch.cuda.synchronize()
    H.check_errors()
    a, cc = _state(m_a), _state(m_c)
    for k, v in a.items():
        assert np.array_equal(cc[k], v), k
    # load_state_dict under a live optimizer whose rows are behind: pending decay is applied to the OLD values first
    m_d, t_d = make()
    run(t_d, subs[:3])
    m_d.load_state_dict(msd)                          # same weights as it holds (after materialisation): a no-op overall
    run(t_d, subs[3:])
    torch.cuda.synchronize()
    d = _state(m_d)
    for k, v in a.items():
        assert np.array_equal(d[k], v), k


def test_other_optimizers_get_dense_table_gradients(monkeypatch):
    """optimizer_fn=SGD with a table above the row-sparse limit: the trainer makes every table take a dense `.grad`
    (row lists are a FusedAdam-only representation), so the large table IS updated: w - lr * grad (oracle)."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    _limit(monkeypatch, 1024)
    c = Case("mmoe")
    model = build_product_model(c)
    tr = CTRTrainer(model, "sgd", optimizer_fn=torch.optim.SGD, optimizer_params={"lr": 0.05}, device="cuda")
    model.train()
    x, y = c.batch(0)
    w0 = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
    tr.train_step(to_device(x), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    H.check_errors()
    _, _, og = make_oracle(c).loss_and_grads(x, y)
    for k, p in model.named_parameters():
        assert getattr(p, "_swr_sparse_grad", None) is None, k
        want = w0[k] - 0.05 * og[k]
        np.testing.assert_allclose(p.detach().cpu().numpy(), want, rtol=0, atol=0.05 * (2e-4 * np.abs(og[k]).max() + 3e-7) + 1e-7,
                                   err_msg=k)


def test_second_lookup_of_a_row_sparse_table_fails_loudly(monkeypatch):
    """Row lists do not accumulate: backward() twice without zero_grad() (or two gathers of one large table) raises
    instead of silently dropping the first gradient."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    _limit(monkeypatch, 1024)
    c = Case("mmoe")
    model = build_product_model(c)
    tr = CTRTrainer(model, "twice", optimizer_params={"lr": LR, "weight_decay": WD}, device="cuda")
    model.train()
    x, y = c.batch(0)
    xd, yd = to_device(x), torch.from_numpy(y).cuda()
    tr.criterion(model(xd), yd).backward()
    with pytest.raises(H.SwrError, match="already pending"):
        tr.criterion(model(xd), yd).backward()
    torch.cuda.synchronize()


def test_adam_skips_an_untouched_parameter_between_two_touched_ones():
    """Three parameters back to back in one storage, the 2-element middle one without gradient: torch leaves it alone
    (no decay, no state); the fused run must not sweep over it."""
    from scenario_wise_rec.optim import FusedAdam
    flat = torch.randn(8 + 2 + 8, device="cuda")
    gflat = torch.randn_like(flat)
    views = [flat[0:8], flat[8:10], flat[10:18]]
    params = [torch.nn.Parameter(v) for v in views]
    for p, v in zip(params, views):
        p.data = v
    params[0].grad, params[2].grad = gflat[0:8], gflat[10:18]
    before = flat.clone()
    opt = FusedAdam(params, lr=0.1, weight_decay=0.5)
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(flat[8:10], before[8:10])                          # the untouched parameter: not a bit moved
    assert not torch.equal(flat[0:8], before[0:8]) and not torch.equal(flat[10:18], before[10:18])
    ref = [torch.nn.Parameter(before[0:8].clone()), torch.nn.Parameter(before[10:18].clone())]
    ref[0].grad, ref[1].grad = gflat[0:8].clone(), gflat[10:18].clone()
    torch.optim.Adam(ref, lr=0.1, weight_decay=0.5).step()
    torch.testing.assert_close(flat[0:8], ref[0].data, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(flat[10:18], ref[1].data, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name,limit", [("mmoe", None), ("mmoe", 2048), ("ppnet", None)])
def test_optimizer_cleared_gradients_replace_the_zero_grad_fill(name, limit, monkeypatch):
    """FusedAdam.clear_grads (CTRTrainer's default): the dense update zeroes each gradient it consumed and
    SwrModule.zero_grad launches no fill -- same trajectory, bit for bit, as the fill; a backward pass that no optimizer
    step consumed is still wiped by the next zero_grad (`ctr_trainer.py:71-73`)."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    _limit(monkeypatch, limit)
    c = Case(name)
    res = {}
    for clear in (False, True):
        model = build_product_model(c)
        tr = CTRTrainer(model, "clear", optimizer_params={"lr": LR, "weight_decay": WD}, device="cuda")
        tr.optimizer.clear_grads = clear
        model.train()
        fills = []
        real = H.lib.swr_zero
        monkeypatch.setattr(H.lib, "swr_zero", lambda *a: (fills.append(1), real(*a))[1])
        for s in range(3):
            x, y = c.batch(s)
            tr.train_step(to_device(x), torch.from_numpy(y).cuda())
        n_fills = len(fills)
        arena = model.arena()["g"]
        if clear:
            assert not torch.count_nonzero(arena)                      # everything written was consumed and zeroed
            assert n_fills == 0, "zero_grad still filled the arena"
        else:
            assert torch.count_nonzero(arena) and n_fills == 2      # (the first zero_grad finds the fresh arena untouched)
        # a backward pass without an optimizer step leaves gradients behind: the next zero_grad must fill
        x, y = c.batch(0)
        tr.forward_backward(to_device(x), torch.from_numpy(y).cuda())
        assert torch.count_nonzero(arena)
        model.zero_grad()
        assert not torch.count_nonzero(arena) and len(fills) > n_fills
        monkeypatch.setattr(H.lib, "swr_zero", real)
        torch.cuda.synchronize()
        H.check_errors()
        res[clear] = _state(model)
    for k, v in res[False].items():
        assert np.array_equal(v, res[True][k]), k


@pytest.mark.parametrize("name,limit", [("mmoe", None), ("mmoe", 2048), ("star", 2048), ("epnet", None), ("hamur_small", None)])
def test_step_bookkeeping_rides_the_loss_launch_without_changing_a_bit(name, limit, monkeypatch):
    """The optimizer's step counter / bias corrections advance as a RIDER of the fused select + BCE launch (swr.h
    swr_select_bce_fwd_adv, ops.offer_loss_rider) instead of a 1-thread launch of their own: the same steps with the rider off
    (SWR_LOSS_RIDER=0: ops.LOSS_RIDER) give the same bits -- state, lazily updated rows, loss sequence -- and where a rider is
    possible (a model whose forward ends in the fused select + BCE; from the second step on) no swr_adam_advance launch is left."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    from scenario_wise_rec.trainers import CTRTrainer
    _limit(monkeypatch, limit)
    c = Case(name)
    res, launches = [], []
    real = ops.lib.swr_adam_advance
    monkeypatch.setattr(ops, "SIDE_STREAM", False)        # the rider is for single-stream steps (short batches)
    monkeypatch.setattr(ops, "_SIDE_MODE", "0")
    for rider in (False, True):
        monkeypatch.setattr(ops, "LOSS_RIDER", rider)
        count = []
        import scenario_wise_rec.optim as optim
        monkeypatch.setattr(optim.lib, "swr_adam_advance", lambda *a, _c=count: (_c.append(1), real(*a))[1])
        model = build_product_model(c)
        tr = CTRTrainer(model, "rider", optimizer_params={"lr": LR, "weight_decay": WD}, device="cuda")
        tr.use_graph = False
        model.train()
        losses = []
        for s in (0, 1, 2, 0, 1):
            x, y = c.batch(s)
            losses.append(float(tr.train_step(to_device(x), torch.from_numpy(y).cuda()).detach()))
        torch.cuda.synchronize()
        H.check_errors()
        if hasattr(model, "materialize"):
            model.materialize()
        res.append((_state(model), losses, int(tr.optimizer._hyper[0][3])))
        launches.append(len(count))
    (s0, l0, n0), (s1, l1, n1) = res
    assert l0 == l1 and n0 == n1
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    if name in ("mmoe", "hamur_small"):      # (EPNet has no domain select, STAR adds its auxiliary logit inside it: plain BCE there,
        assert launches[1] <= 1 < launches[0], launches      # the advance stays a launch.)  Step 1 creates the device scalars; steps 2..5 ride
