"""Parity at the FULL BASELINE.json sizes: what tests/test_baseline_shapes_gpu.py checks at
reduced vocabularies and batches, here at the sizes bench.py runs.

* `test_cfg2_full`: config 2 exactly as benched -- batch 65 536, the real 4 371 900-row video table (row-sparse gradients,
  exact lazy Adam), FOUR rotating batches through the captured step (trainers/graph.py GraphedStep: two eager warm-up
  steps, capture, three replays with lazily updated rows lagging behind) -- against the torch-CPU restatement of the
  reference step (oracle/torch_port.py, pinned on the golden vectors by tests/test_oracle_golden.py): loss sequence,
  train-mode logits of a fresh batch within 1e-4, every dense parameter, the rows of the large table that were looked
  up, and a sample of rows that never were (they must have decayed exactly like the reference's dense Adam decays them).
* `test_cfg3_full` / `test_cfg4_full`: STAR at its per-GPU shard (batch 16 384, all 23 Ali-CCP tables at their full sizes:
  238 635 / 467 298 / 263 942 ... rows) and PLE (batch 8 192, the 748 000-row x 32 user table): one step against the fp64
  oracle holding the FULL tables (probabilities, loss, every gradient incl. the row lists of the large tables, the state
  after Adam for every row -- rows nobody looked up carry the oracle's dense decay), then captured + lazy rows == eager +
  dense sweep bitwise over rotating batches.
* `test_cfg5_full_tables` / `test_cfg6_full_tables`: the two 50 M-row x 64 tables of BASELINE config 5 (12.8 GB each) at
  full size with hashed 40-bit ids.  The fp64 oracle cannot hold such a table, so it runs on the COMPACTED row set: the
  rows any step looks up, gathered from the device table before training, with ids remapped -- exact for the looked-up
  rows (gradients and Adam state of a row depend on that row only) -- and rows nobody looked up are checked against the
  closed form of three decay-only Adam steps.
"""
import copy

import numpy as np
import pytest
import torch

import bench
from _golden import KinkTolerantGradCheck, assert_probs_close, dekink_mmoe_state, logit, perturb_product
from oracle.nn import Dense, Sparse
from oracle.optim import Adam
from test_baseline_shapes_gpu import mix64, oracle_for, run_config

pytestmark = pytest.mark.gpu
LR, WD = 1e-3, 1e-5
# tensors whose gradient holds a ReLU unit within fp32 rounding of its kink (KinkTolerantGradCheck's second form), MEASURED
# per configuration with these seeds (the product path is bitwise deterministic: the counts do not move from run to run);
# `pytest -s` prints the list.  None = the old "up to half of the tensors" cap.
CFG3_KINKED, CFG4_KINKED, CFG5_SHARD_KINKED, CFG6_SHARD_KINKED = 36, 0, 10, 16


def _close_but_for_adam_noise(got, want, n_steps, what, frac=2e-4):
    """Parameters after several Adam steps: equal to rounding, except where a gradient entry is numerical noise around
    zero (Adam turns its SIGN into a full +-lr step, in the reference as well): those few may differ by the steps taken."""
    err = np.abs(got.astype(np.float64) - want)
    tight = 3e-5 + 2e-4 * np.abs(want)
    bad = err > tight
    assert err.max() <= 2.2 * LR * n_steps + 1e-4 * np.abs(want).max(), f"{what}: max error {err.max():.3e}"
    assert bad.mean() <= frac, f"{what}: {int(bad.sum())} of {bad.size} entries differ by more than rounding (max {err.max():.3e})"


def _cfg2_model(state0=None):
    cfg = bench.CONFIGS[2]
    model, _feats = bench.build_model(cfg, seed=7)
    perturb_product(model, 31)
    if state0 is not None:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state0.items()})
    return model


def test_cfg2_full():
    """Three parts, all at batch 65 536 with the 4 371 900-row table:
    A. ONE step from a common state against the torch-CPU port: logits, loss, every gradient, the state after Adam, the
       train-mode logits of a fresh batch.
    B. the benched regime -- two eager warm-up steps, capture, three replays over rotating batches, exact LAZY row updates
       lagging behind -- must give the same BITS as five eager steps with a dense Adam sweep over the whole table every
       step (captured == eager and lazy == sweep, at full size).
    C. the same five steps against the port.  Adam divides by sqrt(v): at this batch size a gradient entry is a mean over
       65 536 samples (1e-7 .. 1e-5) and the first steps move every entry by ~lr regardless of it, so entries whose
       momentum passes through zero take steps whose SIGN is rounding noise -- in the reference as much as here (two runs
       of the reference on different thread counts differ the same way).  The trajectory is therefore pinned by its loss
       sequence (5e-5) and an envelope: no entry further than the steps taken, at most a few per cent beyond rounding."""
    from oracle.torch_port import MMoEPort
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.trainers.graph import GraphedStep
    cfg = bench.CONFIGS[2]
    B = cfg["batch"]
    model = _cfg2_model()
    state0 = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    batches = [bench.synth_batch(cfg, B, seed=4000 + j) for j in range(5)]       # 0-3 train (0 twice: warm-up), 4 probes
    feats = [Dense(f"d{i}") for i in range(cfg["n_dense"])] + [Sparse(f"s{i}", v, cfg["embed_dim"]) for i, v in enumerate(cfg["vocabs"])]
    # part A's state: no ReLU unit of the step on batch 0 within 2e-5 of its kink (13.6 M pre-activations; BatchNorm betas moved
    # by < 2e-3) -- every gradient entry is then pinned, no "kink" allowance (_golden.dekink_mmoe_state)
    state0 = dekink_mmoe_state(state0, feats, cfg["hyper"], batches[0][0])
    model = _cfg2_model(state0)
    big = "embedding.embed_dict.s1.weight"

    def to_dev(j):
        return {k: torch.from_numpy(v).cuda() for k, v in batches[j][0].items()}, torch.from_numpy(batches[j][1]).cuda()
    dev = [to_dev(j) for j in range(5)]

    # ---- A: one step
    trainer = CTRTrainer(model, "cfg2-full", optimizer_params={"lr": LR, "weight_decay": WD}, device="cuda")
    trainer.use_graph = False
    model.train()
    p = model(dev[0][0])
    loss = trainer.criterion(p, dev[0][1])
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    H.check_errors()
    port = MMoEPort(feats, cfg["hyper"], state0, threads=16)
    pp, pl, pg = port.loss_and_grads(*batches[0])
    assert_probs_close(p.detach().cpu().numpy(), pp, tol=1e-4)
    assert abs(float(loss.detach()) - pl) < 2e-6 * max(1.0, abs(pl))
    named = dict(model.named_parameters())
    kinks = KinkTolerantGradCheck()
    for k, g in pg.items():
        if np.abs(g).max() < 1e-7:
            continue                             # a bias in front of a BatchNorm: its true gradient is zero, what is computed is noise
        prm = named[k]
        sg = getattr(prm, "_swr_sparse_grad", None)
        if sg is not None:
            r, gg = sg[0].cpu().numpy(), sg[1].cpu().numpy().astype(np.float64)
            got = np.zeros(tuple(prm.shape))
            np.add.at(got, r[r >= 0], gg[r >= 0])
        else:
            got = prm.grad.cpu().numpy()
        kinks.check(got, g, 3e-4 * float(np.abs(g).max()) + 3e-9, k)
    kinks.finish(max_kinked=0)
    trainer.optimizer.step()
    port.step(*batches[0], lr=LR, weight_decay=WD)
    torch.cuda.synchronize()
    H.check_errors()
    got1 = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    for k, t in port.p.items():
        w = t.detach().numpy()
        if k == big:
            rows = np.unique(batches[0][0]["s1"])
            _close_but_for_adam_noise(got1[k][rows], w[rows], 1, "looked-up rows after one step", frac=1e-4)
            idle = np.setdiff1d(np.arange(0, w.shape[0], 37), rows)
            np.testing.assert_allclose(got1[k][idle], w[idle], rtol=3e-6, atol=1e-9)
        else:
            _close_but_for_adam_noise(got1[k], w, 1, k, frac=2e-3 if w.size < 2000 else 1e-4)
    with torch.no_grad():
        assert_probs_close(model(dev[4][0]).cpu().numpy(), port.forward(batches[4][0]).numpy(), tol=1e-4)
    del model, trainer

    # ---- B: captured + lazy == eager + dense sweep, bitwise
    def run(lazy, graphed):
        m = _cfg2_model(state0)
        tr = CTRTrainer(m, "cfg2-full", optimizer_params={"lr": LR, "weight_decay": WD, "lazy_rows": lazy}, device="cuda")
        tr.use_graph = False
        m.train()
        losses = []
        if graphed:
            g = GraphedStep(tr, dev[0][0], dev[0][1], warmup=2)                 # steps 1, 2 on batch 0 (eager), then capture
            for j in (1, 2, 3):
                g.load(*dev[j])
                losses.append(float(g.replay().clone()))
        else:
            for j in (0, 0, 1, 2, 3):
                losses.append(float(tr.train_step(*dev[j]).detach()))
            losses = losses[2:]
        torch.cuda.synchronize()
        H.check_errors()
        return {k: v.cpu().numpy() for k, v in m.state_dict().items()}, losses
    got, losses = run(True, True)
    ref, ref_losses = run(False, False)
    assert losses == ref_losses
    for k in ref:
        assert np.array_equal(got[k], ref[k]), f"{k}: captured + lazy differs from eager + sweep (max {np.abs(got[k].astype(np.float64) - ref[k]).max():.3e})"

    # ---- C: the five steps against the port
    port = MMoEPort(feats, cfg["hyper"], state0, threads=16)
    want_losses = [port.step(*batches[j], lr=LR, weight_decay=WD)[1] for j in (0, 0, 1, 2, 3)]
    np.testing.assert_allclose(losses, want_losses[2:], rtol=5e-5)
    looked = np.unique(np.concatenate([batches[j][0]["s1"] for j in range(4)]))
    for k, t in port.p.items():
        w = t.detach().numpy()
        g_, w_ = (got[k][looked], w[looked]) if k == big else (got[k], w)
        err = np.abs(g_.astype(np.float64) - w_)
        assert err.max() <= 2.2 * LR * 5, f"{k}: {err.max():.3e}"
        assert (err > 3e-5 + 2e-4 * np.abs(w_)).mean() <= 0.3, k       # (0.09 .. 0.21 over the layouts this path has had)
    idle = np.setdiff1d(np.random.default_rng(0).integers(0, cfg["vocabs"][1], size=200000), looked)
    np.testing.assert_allclose(got[big][idle], port.p[big].detach().numpy()[idle], rtol=3e-6, atol=1e-9)    # five decay-only steps


def _full_table_case(n, batch, hash_seeds):
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    cfg = copy.deepcopy(bench.CONFIGS[n])
    cfg["batch"] = batch
    B = batch
    with torch.device("cuda"):
        model, feats = bench.build_model(cfg, seed=13)
    perturb_product(model, 41)
    for f in feats:
        if f.name in hash_seeds:
            f.hash_seed = hash_seeds[f.name]
    big_names = [f.name for f in feats if f.name in hash_seeds]
    V = {f.name: f.vocab_size for f in feats if hasattr(f, "vocab_size")}
    rng = np.random.default_rng(n)
    steps = []
    for s in range(3):
        x, y = bench.synth_batch(cfg, B, seed=600 + s)
        for name in big_names:
            raw = rng.integers(0, 1 << 40, size=B, dtype=np.int64)
            raw[: B // 4] = raw[B // 4: B // 2]                                 # repeated ids inside a batch
            if s:
                raw[B // 2: B // 2 + B // 8] = steps[0][0][name][: B // 8]      # rows that come back after lagging a step or two
            x[name] = raw
        steps.append((x, y))
    rows = {name: [(mix64(x[name].astype(np.uint64) ^ np.uint64(hash_seeds[name])) % np.uint64(V[name])).astype(np.int64)
                   for x, _y in steps] for name in big_names}
    compact = {name: np.unique(np.concatenate(rows[name])) for name in big_names}
    named = dict(model.named_parameters())
    key_of = {name: next(k for k in named if k.endswith(f"embed_dict.{name}.weight")) for name in big_names}
    # state for the oracle: everything but the big tables, plus their looked-up rows (gathered on the device)
    state0 = {}
    for k, v in model.state_dict().items():
        if k in key_of.values():
            continue
        state0[k] = v.detach().cpu().numpy().copy()
    idle = {}
    for name in big_names:
        w = named[key_of[name]]
        state0[key_of[name]] = w.detach()[torch.from_numpy(compact[name]).cuda()].cpu().numpy().copy()
        cand = np.setdiff1d(rng.integers(0, V[name], size=50000), compact[name])
        idle[name] = (cand, w.detach()[torch.from_numpy(cand).cuda()].cpu().numpy().copy())

    trainer = CTRTrainer(model, "full-tables", optimizer_params={"lr": LR, "weight_decay": WD}, device="cuda")
    trainer.use_graph = False
    model.train()
    losses = []
    for x, y in steps:
        losses.append(float(trainer.train_step({k: torch.from_numpy(v).cuda() for k, v in x.items()}, torch.from_numpy(y).cuda()).detach()))
    torch.cuda.synchronize()
    H.check_errors()
    model.materialize()
    torch.cuda.synchronize()

    ocfg = copy.deepcopy(cfg)
    ocfg["vocabs"] = [len(compact[f"s{i}"]) if f"s{i}" in compact else v for i, v in enumerate(cfg["vocabs"])]
    om = oracle_for(ocfg, state0)
    opt = Adam(lr=LR, weight_decay=WD)
    want_losses = []
    for s, (x, y) in enumerate(steps):
        xo = dict(x)
        for name in big_names:
            xo[name] = np.searchsorted(compact[name], rows[name][s])
        _p, l, g = om.loss_and_grads(xo, y)
        want_losses.append(l)
        opt.step(om.state, g)
    np.testing.assert_allclose(losses, want_losses, rtol=5e-5)
    for name in big_names:
        w = named[key_of[name]].detach()
        got = w[torch.from_numpy(compact[name]).cuda()].cpu().numpy()
        _close_but_for_adam_noise(got, om.state[key_of[name]], 3, f"looked-up rows of {name} ({V[name]} rows)", frac=0.02)
        # rows never looked up: three decay-only steps from their initial values (closed form = the oracle's Adam on g = 0)
        cand, w0 = idle[name]
        st = {"w": w0.astype(np.float64)}
        o2 = Adam(lr=LR, weight_decay=WD)
        for _ in range(3):
            o2.step(st, {"w": np.zeros_like(st["w"])})
        np.testing.assert_allclose(w[torch.from_numpy(cand).cuda()].cpu().numpy(), st["w"], rtol=3e-6, atol=1e-9,
                                   err_msg=f"idle rows of {name}")
    buffers = dict(model.named_buffers())
    for k, want in om.state.items():
        if k in key_of.values() or k.endswith("num_batches_tracked"):
            continue
        got = buffers[k].cpu().numpy() if "running" in k else named[k].detach().cpu().numpy()
        if "running" in k:
            np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 + 0.1 * LR * 9, err_msg=k)
        else:
            _close_but_for_adam_noise(got, want, 3, k, frac=0.05)     # (three steps: see test_cfg2_full part C on Adam's sign noise)


def test_cfg6_full_tables():
    """PPNet over the two 50 M-row hashed tables (BASELINE config 5's "100M-row hashed vocab", PPNet half)."""
    _full_table_case(6, 2048, {"s0": 0x5DEECE66, "s1": 0xB5297A4D})


def test_cfg5_full_tables():
    """HamurSmall over the same tables (the oracle's per-sample adapter products keep the batch small)."""
    _full_table_case(5, 512, {"s0": 0x2545F491, "s1": 0x9E3779B1})


def _captured_lazy_equals_eager_sweep(n, n_steps=4, keep_device_init=False):
    """Config n at its full size: two eager warm-up steps + capture + replays over rotating batches with lazily updated rows
    == the same steps launched eagerly with a dense Adam sweep over every table, BITWISE (state and loss sequence)."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.trainers.graph import GraphedStep
    cfg = copy.deepcopy(bench.CONFIGS[n])
    if not keep_device_init:
        cfg.pop("on_device_init", None)
    B = cfg["batch"]
    batches = [bench.synth_batch(cfg, B, seed=7100 + 10 * n + j) for j in range(n_steps - 1)]
    dev = [({k: torch.from_numpy(v).cuda() for k, v in x.items()}, torch.from_numpy(y).cuda()) for x, y in batches]
    order = [0, 0] + list(range(1, n_steps - 1))                     # batch 0 twice (the warm-up steps), then the others

    def run(lazy, graphed):
        if cfg.get("on_device_init"):                     # 2 x 50 M rows x 64: initialised in HBM (12.8 GB each)
            with torch.device("cuda"):
                model, _feats = bench.build_model(cfg, seed=17)
        else:
            model, _feats = bench.build_model(cfg, seed=17)
        perturb_product(model, 43)
        tr = CTRTrainer(model, f"cfg{n}-full", optimizer_params={"lr": LR, "weight_decay": WD, "lazy_rows": lazy}, device="cuda")
        tr.use_graph = False
        model.train()
        losses = []
        if graphed:
            g = GraphedStep(tr, dev[0][0], dev[0][1], warmup=2)
            for j in order[2:]:
                g.load(*dev[j])
                losses.append(float(g.replay().clone()))
        else:
            for j in order:
                losses.append(float(tr.train_step(*dev[j]).detach()))
            losses = losses[2:]
        torch.cuda.synchronize()
        H.check_errors()
        if hasattr(model, "materialize"):
            model.materialize()                           # lazily updated rows brought up to date
            torch.cuda.synchronize()
        # (tensors above 1 GiB stay on the device: a 64-bit checksum pair per tensor instead of 12.8 GB through host memory)
        out = {}
        for k, v in model.state_dict().items():
            if v.numel() * v.element_size() > (1 << 30):
                flat, s1, s2 = v.detach().reshape(-1).view(torch.int32), 0, 0
                for c0 in range(0, flat.numel(), 1 << 27):
                    w = flat[c0:c0 + (1 << 27)].to(torch.int64)
                    idx = torch.arange(c0, c0 + w.numel(), device=w.device, dtype=torch.int64)
                    s1 += int(w.sum())
                    s2 += int((w * ((idx % 1000003) + 1)).sum())
                out[k] = np.array([s1 % (1 << 62), s2 % (1 << 62)], dtype=np.int64)
            else:
                out[k] = v.cpu().numpy()
        import gc
        g = None
        del model, tr
        gc.collect()
        torch.cuda.empty_cache()
        return out, losses
    got, losses = run(True, True)
    ref, ref_losses = run(False, False)
    assert losses == ref_losses
    for k in ref:
        assert np.array_equal(got[k], ref[k]), f"{k}: captured + lazy differs from eager + sweep"


def _shard_case(n, hash_seeds, n_slice=2048):
    """Config n at the shard bench.py runs (batch 32 768 of BASELINE config 5's 262 144 / 8, both 50 M-row tables at full size):
    (a) captured + lazily updated rows == eager + dense Adam sweep, BITWISE, over rotating batches (the benched regime);
    (b) the product path against the fp64 oracle on a 2 048-row slice of a batch of that shard: eval-mode probabilities within
        1e-4 in the logit and the BCE loss, with the oracle holding the COMPACTED rows of the big tables (the rows the slice
        looks up, gathered from the device tables; exact -- a row's value does not depend on the others)."""
    from scenario_wise_rec import _hip as H
    import gc
    cfg = copy.deepcopy(bench.CONFIGS[n])
    assert cfg["batch"] == 32768
    gc.collect()
    torch.cuda.empty_cache()                  # (each arm below holds 77 GB of tables + Adam state: nothing of earlier tests may linger)
    _captured_lazy_equals_eager_sweep(n, keep_device_init=True)
    gc.collect()
    torch.cuda.empty_cache()
    with torch.device("cuda"):
        model, feats = bench.build_model(cfg, seed=13)
    perturb_product(model, 41)
    for f in feats:
        if f.name in hash_seeds:
            f.hash_seed = hash_seeds[f.name]
    x, y = bench.synth_batch(cfg, cfg["batch"], seed=900 + n)
    rng = np.random.default_rng(n)
    for name in hash_seeds:
        x[name] = rng.integers(0, 1 << 40, size=cfg["batch"], dtype=np.int64)
    xs = {k: v[:n_slice] for k, v in x.items()}
    ys = y[:n_slice]
    model.eval()
    with torch.no_grad():
        p = model({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in xs.items()}).cpu().numpy()
    torch.cuda.synchronize()
    H.check_errors()
    V = {f.name: f.vocab_size for f in feats if hasattr(f, "vocab_size")}
    named = dict(model.named_parameters())
    state0, xo, ocfg = {}, dict(xs), copy.deepcopy(cfg)
    big_keys = {}
    for name, seed in hash_seeds.items():
        rows = (mix64(xs[name].astype(np.uint64) ^ np.uint64(seed)) % np.uint64(V[name])).astype(np.int64)
        compact = np.unique(rows)
        key = next(k for k in named if k.endswith(f"embed_dict.{name}.weight"))
        big_keys[key] = named[key].detach()[torch.from_numpy(compact).cuda()].cpu().numpy().copy()
        xo[name] = np.searchsorted(compact, rows)
        ocfg["vocabs"][int(name[1:])] = len(compact)
    for k, v in model.state_dict().items():
        state0[k] = big_keys[k] if k in big_keys else v.detach().cpu().numpy().copy()
    om = oracle_for(ocfg, state0)
    op = om.predict(xo)
    assert_probs_close(p, op, tol=1e-4)
    bce = lambda q: float(-np.mean(ys * np.log(np.clip(q, 1e-38, None)) + (1 - ys) * np.log(np.clip(1 - q, 1e-38, None))))
    assert abs(bce(p.astype(np.float64)) - bce(op.astype(np.float64))) < 2e-6 * max(1.0, abs(bce(op.astype(np.float64))))


def _shard_step_gradients(n, hash_seeds, max_kinked, threads=32):
    """ONE training step of config n at the benched shard (batch 32 768, both 50 M-row hashed tables at full size) against an
    fp64 oracle of the SAME batch: train-mode probabilities (1e-4 in the logit), loss, and EVERY parameter gradient -- dense
    arena gradients and the row lists of the two big tables (gradients were pinned at 512 .. 2 048 rows
    only, and batch-size-dependent dispatch -- side streams, grid fill, the pipelined rowmat, bnmix rows per workgroup, v4 tails
    -- means "same kernels" does not follow from the small shapes).  The oracle is oracle/torch_port.TorchPort in fp64 (pinned on
    the golden vectors and, in this fp64 / re-associated-adapter form, on the numpy oracle: tests/test_oracle_golden.py) holding
    the COMPACTED rows of the big tables (exact: a row's gradient depends on that row only).  ReLU units within fp32 rounding of
    zero: _golden.KinkTolerantGradCheck with the configuration's MEASURED number of affected tensors as the cap."""
    from oracle.torch_port import TorchPort
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    import gc
    cfg = copy.deepcopy(bench.CONFIGS[n])
    B = cfg["batch"]
    assert B == 32768
    gc.collect()
    torch.cuda.empty_cache()
    with torch.device("cuda"):
        model, feats = bench.build_model(cfg, seed=13)
    perturb_product(model, 41)
    for f in feats:
        if f.name in hash_seeds:
            f.hash_seed = hash_seeds[f.name]
    x, y = bench.synth_batch(cfg, B, seed=910 + n)
    rng = np.random.default_rng(50 + n)
    for name in hash_seeds:
        raw = rng.integers(0, 1 << 40, size=B, dtype=np.int64)
        raw[: B // 8] = raw[B // 8: B // 4]                                  # repeated ids inside the batch
        x[name] = raw
    V = {f.name: f.vocab_size for f in feats if hasattr(f, "vocab_size")}
    named = dict(model.named_parameters())
    state0, xo, ocfg, compact = {}, dict(x), copy.deepcopy(cfg), {}
    big_keys = {}
    for name, seed in hash_seeds.items():
        rows = (mix64(x[name].astype(np.uint64) ^ np.uint64(seed)) % np.uint64(V[name])).astype(np.int64)
        compact[name] = np.unique(rows)
        key = next(k for k in named if k.endswith(f"embed_dict.{name}.weight"))
        big_keys[key] = name
        state0[key] = named[key].detach()[torch.from_numpy(compact[name]).cuda()].cpu().numpy().copy()
        xo[name] = np.searchsorted(compact[name], rows)
        ocfg["vocabs"][int(name[1:])] = len(compact[name])
    for k, v in model.state_dict().items():
        if k not in big_keys:
            state0[k] = v.detach().cpu().numpy().copy()

    trainer = CTRTrainer(model, f"cfg{n}-shard-step", optimizer_params={"lr": LR, "weight_decay": WD}, device="cuda")
    trainer.use_graph = False
    model.train()
    p = model({k: torch.from_numpy(v).cuda() for k, v in x.items()})
    loss = trainer.criterion(p, torch.from_numpy(y).cuda())
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    H.check_errors()

    dense = [Dense(f"d{i}") for i in range(ocfg["n_dense"])]
    sparse = [Sparse(f"s{i}", v, ocfg["embed_dim"]) for i, v in enumerate(ocfg["vocabs"])]
    hyper = copy.deepcopy(ocfg["hyper"])
    if ocfg["family"] == "PPNet":
        nid = ocfg["id_features"]
        hyper.update(id_features=sparse[:nid], agn_features=dense + sparse[nid:])
    else:
        hyper["features"] = dense + sparse
    st64 = {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in state0.items()}
    port = TorchPort(ocfg["family"], hyper, st64, dtype=torch.float64, threads=threads, materialize_adapter=False, track_buffers=False)
    op, oloss, ograds = port.loss_and_grads(xo, y)
    assert_probs_close(p.detach().cpu().numpy(), op, tol=1e-4)
    assert abs(float(loss.detach()) - oloss) < 2e-6 * max(1.0, abs(oloss))
    kinks = KinkTolerantGradCheck()
    for k, g in ograds.items():
        prm = named[k]
        sg = getattr(prm, "_swr_sparse_grad", None)
        if k in big_keys:
            assert sg is not None, f"{k}: the 50 M-row table must take a row-sparse gradient"
            r, gg = sg[0].cpu().numpy(), sg[1].cpu().numpy().astype(np.float64)
            keep = r >= 0
            assert np.isin(r[keep], compact[big_keys[k]]).all(), f"{k}: a gradient row nobody looked up"
            got = np.zeros(g.shape)
            np.add.at(got, np.searchsorted(compact[big_keys[k]], r[keep]), gg[keep])
        elif sg is not None:
            r, gg = sg[0].cpu().numpy(), sg[1].cpu().numpy().astype(np.float64)
            got = np.zeros(tuple(prm.shape))
            np.add.at(got, r[r >= 0], gg[r >= 0])
        else:
            assert prm.grad is not None, f"{k}: no gradient"
            got = prm.grad.cpu().numpy()
        kinks.check(got, g, 2e-4 * max(1e-6, float(np.abs(g).max())) + 3e-7, k)
    kinks.finish(max_kinked)
    for k, prm in named.items():
        if k not in ograds:                  # reference grad None (PPNet's agnostic tables): untouched
            assert not getattr(prm, "_swr_touched", False), k


def test_cfg5_shard_step_gradients():
    """HamurSmall, batch 32 768: the pipelined rowmat at D = 8, the side-stream step, every gradient against fp64."""
    _shard_step_gradients(5, {"s0": 0x2545F491, "s1": 0x9E3779B1}, max_kinked=CFG5_SHARD_KINKED)


def test_cfg6_shard_step_gradients():
    """PPNet, batch 32 768: every gradient against fp64 (the agnostic tables stay untouched)."""
    _shard_step_gradients(6, {"s0": 0x5DEECE66, "s1": 0xB5297A4D}, max_kinked=CFG6_SHARD_KINKED)


def test_cfg5_shard():
    """HamurSmall at bench.py --config 5's shard (batch 32 768, 2 x 50 M hashed rows x 64)."""
    _shard_case(5, {"s0": 0x2545F491, "s1": 0x9E3779B1})


def test_cfg6_shard():
    """PPNet at bench.py --config 6's shard (batch 32 768, the same tables)."""
    _shard_case(6, {"s0": 0x5DEECE66, "s1": 0xB5297A4D})


def test_cfg3_full():
    """Ali-CCP 3-domain STAR at the per-GPU shard of BASELINE config 3 (batch 131 072 / 8), full vocabularies."""
    cfg = copy.deepcopy(bench.CONFIGS[3])
    assert cfg["batch"] == 16384 and max(cfg["vocabs"]) == 467298
    run_config(cfg, seed=1, full_size=True, max_kinked=CFG3_KINKED)
    _captured_lazy_equals_eager_sweep(3)


def test_cfg4_full():
    """Mind 4-domain PLE at the per-GPU shard of BASELINE config 4 (batch 65 536 / 8), the 748 000-row user table."""
    cfg = copy.deepcopy(bench.CONFIGS[4])
    assert cfg["batch"] == 8192 and max(cfg["vocabs"]) == 748000
    run_config(cfg, seed=4, full_size=True, max_kinked=CFG4_KINKED)
    _captured_lazy_equals_eager_sweep(4)
