"""The oracle (oracle/, numpy) against every golden vector the reference
produced (tests/golden/*.npz).  This is what pins the oracle: forward in eval
and train mode, every parameter gradient, BN buffers, and the state after 1
and 3 Adam steps, for all in-scope model families and the edge cases
(empty domain, out-of-range domain id, narrow input dtypes, odd batch sizes,
DataParallel sharding)."""
import numpy as np
import pytest

from _golden import Case, assert_probs_close, case_names, make_oracle, assert_state_close
from oracle.optim import Adam


SINGLE = [n for n in case_names() if "_dp" not in n]
DP = [n for n in case_names() if "_dp" in n]


@pytest.mark.parametrize("name", SINGLE)
def test_eval_forward(name):
    c = Case(name)
    m = make_oracle(c)
    x, _ = c.batch(0)
    assert_probs_close(m.predict(x), c.z["eval_probs"], tol=2e-5)


@pytest.mark.parametrize("name", SINGLE)
def test_train_step_grads_and_state(name):
    c = Case(name)
    m = make_oracle(c)
    opt = Adam(lr=c.meta["lr"], weight_decay=c.meta["weight_decay"])
    losses = []
    for s in range(3):
        x, y = c.batch(s)
        probs, loss, grads = m.loss_and_grads(x, y)
        losses.append(loss)
        if s == 0:
            assert_probs_close(probs, c.z["train_probs"], tol=2e-5)
            want = c.group("grad")
            assert set(grads) == set(want), (set(grads) ^ set(want))
            for k, g in want.items():
                scale = max(1e-6, float(np.abs(g).max()))
                np.testing.assert_allclose(grads[k], g, rtol=0, atol=2e-4 * scale + 3e-7, err_msg=k)
        opt.step(m.state, grads)
        if s in (0, 2):
            want = c.group("state1" if s == 0 else "state3")
            for k, v in want.items():
                if k.endswith("num_batches_tracked"):
                    assert int(m.state[k]) == int(v), k
                else:
                    assert_state_close(m.state[k], v, c, k, s + 1)
    np.testing.assert_allclose(losses, c.z["losses"], rtol=2e-5)
    x, _ = c.batch(0)
    assert_probs_close(m.predict(x), c.z["eval3_probs"], tol=1e-4)


@pytest.mark.parametrize("name", DP)
def test_dataparallel_semantics(name):
    """N row shards, each with its own BN batch statistics, loss = mean over
    the global batch, gradients summed, shard 0's buffers kept
    (`ctr_trainer.py:45-47`)."""
    c = Case(name)
    n = c.meta["n_shards"]
    x, y = c.batch(0)
    B = len(y)
    sh = B // n
    total, probs, state_after = {}, [], None
    for r in range(n):
        m = make_oracle(c)
        xs = {k: v[r * sh:(r + 1) * sh] for k, v in x.items()}
        p, _, g = m.loss_and_grads(xs, y[r * sh:(r + 1) * sh])
        probs.append(p)
        for k, v in g.items():
            total[k] = total.get(k, 0) + v * (sh / B)       # local mean-loss -> global mean-loss
        if r == 0:
            state_after = m.state
    assert_probs_close(np.concatenate(probs), c.z["train_probs"], tol=2e-5)
    for k, g in c.group("grad").items():
        scale = max(1e-6, float(np.abs(g).max()))
        np.testing.assert_allclose(total[k], g, rtol=0, atol=2e-4 * scale + 3e-7, err_msg=k)
    Adam(lr=c.meta["lr"], weight_decay=c.meta["weight_decay"]).step(state_after, total)
    for k, v in c.group("state1").items():
        if not k.endswith("num_batches_tracked"):
            np.testing.assert_allclose(state_after[k], v, rtol=1e-4, atol=2e-5, err_msg=k)


def test_torch_port_matches_the_oracle():
    """oracle/torch_port.py (the multi-threaded CPU port bench.py times as `cpu_baseline`) computes the step the
    numpy oracle computes: forward, loss, every gradient and the state after one Adam step on the golden MMoE case."""
    from _golden import Case, oracle_features
    from oracle.torch_port import MMoEPort
    c = Case("mmoe")
    feats = oracle_features(c.schemas[0])
    port = MMoEPort(feats, c.hyper, c.group("state0"))
    x, y = c.batch(0)
    p, loss, grads = port.loss_and_grads(x, y)
    np.testing.assert_allclose(p, c.z["train_probs"], rtol=1e-5, atol=1e-6)
    assert abs(loss - float(c.z["loss0"])) < 1e-6
    for k, g in c.group("grad").items():
        np.testing.assert_allclose(grads[k], g, rtol=0, atol=2e-5 * max(1e-6, float(np.abs(g).max())) + 1e-8, err_msg=k)
    port2 = MMoEPort(feats, c.hyper, c.group("state0"))
    port2.step(x, y, lr=c.meta["lr"], weight_decay=c.meta["weight_decay"])
    for k, v in c.group("state1").items():
        got = (port2.p.get(k, port2.buf.get(k))).detach().numpy()
        if k.endswith("num_batches_tracked"):
            assert int(got) == int(v)
        else:
            assert_state_close(got, v, c, k, 1)
