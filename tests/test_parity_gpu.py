"""Model-level parity on the GPU: the HIP path (product models, through the C ABI) against the golden
vectors the reference produced (tests/golden/*.npz) -- forward in eval and train mode, every parameter
gradient, the state after 1 and 3 optimisation steps driven by CTRTrainer, and the loss sequence.

Tolerance: the north star's 1e-4 on fp32 logits; rows zeroed by the domain select must be exactly 0.0
(bit-exact routing).  Gradients: 2e-4 of the largest entry of each tensor (+3e-7 noise floor, see
tests/_golden.state_atol)."""
import numpy as np
import pytest
import torch

from _golden import Case, assert_probs_close, build_product_model, case_names, state_atol, to_device

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
                    np.testing.assert_allclose(got[k], v, rtol=1e-4, atol=state_atol(c, k, s + 1),
                                               err_msg=f"step{s + 1}:{k}")
    H.check_errors()
    np.testing.assert_allclose(losses, c.z["losses"], rtol=5e-5)
    model.eval()
    with torch.no_grad():
        p = model(to_device(c.batch(0)[0])).cpu().numpy()
    assert_probs_close(p, c.z["eval3_probs"], tol=2e-4)
