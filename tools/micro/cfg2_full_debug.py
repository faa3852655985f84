"""Debug aid for tests/test_full_size_gpu.py::test_cfg2_full: per-tensor differences between the product and the torch-CPU port
after k steps (eager or graphed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
from _golden import perturb_product, logit
from oracle.nn import Dense, Sparse
from oracle.torch_port import MMoEPort
from scenario_wise_rec import _hip as H
from scenario_wise_rec.trainers import CTRTrainer

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = bench.CONFIGS[2]
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["batch"]
model, _ = bench.build_model(cfg, seed=7)
perturb_product(model, 31)
state0 = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
batches = [bench.synth_batch(cfg, B, seed=4000 + j) for j in range(5)]
trainer = CTRTrainer(model, "dbg", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
trainer.use_graph = False
model.train()
dev = [({k: torch.from_numpy(v).cuda() for k, v in x.items()}, torch.from_numpy(y).cuda()) for x, y in batches]
feats = [Dense(f"d{i}") for i in range(cfg["n_dense"])] + [Sparse(f"s{i}", v, cfg["embed_dim"]) for i, v in enumerate(cfg["vocabs"])]
port = MMoEPort(feats, cfg["hyper"], state0, threads=16)
seq = [0, 0, 1, 2, 3][:n_steps]
# gradients of the first step
p = model(dev[0][0]); loss = trainer.criterion(p, dev[0][1]); model.zero_grad(); loss.backward(); torch.cuda.synchronize()
pp, pl, pg = port.loss_and_grads(*batches[0])
print("step-1 forward: max logit err", np.abs(logit(p.detach().cpu().numpy()) - logit(pp)).max(), "loss", float(loss), pl)
named = dict(model.named_parameters())
for k, g in pg.items():
    prm = named[k]
    sg = getattr(prm, "_swr_sparse_grad", None)
    if sg is not None:
        r, gg = sg[0].cpu().numpy(), sg[1].cpu().numpy().astype(np.float64)
        got = np.zeros(tuple(prm.shape)); np.add.at(got, r[r >= 0], gg[r >= 0])
    else:
        got = prm.grad.cpu().numpy()
    e = np.abs(got - g).max() / (np.abs(g).max() + 1e-30)
    if e > 1e-4:
        print(f"  grad {k}: rel err {e:.2e} (max |g| {np.abs(g).max():.2e})")
model.zero_grad()
for j in seq:
    trainer.train_step(*dev[j])
    port.step(*batches[j], lr=1e-3, weight_decay=1e-5)
torch.cuda.synchronize(); H.check_errors()
got = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
for k, t in port.p.items():
    w = t.detach().numpy()
    err = np.abs(got[k].astype(np.float64) - w)
    if err.max() > 2e-5:
        print(f"{k:44s} max err {err.max():.2e}  frac > 3e-5: {(err > 3e-5 + 2e-4 * np.abs(w)).mean():.2e}")
with torch.no_grad():
    pr = model(dev[4][0]).cpu().numpy(); wp = port.forward(batches[4][0]).numpy()
print("probe max logit err", np.abs(logit(pr) - logit(wp)).max())
