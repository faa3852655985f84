#!/usr/bin/env python3
"""Time the device evaluation metrics (ops.eval_metrics) against the reference's host path (.tolist() + sklearn)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import numpy as np
import torch
from sklearn.metrics import log_loss, roc_auc_score
from scenario_wise_rec import ops
n, D = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, 5
g = torch.Generator(device="cuda").manual_seed(0)
p = torch.rand(n, device="cuda", generator=g)
y = (torch.rand(n, device="cuda", generator=g) < 0.2).float()
d = torch.randint(0, D, (n,), device="cuda", generator=g)
ops.eval_metrics(p, y, d, D); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    rows, pos, two_u, ll = ops.eval_metrics(p, y, d, D)
t_dev = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
pl, yl, dl = p.cpu().tolist(), y.cpu().tolist(), d.cpu().tolist()
for k in range(D):
    t = [a for a, dd in zip(yl, dl) if dd == k]
    q = [a for a, dd in zip(pl, dl) if dd == k]
    log_loss(t, q); roc_auc_score(t, q)
auc = roc_auc_score(yl, pl); ll_h = log_loss(yl, pl)
t_host = time.perf_counter() - t0
print(f"n={n} D={D}: device {t_dev * 1e3:.2f} ms (incl. the host read-back), reference host path {t_host:.2f} s; "
      f"auc dev {two_u[D] / (2.0 * pos[D] * (rows[D] - pos[D])):.15f} host {auc:.15f}; logloss dev {ll[D] / rows[D]:.15f} host {ll_h:.15f}")
