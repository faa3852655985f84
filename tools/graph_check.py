#!/usr/bin/env python3
"""Bench-scale check: N steps eager vs (2 eager + capture + N-2 replays) must give identical parameters."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def run(use_graph, n=int(os.environ.get("N_STEPS", "5")), B=None):
    cfg = dict(bench.CONFIGS[2])
    if B:
        cfg["batch"] = B
    from scenario_wise_rec.trainers import CTRTrainer
    model, _ = bench.build_model(cfg)
    tr = CTRTrainer(model, "chk", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
    model.train()
    xh, yh = bench.synth_batch(cfg, cfg["batch"], seed=int(os.environ.get("SEED", "1")))
    x = {k: torch.from_numpy(v).cuda() for k, v in xh.items()}
    y = torch.from_numpy(yh).cuda()
    losses = []
    if use_graph:
        from scenario_wise_rec.trainers.graph import GraphedStep
        g = GraphedStep(tr, x, y, warmup=2)
        for _ in range(n - 2):
            if os.environ.get("BURST"):
                g.replay()                       # back-to-back launches, no host sync in between
            else:
                losses.append(float(g.replay()))
    else:
        for _ in range(n):
            losses.append(float(tr.train_step(x, y).detach()))
    torch.cuda.synchronize()
    from scenario_wise_rec import _hip as H
    H.check_errors()
    return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else None
    if os.environ.get("GRAPH_ONLY"):
        lg, sg = run(True, B=B)
        print("graph losses", [round(v, 6) for v in lg])
        sys.exit(0)
    le, se = run(False, B=B)
    lg, sg = run(True, B=B)
    print("eager losses", le)
    print("graph losses", lg)
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    for k in bad[:10]:
        d = (se[k].double() - sg[k].double()).abs()
        print(f"  differs: {k} max {d.max().item():.3e} n {int((d > 0).sum())}/{d.numel()}")
    print("IDENTICAL" if not bad else f"{len(bad)} tensors differ")
