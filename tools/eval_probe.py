#!/usr/bin/env python3
"""Eval-mode forward of the config-2 model (batch 65 536): routed head vs the dense all-domain path + select."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import torch
import bench
cfg = bench.CONFIGS[2]
model, feats = bench.build_model(cfg)
model.to("cuda").eval()
xh, _ = bench.synth_batch(cfg, cfg["batch"], seed=1)
x = {k: torch.from_numpy(v).cuda() for k, v in xh.items()}
def run(tag):
    with torch.no_grad():
        for _ in range(3):
            p = model(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            p = model(x)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            g.replay()
        e1.record(); e1.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per eval forward of {cfg['batch']} rows (hipGraph replay)")
    return p.clone()
a = run("routed head")
os.environ["SWR_ROUTED_EVAL"] = "0"
b = run("dense all-domain path + select")
print("max |difference| of the probabilities:", float((a - b).abs().max()))
