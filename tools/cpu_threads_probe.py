"""Which torch thread count suits the GPU box's host for the cpu_baseline leg (oracle/torch_port.py)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")]
import torch
import bench
from oracle.nn import Dense, Sparse
from oracle.torch_port import MMoEPort
cfg = bench.CONFIGS[2]
B = cfg["batch"]
feats = [Dense(f"d{i}") for i in range(cfg["n_dense"])] + [Sparse(f"s{i}", v, cfg["embed_dim"]) for i, v in enumerate(cfg["vocabs"])]
model, _ = bench.build_model(cfg)
state = {k: v.detach().numpy() for k, v in model.state_dict().items()}
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
x, y = bench.synth_batch(cfg, B, seed=1)
for th in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    port = MMoEPort(feats, cfg["hyper"], state, threads=th)
    t0 = time.perf_counter(); port.step(x, y); t1 = time.perf_counter(); port.step(x, y); t2 = time.perf_counter()
    print(f"threads {th}: warm-up {t1 - t0:.2f} s, step {t2 - t1:.2f} s = {B / (t2 - t1):.0f} samples/s", flush=True)
