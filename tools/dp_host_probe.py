"""Is the N > 1 step (three graphs + two collectives per step) bound by the HOST?  World size 1 over RCCL on one GPU:
host enqueue time per step (no synchronisation inside the loop) against the GPU's time per step.
    SWR_BENCH_FORCE_DP=1 python tools/dp_host_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import bench  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from scenario_wise_rec.parallel import DataParallelStep
    from scenario_wise_rec.trainers import CTRTrainer
    cfg = bench.CONFIGS[2]
    B = cfg["batch"]
    model, _ = bench.build_model(cfg)
    tr = CTRTrainer(model, "probe", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda:0")
    model.train()
    bs = []
    for k in range(4):
        xh, yh = bench.synth_batch(cfg, B, seed=100 + k)
        bs.append(({n: torch.from_numpy(v).to(dev) for n, v in xh.items()}, torch.from_numpy(yh).to(dev)))
    step = DataParallelStep(tr, world_size=1)
    step.capture(*bs[0])
    for i in range(10):
        step.load(*bs[i % 4]); step.replay()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        step.load(*bs[i % 4]); step.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step (queue drained in "
          f"{1e3 * (t2 - t1):.2f} ms after the loop)")
    # per call: how long the host spends in each piece of replay()
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(100):
        step.load(*bs[i % 4]); step.replay()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
