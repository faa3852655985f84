#!/usr/bin/env python3
"""bench.py -- train samples/s of the multi-domain CTR hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1..6] [--no-graph] [--no-cpu-baseline]

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): KuaiRand-shaped 5-domain
MMoE with 4 experts, embed_dim 16, batch 65 536 per GPU, synthetic device-resident inputs, random-init
weights.  One step = the reference's training step (`trainers/ctr_trainer.py:67-73`): fused lookup ->
experts/gates -> towers -> domain select -> BCE -> backward (all parameter gradients, embedding tables
included) -> Adam(lr 1e-3, weight_decay 1e-5) on every parameter.  fp32 results end to end (every fp32 product as six
bf16 MFMA products with fp32 accumulation).  The timed region rotates four device-resident batches through the captured
step (one batch-load launch + one graph replay per step), so rows are not cache-hot and the lazy Adam has real lag.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, measured live with
HIP events on the launch stream) and `cpu_baseline` (the numpy oracle timed on this box's host cores on a
bounded sample).  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), data parallel; the metric's
configuration defaults to STRONG scaling there (the 65 536-row batch sharded by row, ctr_trainer.py:45-47) with the weak
regime in `config.other_scaling`; the N = 1 line carries `config.strong_shard` (the 8 192-row shard timed on this GPU).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md; not the 2:1-sparsity headline)

# KuaiRand-1K schema (scripts/run_kuairand_ctr_multi_domain.py:15-57, scripts/data/kuairand/load_data_1k.py:23-49,
# README.md:137): user_id, video_id, 8 user categoricals, 18 one-hot groups, 4 video categoricals; 4 dense
KUAIRAND_VOCABS = [1000, 4371900, 8, 2, 3, 2, 8, 8, 7, 7,
                   3, 8, 51, 1472, 16, 35, 4, 119, 455, 8, 6, 6, 3, 3, 3, 3, 3, 3,
                   3, 8, 200, 300]
KUAIRAND_DOMAIN_SHARES = [0.207, 0.666, 0.077, 0.035, 0.016]      # README.md:42-46

# Ali-CCP field cardinalities (scripts/run_ali_ccp_ctr_ranking_multi_domain.py:11-34 names the 23 sparse + 8 dense fields;
# README.md:138: 238 k users, 467 k items; the other sizes are this build's choice, of the public dataset's magnitudes)
ALICCP_VOCABS = [238635, 98, 14, 3, 8, 4, 4, 3, 5, 467298, 6929, 263942, 106399, 5888, 104830, 51878, 37148, 4,
                 5853, 105622, 53843, 31858, 3]

CONFIGS = {
    # the metric's configuration (BASELINE.json configs[1])
    2: dict(name="kuairand_mmoe4_e16_b65536", family="MMOE", vocabs=KUAIRAND_VOCABS, embed_dim=16, n_dense=4,
            batch=65536, domain_shares=KUAIRAND_DOMAIN_SHARES,
            hyper=dict(domain_num=5, n_expert=4, expert_params={"dims": [32]}, tower_params={"dims": [16]})),
    # the other BASELINE.json configurations, runnable with --config N (not the bench line; configs 3-5 are 8-GPU
    # configurations: `batch` is the per-GPU shard, global batch / 8)
    1: dict(name="movielens_sharedbottom_e8_b4096", family="SharedBottom", vocabs=[6040, 3706, 2, 21, 3439, 18],
            embed_dim=8, n_dense=1, batch=4096, domain_shares=[0.211, 0.395, 0.394],
            hyper=dict(domain_num=3, bottom_params={"dims": [128]}, tower_params={"dims": [8]})),
    3: dict(name="aliccp_star_e16_b131072_over8", family="Star", vocabs=ALICCP_VOCABS, embed_dim=16, n_dense=8,
            batch=131072 // 8, domain_shares=[0.378, 0.0075, 0.615],
            hyper=dict(num_domains=3, fcn_dims=[256, 128, 64, 32, 16, 8], aux_dims=[16])),
    4: dict(name="mind_ple_e32_b65536_over8", family="PLE", vocabs=[748000, 20000, 300], embed_dim=32, n_dense=0,
            batch=65536 // 8, domain_shares=[0.459, 0.198, 0.180, 0.163],
            hyper=dict(domain_num=4, n_level=1, n_expert_specific=2, n_expert_shared=1,
                       expert_params={"dims": [64, 32]}, tower_params={"dims": [16]})),
    5: dict(name="synthetic8_hamursmall_100Mrows_e64_b262144_over8", family="HamurSmall",
            vocabs=[50_000_000, 50_000_000, 1000, 1000, 100, 10], embed_dim=64, n_dense=4, batch=262144 // 8,
            domain_shares=[1.0] * 8, on_device_init=True,
            hyper=dict(domain_num=8, fcn_dims=[256, 128], hyper_dims=[64], k=35)),
    6: dict(name="synthetic8_ppnet_100Mrows_e64_b262144_over8", family="PPNet",
            vocabs=[50_000_000, 50_000_000, 1000, 1000, 100, 10, 8], embed_dim=64, n_dense=4, batch=262144 // 8,
            domain_shares=[1.0] * 8, on_device_init=True, id_features=2,
            hyper=dict(domain_num=8, fcn_dims=[128, 64, 32])),
}


def synth_batch(cfg, B, seed, zipf=True):
    """Synthetic batch (numpy, host): Zipf(1.05) ids clipped to V for tables with V >= 1e4, uniform otherwise,
    dense U[0,1), domain ~ the dataset's interaction shares, labels Bernoulli(0.2)."""
    rng = np.random.default_rng(seed)
    x = {}
    for i, v in enumerate(cfg["vocabs"]):
        if zipf and v >= 10000:
            ids = np.minimum(rng.zipf(1.05, size=B) - 1, v - 1)
            ids = (ids * 2654435761) % v            # scatter the popular rows over the table
        else:
            ids = rng.integers(0, v, size=B)
        x[f"s{i}"] = ids.astype(np.int64)
    for i in range(cfg["n_dense"]):
        x[f"d{i}"] = rng.random(B).astype(np.float32)
    p = np.asarray(cfg["domain_shares"], dtype=np.float64)
    x["domain_indicator"] = rng.choice(len(p), size=B, p=p / p.sum()).astype(np.int64)
    y = (rng.random(B) < 0.2).astype(np.float32)
    return x, y


def build_model(cfg, seed=2024):
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.models import multi_domain as md
    torch.manual_seed(seed)
    dense = [DenseFeature(f"d{i}") for i in range(cfg["n_dense"])]
    sparse = [SparseFeature(f"s{i}", v, cfg["embed_dim"]) for i, v in enumerate(cfg["vocabs"])]
    feats = dense + sparse
    hyper = copy.deepcopy(cfg["hyper"])    # (the HAMUR constructors append k*k to the caller's hyper_dims, hamur.py:77,288)
    if cfg["family"] == "PPNet":           # id part (user, item) | scenario-agnostic part (ppnet.py:33)
        nid = cfg["id_features"]
        return md.PPNet(sparse[:nid], dense + sparse[nid:], **hyper), feats
    return getattr(md, cfg["family"])(feats, **hyper), feats


def gather_bytes_per_sample(cfg, idx_bytes=8):
    fs, e, fd = len(cfg["vocabs"]), cfg["embed_dim"], cfg["n_dense"]
    k0 = fs * e + fd
    return fs * (idx_bytes + 4 * e) + 4 * fd + 4 * k0          # 4 384 B at config 2


def time_kernel_events(fn, iters, stream, reps=20):
    """Average duration (ms) of one `fn` launch: `reps` launches are captured into a hipGraph (so the host cannot be
    the bottleneck), the graph is replayed `iters` times on `stream`, bracketed by HIP events recorded on that stream."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                fn()
        g.replay()
        start.record(stream)
        for _ in range(iters):
            g.replay()
        stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / (iters * reps)


def cpu_baseline(cfg, seconds_budget=20.0):
    """The reference's CPU path restated (oracle/torch_port.py: torch CPU operators, every family pinned on the golden vectors),
    timed on this box: the same model and batch shape as the GPU run, fwd + BCE + bwd + dense Adam over every table row
    (SURVEY.md 8d).  torch's CPU kernels do not scale with the thread count on this step (on the 2 x 64-core host of the GPU
    box 8-16 threads give 84 K samples/s at config 2, 64 threads 47 K, 256 threads 1 K), so the leg first tries 8 / 16 / 32 / 64
    threads with one step each and then times >= 3 steps at the fastest setting; `cores` is that thread count.
    Bounded sample: tables above 2^21 rows are cut to 2^21 rows (ids modulo) -- the CPU step's dense Adam over 2 x 50 M x 64
    fp32 rows (config 5 / 6: 100 GB of parameter + gradient + moments) does not fit a test host -- and HAMUR, whose reference
    forward materialises a [B, 128, 32] weight per sample, adapter and domain, runs at batch 4 096; both are named in `sample`."""
    from oracle.nn import Dense, Sparse
    from oracle.torch_port import TorchPort
    ncpu = os.cpu_count() or 1
    cap = 1 << 21
    cut = [v for v in cfg["vocabs"] if v > cap]
    ccfg = dict(cfg, vocabs=[min(v, cap) for v in cfg["vocabs"]], on_device_init=False)
    B = min(cfg["batch"], 4096) if cfg["family"].startswith("Hamur") else cfg["batch"]
    dense = [Dense(f"d{i}") for i in range(ccfg["n_dense"])]
    sparse = [Sparse(f"s{i}", v, ccfg["embed_dim"]) for i, v in enumerate(ccfg["vocabs"])]
    hyper = copy.deepcopy(ccfg["hyper"])
    if ccfg["family"] == "PPNet":
        nid = ccfg["id_features"]
        hyper.update(id_features=sparse[:nid], agn_features=dense + sparse[nid:])
    else:
        hyper["features"] = dense + sparse
    model, _ = build_model(ccfg)
    state = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    del model
    batches = []
    for j in range(2):
        xh, yh = synth_batch(cfg, B, seed=1 + j)
        for i, v in enumerate(cfg["vocabs"]):
            if v > cap:
                xh[f"s{i}"] = xh[f"s{i}"] % cap
        batches.append((xh, yh))
    port = TorchPort(ccfg["family"], hyper, state, threads=min(8, ncpu))
    port.step(*batches[0])                      # warm-up (allocations, Adam state for every table row)
    tried = {}
    for th in sorted({min(t, ncpu) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        port.step(*batches[1])
        tried[th] = time.perf_counter() - t0
        if tried[th] > 2.5 * min(tried.values()) or sum(tried.values()) > seconds_budget:
            break                               # more threads only get slower from here / the budget is spent on probing
    cores = min(tried, key=tried.get)
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < seconds_budget / 2 and n < 12):
        port.step(*batches[n % 2])
        n += 1
        if n >= 1 and time.perf_counter() - t0 > 1.5 * seconds_budget:
            break
    dt = time.perf_counter() - t0
    rows = sum(ccfg["vocabs"])
    out = {"value": n * B / dt, "unit": "samples/s", "cores": cores, "cores_tried": sorted(tried), "host_logical_cpus": ncpu,
           "kind": "port",
           "sample": f"{n} full steps (fwd+BCE+bwd+dense Adam on all {rows} table rows) of {ccfg['family']} at batch {B} after warm-up: "
                     f"torch-CPU port of the reference step (oracle/torch_port.py) at the fastest of {sorted(tried)} threads "
                     f"({cores}; host has {ncpu} logical CPUs, one step took " + ", ".join(f"{th}: {t:.2f} s" for th, t in sorted(tried.items())) + ")"
                     + (f"; tables of {cut} rows cut to {cap} rows (ids modulo): the CPU's dense Adam state for the full tables does not fit" if cut else "")
                     + (f"; batch {B} instead of {cfg['batch']}: the reference's per-sample adapter weights [B, 128, 32] per domain" if B != cfg["batch"] else "")}
    if cfg["name"].startswith("kuairand_mmoe4_e16_b65536") and B == 65536:
        out["reference_in_build_container"] = {"value": 46800.0, "unit": "samples/s", "cores": 8,
                                               "note": "the reference itself, same config, 8 vCPU build container (SURVEY.md section 6)"}
    return out


def _stage(msg):
    if os.environ.get("SWR_BENCH_VERBOSE"):
        torch.cuda.synchronize()
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


N_ROTATE = 4      # device-resident batches rotated through the captured step inside the timed region


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong: the configuration's batch is the GLOBAL batch, split by row over the GPUs (65 536 -> 8 192 per "
                         "GPU at 8) -- what the reference's DataParallel does with one batch (ctr_trainer.py:45-47) and the default "
                         "of the metric's configuration for N > 1; weak: that batch PER GPU (the default of configs 1, 3-6, whose "
                         "`batch` already is the per-GPU shard of an 8-GPU configuration).  The other regime is reported beside "
                         "it in `config.other_scaling`")
    ap.add_argument("--batch", type=int, default=0,
                    help="per-GPU batch override (0 = the configuration's): --config 2 --batch 8192 is the strong-scaling shard of "
                         "the 65 536-row batch at 8 GPUs, runnable on one")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-shard", action="store_true", help="skip `config.strong_shard` (the 8 192-row shard of config 2 on this GPU)")
    ap.add_argument("--no-roofline", action="store_true", help="profiling runs: skip the stand-alone kernel timing leg")
    ap.add_argument("--single-batch", action="store_true", help="replay ONE batch (cache-hot rows, no lazy-Adam lag): round-1 behaviour")
    ap.add_argument("--uniform-ids", action="store_true", help="uniform ids for the large tables (worst case for the gather)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.scaling is None:
        args.scaling = "strong" if (args.config == 2 and args.gpus > 1 and not args.batch and cfg["batch"] % args.gpus == 0) else "weak"
    if args.batch:
        cfg = dict(cfg, batch=args.batch, name=f"{cfg['name']}_shard{args.batch}")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start one process per GPU ourselves (the driver's form: torch.distributed.run, one rank per GPU over
        # RCCL) and hand its exit code on -- `--gpus 8` must never quietly measure ONE rank and print n_gpus 1
        import socket
        import subprocess
        backend = os.environ.get("SWR_BENCH_BACKEND", "nccl")
        if backend == "nccl" and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (SWR_BENCH_BACKEND=gloo runs the "
                             f"{args.gpus}-rank code path with the ranks sharing the visible GPUs)")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    # stdout carries ONE JSON line and nothing else: RCCL prints a five-line banner (version, hostname, library path) through C
    # stdio when a communicator is created, and the buffer is flushed at exit -- BEHIND the line.  Everything but the line goes to
    # stderr: fd 1 is pointed at fd 2 for the run and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would claim the wrong number of GPUs")
    # SWR_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a one-GPU box (ranks share cuda:0); the driver's
    # multi-GPU runs use the default: backend nccl (= RCCL), one GPU per rank
    backend = os.environ.get("SWR_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # SWR_BENCH_FORCE_DP=1: take the N > 1 code path (process group, exchange step, three graphs) with world size 1 --
    # the RCCL call sequence of the multi-GPU run, checkable on a one-GPU box
    use_dp = world > 1 or bool(os.environ.get("SWR_BENCH_FORCE_DP"))
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    if cfg.get("on_device_init"):          # 100 M-row tables: initialise in HBM, not through 26 GB of host memory
        with torch.device(dev):
            model, feats = build_model(cfg)
    else:
        model, feats = build_model(cfg)
    if os.environ.get("SWR_BENCH_FREEZE_TABLES"):          # debugging aid: no embedding backward / table update
        for n_, p_ in model.named_parameters():
            if "embed_dict" in n_:
                p_.requires_grad_(False)
    trainer = CTRTrainer(model, cfg["name"], optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device=str(dev))
    trainer.use_graph = False              # bench.py drives capture / replay itself
    model.train()
    B = cfg["batch"]
    if args.scaling == "strong":
        if B % world:
            raise SystemExit(f"strong scaling: batch {B} does not split over {world} GPUs")
        B //= world                        # rows [rank * B, (rank + 1) * B) of the global batch
    n_rot = 1 if args.single_batch else N_ROTATE
    batches = []
    for j in range(n_rot):
        xh, yh = synth_batch(cfg, B, seed=2022 + args.config + 1000 * rank + 77 * j, zipf=not args.uniform_ids)
        batches.append(({k: torch.from_numpy(v).to(dev) for k, v in xh.items()}, torch.from_numpy(yh).to(dev)))
    x, y = batches[0]

    stepper = None
    if use_dp:
        from scenario_wise_rec.parallel import DataParallelStep
        stepper = DataParallelStep(trainer, world)
        step_fn = stepper.train_step
    else:
        step_fn = trainer.train_step

    # ---- warm-up (eager), then capture the step into hipGraph(s) ------------------------------------------
    # 1 GPU: the whole step is ONE graph.  N GPUs: three graphs with the two RCCL all-gathers and the row merge issued
    # eagerly in between (parallel.DataParallelStep).
    graph = None
    if not args.no_graph:
        try:
            if not use_dp:
                from scenario_wise_rec.trainers.graph import GraphedStep
                graph = GraphedStep(trainer, x, y, warmup=args.warmup)
            else:
                graph = stepper.capture(x, y, warmup=args.warmup)
            _stage("captured")
            graph.replay()
            torch.cuda.synchronize()
            _stage("first replay")
        except Exception as e:                     # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    if graph is None:
        for _ in range(max(1, args.warmup)):
            step_fn(x, y)
        torch.cuda.synchronize()
    H.check_errors()

    def run(i):
        """One step on batch i % n_rot: the batch is copied into the captured input buffers (ONE launch for all columns,
        trainers/graph.py load_batch) and the graph replayed -- the per-step work of the drop-in loop."""
        xb, yb = batches[i % n_rot]
        if graph is None:
            return step_fn(xb, yb)
        if n_rot > 1:
            graph.load(xb, yb)
        return graph.replay()

    def timed(n_steps, first, run=run):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            loss = run(first + i)
        t_issue = time.perf_counter() - t0         # host side only: how far ahead of the device the launch loop runs
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        print(f"[bench] {n_steps} steps issued in {t_issue * 1e3:.2f} ms, finished in {dt * 1e3:.2f} ms", file=sys.stderr)
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, loss

    for i in range(2 * n_rot):                 # every batch through the captured step once before timing
        run(i)
    _stage("pre-timed replays")

    # ---- timed region: exactly K steps between barrier + synchronize ------------------------------------
    dt, loss = timed(args.steps, 0)
    H.check_errors()
    if os.environ.get("SWR_STAMPS"):
        from scenario_wise_rec import ops as _ops
        print("[bench] stamps (us): " + "  ".join(f"{n}={t:.1f}" for n, t in _ops.read_stamps().items()), file=sys.stderr)
    final_loss = float(loss.detach())
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    # beside it (NOT the reported value): the same K steps replaying ONE batch -- every row it touches is cache-hot and
    # the lazy Adam never has a row lagging (what round 1 reported)
    single_ms = None
    if graph is not None and n_rot > 1:
        n_rot_saved, n_rot = n_rot, 1
        run(0)
        dts, _ = timed(args.steps, 0)
        single_ms = dts / args.steps * 1e3
        n_rot = n_rot_saved

    # ---- N > 1: the OTHER scaling regime beside the reported one, in the same line (`config.other_scaling`): weak = the
    # configuration's batch per GPU, strong = that batch split by row over the GPUs (65 536 -> 8 192 per GPU at 8, where
    # the ~40 launches of a step are a latency floor).  A second capture of the step at the other batch shape.
    other = None
    if world > 1 and graph is not None and (args.scaling == "strong" or cfg["batch"] % world == 0):
        B2 = cfg["batch"] if args.scaling == "strong" else cfg["batch"] // world
        b2 = []
        for j in range(n_rot):
            xh, yh = synth_batch(cfg, B2, seed=5022 + args.config + 1000 * rank + 77 * j, zipf=not args.uniform_ids)
            b2.append(({k: torch.from_numpy(v).to(dev) for k, v in xh.items()}, torch.from_numpy(yh).to(dev)))
        if use_dp:
            g2 = stepper.capture(b2[0][0], b2[0][1], warmup=max(2, args.warmup))
        else:
            from scenario_wise_rec.trainers.graph import GraphedStep
            g2 = GraphedStep(trainer, b2[0][0], b2[0][1], warmup=max(2, args.warmup))

        def run2(i):
            if n_rot > 1:
                g2.load(*b2[i % n_rot])
            return g2.replay()
        for i in range(2 * n_rot):
            run2(i)
        dt2, _ = timed(args.steps, 0, run2)
        H.check_errors()
        other = {"scaling": "weak" if args.scaling == "strong" else "strong", "global_batch": world * B2, "per_gpu_batch": B2,
                 "ms_per_step": dt2 / args.steps * 1e3, "value": world * B2 * args.steps / dt2, "unit": "samples/s"}

    # ---- what the process group really was: every rank's device as the runtime names it, gathered over the group itself
    # (N > 1: the driver can see from the line that RCCL connected N ranks on N different GPUs)
    ranks_info = None
    if dist is not None:
        prop = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": prop.name,
                "pci_bus_id": getattr(prop, "pci_bus_id", None), "uuid": str(getattr(prop, "uuid", "")),
                "visible_devices": torch.cuda.device_count()}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": gathered,
                      "distinct_devices": len({(g["pci_bus_id"], g["uuid"], g["device_index"]) for g in gathered})}

    # which collectives carried the step's exchange (N > 1; parallel.DataParallelStep): the gradient arena by ONE all-reduce above
    # SWR_DP_ALLREDUCE_BYTES (1 MiB), else by one all-gather + a rank-ordered local sum (the 0.5 MB arena of config 2: one
    # latency-bound collective either way); the large tables' row lists always by one all-gather
    dp_exchange = None
    if dist is not None and stepper is not None:
        from scenario_wise_rec.parallel import allreduce_min_bytes
        xb = getattr(stepper, "_xb", None)
        arena_bytes = 4 * int(xb["A"]) if xb else None
        dp_exchange = {"arena_collective": (("all_reduce" if xb["allreduce"] else "all_gather + rank-ordered local sum") if xb else
                                            "all_reduce above / all_gather + local sum below the threshold"),
                       "allreduce_min_bytes": allreduce_min_bytes(), "arena_bytes": arena_bytes,
                       "row_lists": "all_gather of (row id, gradient) per large table, merged by swr_dp_finish", "backend": dist.get_backend()}
    if rank != 0:
        return

    bf16_mode = H.lib.swr_gemm_precision_mode() == 2
    # ---- roofline of the dominant kernel (see DESIGN.md "Measurement") ------------------------------------
    print(f"[bench] timed region done: {ms:.3f} ms/step", file=sys.stderr, flush=True)
    roof = None
    if not args.no_roofline:
        if args.config == 2:
            roof = measure_roofline(cfg, model, trainer, x, dev, args.steps, B)
        else:
            # the other configurations: primary = the widest product of THEIR step (the kernel class that bounds it), timed
            # stand-alone at the step's shapes; the lookup beside it
            roof = product_roofline(cfg, dev, args.steps, B, args.config, ms)
            roof.setdefault("also", {})["gather"] = gather_roofline(cfg, model, x, dev, args.steps, B, args.config)
        if roof is not None:
            roof.setdefault("also", {})["step"] = step_roofline(ms, args.config)
    shard = None
    if world == 1 and not use_dp and args.config == 2 and not args.batch and graph is not None and not args.no_strong_shard:
        try:
            shard = strong_shard_block(cfg, args, dev, ms)
        except Exception as e:                     # noqa: BLE001  (the block is a report beside the line, never the line)
            shard = {"error": f"{type(e).__name__}: {e}"}
    out = {
        "metric": "train samples/sec at batch 65 536, KuaiRand 5-domain MMoE, 1/2/4/8 MI355X" if args.config == 2
                  else "train samples/sec, " + cfg["name"],
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "bf16" if bf16_mode else "f32", "data": "synthetic",
        "config": {"workload": cfg["name"], "global_batch": world * B, "per_gpu_batch": B,
                   "step": "fwd+BCE+bwd+Adam(all params; dense-Adam semantics on every table row, applied lazily but exactly)", "parallelism": f"dp{world}",
                   "ids": "uniform" if args.uniform_ids else "zipf1.05(large tables)+uniform", "hipgraph": graph is not None,
                   "batches_rotated": n_rot, "ms_per_step_single_batch_replayed": single_ms,
                   "final_loss": final_loss, "scaling": args.scaling, "other_scaling": other, "strong_shard": shard, "process_group": ranks_info,
                   "dp_exchange": dp_exchange,
                   "precision_mode": ("bf16 perf mode (SWR_GEMM=bf16): ONE bf16 MFMA product per k-group, operands rounded to bf16 -- "
                                      "NOT the parity path (max logit error ~1e-3..1e-2 at these widths, tests/test_perf_mode_gpu.py); "
                                      "reported beside the fp32-accurate line, never instead of it") if bf16_mode else
                                     "fp32-accurate: every fp32 product as six bf16 MFMA products, fp32 accumulate (parity path)"},
        "roofline": roof,
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(cfg)
    sys.stdout.flush()
    os.write(out_fd, (json.dumps(out) + "\n").encode())


def strong_shard_block(cfg, args, dev, ms_full, ways=8):
    """`config.strong_shard` of the N = 1 line: the step of ONE rank of the 8-GPU strong-scaling run -- the metric's 65 536-row
    batch sharded by row, 8 192 rows per GPU (`north_star`; reference `ctr_trainer.py:45-47`: DataParallel scatters one batch) --
    timed on this GPU, (a) as the single-GPU graphed step and (b) through the N > 1 code path at world size 1 over RCCL (process
    group, row-list all-gather, arena collective, merge: every launch and collective call of the 8-GPU step, its wire time not
    included).  `projected_speedup_1_to_8` = this line's ms_per_step / (b): what 8 GPUs would reach if the collectives hid
    completely -- an upper bound, reported because no 8-GPU node has been available to this build; the driver's own N = 8 run
    supersedes it."""
    import socket
    import torch.distributed as dist
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.trainers.graph import GraphedStep
    B8 = cfg["batch"] // ways
    steps = max(args.steps, 100)

    def fresh():
        model, _ = build_model(cfg)
        tr = CTRTrainer(model, cfg["name"] + f"_shard{B8}", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device=str(dev))
        tr.use_graph = False
        model.train()
        return tr

    batches = []
    for j in range(N_ROTATE):
        xh, yh = synth_batch(cfg, B8, seed=9022 + 77 * j, zipf=not args.uniform_ids)
        batches.append(({k: torch.from_numpy(v).to(dev) for k, v in xh.items()}, torch.from_numpy(yh).to(dev)))

    def time_graph(g):
        for i in range(2 * N_ROTATE):
            g.load(*batches[i % N_ROTATE])
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g.load(*batches[i % N_ROTATE])
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    tr = fresh()
    g = GraphedStep(tr, batches[0][0], batches[0][1], warmup=max(2, args.warmup))
    ms_graph = time_graph(g)
    H.check_errors()
    out = {"global_batch": cfg["batch"], "ways": ways, "per_gpu_batch": B8, "steps": steps, "ms_per_step_graphed": ms_graph,
           "samples_per_s_per_gpu": B8 / ms_graph * 1e3}
    del g, tr
    try:
        if not dist.is_initialized():
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from scenario_wise_rec.parallel import DataParallelStep
        tr = fresh()
        st = DataParallelStep(tr, 1)
        g = st.capture(batches[0][0], batches[0][1], warmup=max(2, args.warmup))
        ms_dp = time_graph(g)
        H.check_errors()
        out.update(ms_per_step_world1_rccl=ms_dp, world1_over_graphed=ms_dp / ms_graph,
                   one_graph=bool(g._graphs[1] is None),
                   projected_speedup_1_to_8=ms_full / ms_dp,
                   projection_note="ms_per_step of this line (65 536 rows on one GPU) / the world-1 RCCL step of the 8 192-row shard: "
                                   "the 1 -> 8 strong-scaling speed-up if the two collectives cost nothing on the wire (upper bound)")
        del g, st, tr
        dist.destroy_process_group()
    except Exception as e:                         # noqa: BLE001
        out["world1_rccl_error"] = f"{type(e).__name__}: {e}"
    return out


def measure_roofline(cfg, model, trainer, x, dev, iters, B):
    """The launches of the stacked expert + gate layer ([B, 516] x [516, 148] algorithmically) and of the lookup in front of
    it, each timed stand-alone at the shapes the step really launches.

    The step runs the FUSED lookup + first layer (csrc/first_layer.hip): the lookup writes row keys,
    one-hot bits and piece offsets only; the forward product Z = A' Wf^T and the weight gradient dWp = dZ^T A' fetch table
    rows through the keys (A' = [9 tables x 16 | 4 dense | 12 zero | 128 one-hot] = 288 columns is never written); dX is
    computed only for the 144 columns of the tables that go through K3.  `achieved` / `frac` use SURVEY.md 8(d)'s
    ALGORITHMIC flops of the layer, 2 * B * N * K with K = 516 -- what the reference's Linear computes -- against the dense
    bf16 MFMA peak; `executed_flops_per_launch` is what the launch multiplies after the folding.  Every fp32 product is six
    bf16 MFMA products (3-way operand split, fp32 accumulate: the 1e-4 logit bar rules plain bf16 out; three for the
    bf16-exact one-hot columns), so the matrix pipes issue up to 6 x the executed flops: `mfma_issue_util`.  `hbm_frac`
    prices the launch's executed bytes against the 8 TB/s HBM peak.
    Primary entry = the longest kernel of the step, the weight-gradient product (`gemm_tn_x6w_kernel` + its fixed-order
    partial-tile reduction).  Timed live with HIP events on the launch stream over graph-captured launches; four
    operand sets are rotated.  With SWR_FUSED_LOOKUP=0 (or SWR_GEMM != default) the written-block launches are timed
    instead (`gemm_tn_x6_kernel`, `gemm_rows_x6_kernel`, `embed_gather_kernel`)."""
    import ctypes as C
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    from scenario_wise_rec._hip import lib
    fs, e, fd = len(cfg["vocabs"]), cfg["embed_dim"], cfg["n_dense"]
    k0 = fs * e + fd
    hyp = cfg["hyper"]
    n1 = hyp["n_expert"] * hyp["expert_params"]["dims"][0] + hyp["domain_num"] * hyp["n_expert"]
    g = torch.Generator(device=dev).manual_seed(1)
    # A' exactly as the step's lookup lays it out (real keys / one-hot bits, real embeddings of the other tables)
    was_training = model.training
    model.train()
    sets, info = [], None
    for j in range(4):
        xb = x if j == 0 else {k: torch.from_numpy(v).to(dev) for k, v in synth_batch(cfg, B, seed=700 + j)[0].items()}
        with torch.no_grad():
            out = model.embedding(xb, model.features, squeeze_dim=True, onehot="layout")
        info = getattr(out, "_swr_onehot", None)
        if info is None or not info.fold:
            A = out.detach()
            kf = k0
        else:
            kf = info.Kp + info.oh_width
            A = None if info.fl is not None else info.wide[:, info.col0:info.col0 + kf].detach()
        sets.append((torch.randn(B, n1, device=dev, generator=g), A, info))
    model.train(was_training)
    folded = info is not None and info.fold
    fused = folded and info.fl is not None
    n_sel = info.n_sel if folded else k0
    Wf = torch.randn(n1, kf, device=dev, generator=g) * 0.05
    Wsel = torch.randn(n_sel, n1, device=dev, generator=g) * 0.05
    bias = torch.zeros(n1, device=dev)
    dW = torch.empty(n1, kf, device=dev)
    db = torch.empty(n1, device=dev)
    Z = torch.empty(B, n1, device=dev)
    dX = torch.empty(B, (n_sel + 3) // 4 * 4, device=dev)
    parts = torch.empty(((B + 31) // 32, n1, 2), device=dev)
    stream = torch.cuda.Stream()
    st = {"i": 0}

    def nxt():
        st["i"] += 1
        return sets[st["i"] % 4]

    if fused:
        # the layer's own parameters through the parameter-sized launch (folded weights in fragment order + the small tables'
        # bf16-term shadows), once per operand set; then the two products as the step launches them
        W = ops._cat_params([m.block(0)[0].weight for m in model.experts] + [gt.block(0)[0].weight for gt in model.gates])
        tabs = (H.OnehotTable * len(info.tables_p))()
        for j, (p_t, vocab, dim, off, col) in enumerate(info.tables_p):
            tabs[j] = H.OnehotTable(p_t.data_ptr(), vocab, dim, off, col)
        for _dz, _a, inf in sets:
            inf.fl["plan"].N = n1
            H.check(lib.swr_fl_prep(C.byref(inf.fl["plan"]), H.ptr(W), W.stride(0), k0, H.ptr(inf.ohtab), tabs, len(inf.tables_p), None, 0,
                                    None, n1, H.ptr(inf.fl["ws"]), H.stream()), "swr_fl_prep")
        nb = lib.swr_fl_dw_workspace_bytes(C.byref(info.fl["plan"]))
        wsd = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        # as the step launches it: where the wide kernel takes the product, dZ is recomputed from dY and Z while it is staged
        # (swr_fl_dw_bn: rows padded to 128 bytes), else read (swr_fl_dw)
        n1p = (n1 + 31) // 32 * 32
        bn_form = bool(ops.DZ_FREE and lib.swr_fl_dw_bn_supported(C.byref(info.fl["plan"]), n1p, n1p))
        if bn_form:
            dYs = [torch.nn.functional.pad(dz, (0, n1p - n1)).contiguous() for dz, _a, _i in sets]
            Zs = [torch.randn(B, n1p, device=dev, generator=g) for _ in sets]
            cfs = [torch.randn(n1, device=dev, generator=g) * 0.1 for _ in range(4)]

        def f_tn():
            dZ, _a, inf = nxt()
            if bn_form:
                j = st["i"] % 4
                H.check(lib.swr_fl_dw_bn(C.byref(inf.fl["plan"]), H.ptr(inf.fl["ws"]), H.ptr(dYs[j]), n1p, H.ptr(Zs[j]), n1p, H.ptr(cfs[0]),
                                         H.ptr(cfs[1]), H.ptr(cfs[2]), H.ptr(cfs[3]), H.ptr(dW), kf, H.ptr(db), H.ptr(wsd), nb, H.stream()),
                        "swr_fl_dw_bn")
                return
            H.check(lib.swr_fl_dw(C.byref(inf.fl["plan"]), H.ptr(inf.fl["ws"]), H.ptr(dZ), n1, H.ptr(dW), kf, H.ptr(db), H.ptr(wsd), nb,
                                  H.stream()), "swr_fl_dw")

        def f_fwd():
            _dz, _a, inf = nxt()
            H.check(lib.swr_fl_fwd(C.byref(inf.fl["plan"]), H.ptr(inf.fl["ws"]), H.ptr(bias), H.ptr(Z), n1, H.ptr(parts), H.stream()),
                    "swr_fl_fwd")
    else:
        def f_tn():
            dZ, A, _i = nxt()
            ops.gemm_tn(dZ, A, dW, B, n1, kf, colsum=db)

        def f_fwd():
            _dZ, A, _i = nxt()
            ops.gemm("nt", A, Wf, Z, B, n1, kf, bias=bias, stat_partials=parts, a_exact_from=ex_from)

    # the one-hot columns of A' are bf16-exact: 3 products instead of 6 (fused: every one-hot group / column tile; written
    # block: whole pairs of 32-column chunks behind `a_exact_from`, include/swr.h)
    ex_from = info.Kp if folded else 0
    k_half = (kf - ex_from if fused else max(0, kf - (ex_from + 63) // 64 * 64)) if ex_from > 0 else 0

    # dX: the step launches fl_dx_kernel (swr_bn_bwd_dx: BatchNorm backward in the A fragment + dX = dZ W[:, sel], rows padded to
    # 128 bytes, dZ not written where the weight gradient recomputes it) when the fused forms are on; else the plain product
    dx_fused = bool(fused and ops.FUSE_BN_DX and lib.swr_bn_bwd_dx_supported(n1, n_sel))
    if dx_fused:
        n1p_ = (n1 + 31) // 32 * 32
        nsp = (n_sel + 31) // 32 * 32
        dYx = [torch.nn.functional.pad(dz, (0, n1p_ - n1)).contiguous() for dz, _a, _i in sets]
        Zx = [torch.randn(B, n1p_, device=dev, generator=g) for _ in sets]
        cfx = [torch.randn(n1, device=dev, generator=g) * 0.1 for _ in range(4)]
        dXp = torch.empty(B, nsp, device=dev)
        dZo = None if (ops.DZ_FREE and bn_form) else torch.empty(B, n1p_, device=dev)
        Wt_sel = torch.empty((n_sel, n1), device=dev)
        for _dz, _a, inf in sets:                  # the B3X image of the selected rows of W^T, once per operand set
            H.check(lib.swr_fl_prep(C.byref(inf.fl["plan"]), H.ptr(W), W.stride(0), k0, H.ptr(inf.ohtab), tabs, len(inf.tables_p), H.ptr(inf.sel),
                                    inf.n_sel, H.ptr(Wt_sel), n1, H.ptr(inf.fl["ws"]), H.stream()), "swr_fl_prep")
        torch.cuda.synchronize()

    def f_dx():
        dZ, _A, inf = nxt()
        if dx_fused:
            j = st["i"] % 4
            H.check(lib.swr_bn_bwd_dx(C.byref(inf.fl["plan"]), H.ptr(inf.fl["ws"]), H.ptr(dYx[j]), n1p_, H.ptr(Zx[j]), n1p_, H.ptr(cfx[0]),
                                      H.ptr(cfx[1]), H.ptr(cfx[2]), H.ptr(cfx[3]), n_sel, H.ptr(dZo), n1p_, H.ptr(dXp), nsp, H.stream()),
                    "swr_bn_bwd_dx")
            return
        ops.gemm("nt", dZ, Wsel, dX, B, n_sel, n1)

    x6 = os.environ.get("SWR_GEMM", "")[:1].lower() != "f"
    peak = BF16_MFMA_PEAK_TFLOPS if x6 else F32_MFMA_PEAK_TFLOPS
    alg_flops = 2.0 * B * n1 * k0

    def entry(kname, fn, pmc_name, k_exec, alg, k3=0, nbytes=None):
        """k3: columns of the reduction range that take 3 bf16 products instead of 6."""
        ms = time_kernel_events(fn, max(10, iters), stream)
        exe = 2.0 * B * n1 * k_exec
        nbytes = 4.0 * B * (n1 + k_exec) if nbytes is None else nbytes
        tf = alg / (ms * 1e-3) / 1e12
        return {"kernel": kname, "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "algorithmic_flops_per_launch": alg, "executed_flops_per_launch": exe, "executed_bytes_per_launch": nbytes,
                "avg_launch_ms": ms,
                "mfma_issue_util": ((6.0 - 3.0 * k3 / k_exec) if x6 else 1.0) * exe / (ms * 1e-3) / 1e12 / peak,
                "mfma_dtype": "bf16 x 6 products per fp32 product (3-way operand split, fp32 accumulate; 3 for one-hot columns)" if x6 else "f32",
                "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(pmc_name),
                "shape": f"[{B}, {k_exec}] x [{k_exec}, {n1}]" + (" (folded from K = %d)" % k0 if folded and k_exec != n_sel else "")}

    if fused:
        # bytes the fused launches move per sample: forward = 32 B of piece offsets per real group half ... in total the
        # offsets, the mask, 48 B per (sample, piece) of bf16 terms (mostly L2 hits on the shadows) and the Z row; the weight
        # gradient = the dZ row (read once per 128-column block), keys / mask words, 4 B per real column of A'
        nr = info.Kp // 16
        fwd_bytes = float(B) * (nr * 8 + 16 + 2 * nr * 48 + 4 * n1)
        # the wide kernel (one workgroup stages dZ and every column of A') where its shape is instantiated and SWR_TN_WIDE != 0,
        # else the blocked one (dZ read once per 128-column block of A')
        wide = os.environ.get("SWR_TN_WIDE", "1") != "0" and (n1 + 31) // 32 == 5 and (kf + 31) // 32 == 9
        n_qblk = 1 if wide else max(1, kf // 128)
        # the transpose-read form (csrc/dw_tr.hip, swr_dw_tr_mode 1) where dZ is recomputed and the shape is instantiated
        tr = bool(bn_form and wide and lib.swr_dw_tr_mode(-1) == 1)
        if tr:
            # per sample: the dY and Z rows, 48 B of bf16 terms + a 4-byte offset per 8-column piece (mostly L2 hits on the table
            # shadows), 16 B of one-hot words; per launch: one partial tile per batch split written, then read by the reduction
            n_splits = min(256, max(1, B // 128))
            dw_bytes = float(B) * (8 * n1 + (48 + 4) * 2 * nr + 16) + 2.0 * n_splits * n1 * kf * 4
            dw_name, dw_label = "dw_tr_kernel", ("dw_tr_kernel (+tn_reduce4): dWp = dZ^T A', dZ recomputed from dY and Z, operands transposed by "
                                                 "ds_read_b64_tr_b16, A' from the pre-split pieces by LDS-DMA")
        else:
            dw_bytes = float(B) * (n_qblk * 4 * n1 * (2 if (wide and ops.DZ_FREE) else 1) + 4 * info.Kp + 4 * (info.Kp // 16) + 16)
            dw_name = "gemm_tn_x6w_kernel" if wide else "gemm_tn_x6g_kernel"
            dw_label = dw_name + " (+tn_reduce): dWp = dZ^T A', A' gathered through the row keys"
        roof = entry(dw_label, f_tn, "void %s<" % dw_name, kf, alg_flops, k3=k_half, nbytes=dw_bytes)
        fwd_ent = entry("fl_fwd_kernel (forward Z = A' Wf^T, lookup fused as the A-operand producer, BN partials in the epilogue)", f_fwd,
                        "void fl_fwd_kernel<", kf, alg_flops, k3=k_half, nbytes=fwd_bytes)
    else:
        kname = "gemm_tn_x6_kernel" if x6 else "gemm_tn_kernel"
        roof = entry(kname + " (+tn_reduce_kernel)", f_tn, "void %s<" % kname, kf, alg_flops)
        fwd_ent = entry("gemm_rows_x6_kernel (forward, BN partials in the epilogue)", f_fwd, "void gemm_rows_x6_kernel<5", kf, alg_flops,
                        k3=k_half)
    roof["traffic_unit"] = "HBM bytes per launch, rocprofv3 PMC (profiles/pmc_hbm_latest.json; null if not collected)"
    roof["traffic_source"] = pmc_source(2)
    roof["also"] = {
        ("fl_fwd_kernel(forward)" if fused else "gemm_rows_x6_kernel(forward)"): fwd_ent,
        # dX is algorithmically [B, 148] x [148, 512]; the small tables' columns are never computed (their gradients
        # come out of the weight-gradient product's one-hot block)
        ("fl_dx_kernel(dX)" if dx_fused else "gemm_rows_x6_kernel(dX)"): (
            entry("fl_dx_kernel (BatchNorm backward in the A fragment + dX = dZ W[:, sel], columns of the K3 tables only; dZ "
                  + ("not written" if dZo is None else "written") + ")", f_dx, "void fl_dx_kernel<", n_sel, 2.0 * B * n1 * fs * e,
                  nbytes=float(B) * (8 * n1 + 4 * n_sel + (0 if dZo is None else 4 * n1))) if dx_fused else
            entry("gemm_rows_x6_kernel (dX = dZ W, columns of the K3 tables only)", f_dx, "void gemm_rows_x6_kernel<5", n_sel,
                  2.0 * B * n1 * fs * e)),
        ("fl_keys_kernel" if fused else "embed_gather_kernel"): gather_roofline(cfg, model, x, dev, iters, B),
        "k3_direct_sums": k3_roofline(cfg, dev, iters, B),
    }
    # the north star's two named fractions, at the top level of `roofline`:
    #  gather_hbm_frac  = executed bytes of the lookup -- the keys launch AND the table-row fetches that moved into the forward
    #                     product -- over the two launches' time, against the 8 TB/s peak;
    #  expert_mfma_busy = matrix-pipe busy fraction of the three first-layer products from the committed PMC pass
    #                     (SQ_VALU_MFMA_BUSY_CYCLES / chip cycles, tools/pmc_mfma_summary.py)
    keys_ent = roof["also"].get("fl_keys_kernel") if fused else None
    if keys_ent is not None:
        both_b = keys_ent["executed_bytes_per_launch"] + fwd_ent["executed_bytes_per_launch"]
        both_ms = keys_ent["avg_launch_ms"] + fwd_ent["avg_launch_ms"]
        roof["gather_hbm_frac"] = both_b / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["gather_hbm_frac_of"] = "fl_keys_kernel + fl_fwd_kernel: executed bytes of both / time of both / 8 TB/s"
    roof["expert_mfma_busy"] = mfma_busy_source()
    return roof


def widest_products(cfg):
    """The products of the widest layer of a configuration's step -- (label, K, N, launches per step) with N the stacked output
    width the step multiplies in ONE launch (all domains / experts side by side) -- and the minimal forward flops per sample of
    the whole dense part (SURVEY.md 8d's count: 2 K N per Linear, train step ~ 3 x forward)."""
    e, fs, fd, h = cfg["embed_dim"], len(cfg["vocabs"]), cfg["n_dense"], cfg["hyper"]
    k0 = fs * e + fd
    fam = cfg["family"]
    if fam == "SharedBottom":
        return "bottom layer %d -> %d" % (k0, h["bottom_params"]["dims"][0]), k0, h["bottom_params"]["dims"][0]
    if fam == "Star":
        return "first FCN layer of the %d domains, %d -> %d each" % (h["num_domains"], k0, h["fcn_dims"][0]), k0, h["num_domains"] * h["fcn_dims"][0]
    if fam == "PLE":
        ne = h["domain_num"] * h["n_expert_specific"] + h["n_expert_shared"]
        return "first layer of the %d experts, %d -> %d each" % (ne, k0, h["expert_params"]["dims"][0]), k0, ne * h["expert_params"]["dims"][0]
    if fam in ("HamurSmall", "HamurLarge"):
        return "first backbone layer of the %d domains, %d -> %d each" % (h["domain_num"], k0, h["fcn_dims"][0]), k0, h["domain_num"] * h["fcn_dims"][0]
    if fam == "PPNet":
        return "first tower layer of the %d domains, %d -> %d each" % (h["domain_num"], k0, h["fcn_dims"][0]), k0, h["domain_num"] * h["fcn_dims"][0]
    return "first layer", k0, 128


def product_roofline(cfg, dev, iters, B, config, ms_per_step):
    """Roofline primary of configurations 1, 3-6: the forward product of the step's widest layer, [B, K] x [K, N] with the
    BatchNorm partials in its epilogue, launched through ops.gemm exactly as basic/layers.LayerBank launches it (bf16-split
    kernel gemm_rows_x6_kernel: six bf16 MFMA products per fp32 product), timed stand-alone with HIP events over graph-captured
    launches, four rotating operand sets.  `achieved` = algorithmic flops 2 B K N / time against the dense bf16 MFMA peak."""
    from scenario_wise_rec import ops
    label, K, N = widest_products(cfg)
    g = torch.Generator(device=dev).manual_seed(5)
    Kp = (K + 3) // 4 * 4
    sets = [(torch.randn(B, Kp, device=dev, generator=g), torch.empty(B, N, device=dev)) for _ in range(4)]
    W = torch.randn(N, Kp, device=dev, generator=g) * 0.05
    bias = torch.zeros(N, device=dev)
    parts = torch.empty(((B + 31) // 32, N, 2), device=dev)
    st = {"i": 0}

    def launch():
        A, Z = sets[st["i"] % 4]
        st["i"] += 1
        ops.gemm("nt", A, W, Z, B, N, K, bias=bias, stat_partials=parts, lda=Kp, ldb=Kp)
    stream = torch.cuda.Stream()
    ms = time_kernel_events(launch, max(10, iters), stream)
    x6 = os.environ.get("SWR_GEMM", "")[:1].lower() != "f"
    peak = BF16_MFMA_PEAK_TFLOPS if x6 else F32_MFMA_PEAK_TFLOPS
    alg = 2.0 * B * K * N
    tf = alg / (ms * 1e-3) / 1e12
    tiles = (N + 31) // 32
    nblk = (tiles + 6) // 7
    nt = (tiles + nblk - 1) // nblk
    nbytes = 4.0 * B * (nblk * K + N)              # A is read once per column block, Z written once
    return {"kernel": "gemm_rows_x6_kernel<%d> (forward product of the widest layer: %s)" % (nt, label), "bound": "mfma",
            "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "algorithmic_flops_per_launch": alg,
            "executed_bytes_per_launch": nbytes, "avg_launch_ms": ms, "shape": f"[{B}, {K}] x [{K}, {N}]",
            "mfma_issue_util": (6.0 if x6 else 1.0) * tf / peak, "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "share_of_step": "this launch and its two backward counterparts (dX, dW: the same flops each) ~ %.0f %% of the step"
                             % (100.0 * 3 * ms / ms_per_step),
            "traffic": pmc_traffic("void gemm_rows_x6_kernel<%d" % nt, config), "traffic_source": pmc_source(config),
            "traffic_unit": "HBM bytes per launch, rocprofv3 PMC (null if not collected)"}


def _pmc_file(config):
    """The committed PMC summary of THIS configuration's bench command (tools/prof_round2.sh / tools/prof_configs_all.sh ->
    tools/pmc_to_json.py): config 2 -> profiles/pmc_hbm_latest.json, config N -> profiles/pmc_hbm_cfgN.json; None if absent
    (counters cannot be collected from inside the bench process)."""
    path = os.path.join(ROOT, "profiles", "pmc_hbm_latest.json" if config == 2 else f"pmc_hbm_cfg{config}.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"]
    except (OSError, KeyError, ValueError):
        return None


def pmc_source(config):
    """Where the `traffic` fields of this line come from: PMC counters cannot be collected from inside the bench process, so they
    are read from the committed summary of the same command under rocprofv3 (tools/prof_round2.sh / prof_configs_all.sh)."""
    name = "pmc_hbm_latest.json" if config == 2 else f"pmc_hbm_cfg{config}.json"
    path = os.path.join(ROOT, "profiles", name)
    src = {"file": "profiles/" + name, "measured_in_this_run": False, "commit": None, "collected": None}
    try:
        with open(path) as f:
            src["collected"] = json.load(f).get("collected")
    except (OSError, ValueError):
        return None
    try:
        import subprocess
        r = subprocess.run(["git", "log", "-1", "--format=%h %cI", "--", path], capture_output=True, text=True, cwd=ROOT, timeout=10)
        src["commit"] = r.stdout.strip() or "no git history on this box: `git log -- profiles/%s` in the repository" % name
    except Exception:
        src["commit"] = "no git on this box: `git log -- profiles/%s` in the repository" % name
    return src


def mfma_busy_source():
    """Matrix-pipe busy fractions of the first layer's three products from profiles/pmc_mfma_latest.json (written by
    tools/pmc_mfma_summary.py --json from two rocprofv3 PMC passes over this bench command); None if absent."""
    path = os.path.join(ROOT, "profiles", "pmc_mfma_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    out = {"source": {"file": "profiles/pmc_mfma_latest.json", "measured_in_this_run": False, "collected": d.get("collected")}}
    for k, v in d.get("kernels", {}).items():
        if any(t in k for t in ("fl_fwd_kernel", "fl_dx_kernel", "dw_tr_kernel", "gemm_tn_x6w_kernel")):
            out[k.replace("void ", "")] = v.get("mfma_busy")
    return out


def pmc_traffic(kernel, config=2):
    """HBM bytes per launch of `kernel` in configuration `config`'s step, or None."""
    kernels = _pmc_file(config)
    if not kernels:
        return None
    for name, ent in kernels.items():          # template arguments may follow (`<5, true>`)
        if name.startswith(kernel):
            return ent["hbm_bytes_per_launch"]
    return None


def step_roofline(ms_per_step, config=2):
    """The whole step against the HBM roofline: sum of the PMC bytes of every kernel of one step / ms_per_step / 8 TB/s.
    Steps in the PMC run = launches of a kernel that runs exactly once per step."""
    kernels = _pmc_file(config)
    if not kernels:
        return None
    once = [v["launches"] for k, v in kernels.items() if k.startswith(("adam_advance_kernel", "adam_dense"))]
    steps = min(once) if once else 0
    if steps <= 0:
        return None
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in kernels.items() if not k.startswith("__amd_rocclr")) / steps
    gbs = total / (ms_per_step * 1e-3) / 1e9
    return {"bound": "hbm", "hbm_bytes_per_step": total, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "ms_per_step": ms_per_step,
            "note": "sum of rocprofv3 PMC bytes (FETCH_SIZE x 2 + WRITE_SIZE) of every kernel of one step / measured step time"}


def k3_roofline(cfg, dev, iters, B, config=2):
    """K3's sums of the mid-size dense-gradient tables (the tables above 16 rows that stay below the row-sparse limit:
    8 at config 2), timed stand-alone through the C ABI on the step's compact dX layout.  Algorithmic bytes per sample:
    n_tables * (4 E + 4) (one gradient row + one key each, SURVEY.md 8d's backward figure without the accumulator
    traffic, which stays in LDS / L2)."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec._hip import lib
    E = cfg["embed_dim"]
    mids = [v for v in cfg["vocabs"] if v > 16 and v * E * 4 <= (1 << 20)]
    if not mids:
        return None
    ld = len(mids) * E
    g = torch.Generator(device=dev).manual_seed(3)
    sets = [(torch.stack([torch.randint(0, v, (B,), device=dev, generator=g, dtype=torch.int32) for v in mids]).contiguous(),
             torch.randn(B, ld, device=dev, generator=g) * 1e-3) for _ in range(4)]
    grads = [torch.zeros(v, E, device=dev) for v in mids]
    slots = (H.EmbedGradSlot * len(mids))()
    for s_, v in enumerate(mids):
        slots[s_] = H.EmbedGradSlot(v, E, E * s_, s_, 0, grads[s_].data_ptr(), None, None)
    nbytes = lib.swr_embed_bwd_workspace_bytes(slots, len(mids), B)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    flag = H.err_flag(dev)
    st = {"i": 0}

    def launch():
        keys, dE = sets[st["i"] % 4]
        st["i"] += 1
        H.check(lib.swr_embed_bwd_reduce_part(slots, len(mids), H.ptr(keys), H.ptr(dE), ld, B, 2, H.ptr(ws), nbytes, H.ptr(flag),
                                              H.stream()), "swr_embed_bwd_reduce_part")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        H.check(lib.swr_embed_bwd_sort(slots, len(mids), H.ptr(sets[0][0]), B, H.ptr(ws), nbytes, H.stream()), "swr_embed_bwd_sort")
    ms = time_kernel_events(launch, max(10, iters), stream)
    alg = float(len(mids) * B * (4 * E + 4))
    gbs = alg / (ms * 1e-3) / 1e9
    return {"kernel": "direct_kernel + finalize_kernel (K3 sums of the %d mid-size tables)" % len(mids), "bound": "hbm",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms, "traffic": pmc_traffic("void direct_kernel<", config),
            "note": "a workgroup is a chain of five load -> convert -> LDS-atomic trips and a 64 KB slab store (latency-bound, DESIGN.md section 4); the slabs add 2 x 25 MB to the algorithmic bytes"}


def gather_roofline(cfg, model, x, dev, iters, B, config=2):
    """HBM roofline of K1 (the north star's >= 50 % target): algorithmic bytes per sample
    F_s (idx + 4 E) + 4 F_d + 4 K0 over the measured launch time of the fused lookup."""
    from scenario_wise_rec.basic.layers import fused_lookup
    if hasattr(model, "embedding"):
        # the lookup the training step launches: with the folded layout when the model's first layer takes it
        fold = cfg["family"] in ("MMOE", "SharedBottom") and model.training
        lookup = lambda b: model.embedding(b, model.features, squeeze_dim=True, onehot="layout" if fold else False)
    else:                                         # PPNet: id + agnostic groups in one fused lookup (ppnet.py:51-54)
        lookup = lambda b: fused_lookup(b, [(model.id_embedding, model.id_features), (model.agn_embedding, model.agn_features, True)])
    stream = torch.cuda.Stream()
    lazies = {}
    for p in model.parameters():                  # time the gather kernel alone: no lazy-row catch-up launches
        if hasattr(p, "_swr_lazy"):
            lazies[p] = p._swr_lazy
            del p._swr_lazy
    # four different batches with their own output buffers (4 x 135 MB + table rows > the 256 MB Infinity Cache), so
    # repeated launches do not measure a cache-resident replay of one batch
    batches = [x] + [{k: torch.from_numpy(v).to(dev) for k, v in synth_batch(cfg, B, seed=900 + j)[0].items()}
                     for j in range(3)]
    state = {"i": 0, "keep": []}

    def launch():
        state["keep"].append(lookup(batches[state["i"] % 4]))
        state["keep"] = state["keep"][-4:]
        state["i"] += 1
    try:
        with torch.no_grad():
            ms = time_kernel_events(launch, max(10, iters), stream)
    finally:
        for p, st in lazies.items():
            p._swr_lazy = st
    nbytes = gather_bytes_per_sample(cfg) * B
    ref = {"algorithmic_bytes_per_launch": nbytes, "achieved": nbytes / (ms * 1e-3) / 1e9,
           "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "note": "SURVEY.md 8(d)'s per-sample bytes of the REFERENCE lookup (F_s (idx + 4E) + 4 F_d + 4 K0) over this launch's "
                   "time: a speed-up figure against the reference's traffic, not an HBM efficiency"}
    exe, layout = nbytes, "plain concat [B, K0]: executed = algorithmic"
    kernel, pmc = "embed_gather_kernel", "void embed_gather_kernel<4>"
    last = state["keep"][-1] if state["keep"] else None
    info = getattr(last, "_swr_onehot", None)
    if info is not None and info.fold and info.fl is not None:
        # fused lookup (csrc/first_layer.hip): ids in; row keys of the tables that keep an embedding, 32 B of one-hot bits, a
        # 4-byte piece offset per (sample, 8-column piece), the bf16 terms of the fp32-sourced pieces (row-sparse tables, dense
        # features: 48 B each, their 32-byte sources read), the fp32 dense block out
        fs_ = len(cfg["vocabs"])
        o = info.fl["offs"]
        exe = B * (fs_ * 8 + 4 * len(info.compact) + 32 + 4 * 2 * (info.Kp // 16) + o.n_fpieces * (48 + 32) + 4 * o.nd4)
        layout = (f"fused: keys of {len(info.compact)} tables, {info.oh_width} one-hot bits, {2 * (info.Kp // 16)} piece offsets, "
                  f"{o.n_fpieces} fp32-sourced pieces per sample; nothing of the concat is written")
        kernel, pmc = "fl_keys_kernel", "void fl_keys_kernel<"
    elif info is not None and info.fold:
        # folded layout: the small tables' embeddings are neither read nor written; what the launch really moves
        fs_, e_ = len(cfg["vocabs"]), cfg["embed_dim"]
        exe = B * (fs_ * 8 + 4 * (info.Kp + info.oh_width) + 4 * e_ * len(info.compact) + 4 * cfg["n_dense"])
        layout = f"folded: [{len(info.compact)} tables x {e_} | {cfg['n_dense']} dense | {info.oh_width} one-hot columns]"
    achieved = exe / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "executed_bytes_per_launch": exe, "avg_launch_ms": ms, "layout": layout,
            "traffic": pmc_traffic(pmc, config), "vs_reference_bytes": ref}


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
