"""Worker of the data-parallel tests (one process per rank; launched by tests/test_parallel*.py).

    python tests/dp_worker.py <mode> <case> <out_dir>

mode "exchange-cpu": gloo on CPU.  Each rank takes its row shard of the golden case, gets its LOCAL gradients
    from the oracle (test infrastructure), packs them the way the product does (flat dense arena + row-sparse
    entries for the largest table) and runs the product's exchange step (scenario_wise_rec.parallel) with a
    numpy row merge injected.  Rank 0 dumps the exchanged gradients.
mode "full-gpu" / "graph-gpu": gloo over device tensors, every rank on cuda:0.  The whole HIP path (DataParallelStep) for one
    step; rank 0 dumps gradients and the state after the step.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "scenario-wise-rec_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

from _golden import Case, build_product_model, make_oracle, to_device


def shard(case, rank, world):
    x, y = case.batch(0)
    B = len(y)
    sh = B // world
    return {k: v[rank * sh:(rank + 1) * sh] for k, v in x.items()}, y[rank * sh:(rank + 1) * sh]


def numpy_merge_rows(urow, ugrad, vocab):
    """Oracle-side stand-in for the HIP row merge: same output convention (sorted rows first-listed, -1 padding)."""
    r, g = urow.numpy(), ugrad.numpy().astype(np.float64)
    full = np.zeros((vocab, g.shape[1]))
    np.add.at(full, r[r >= 0], g[r >= 0])
    rows = np.unique(r[r >= 0])
    out_r = np.full(len(r), -1, np.int32)
    out_g = np.zeros_like(g, dtype=np.float32)
    out_r[:len(rows)] = rows
    out_g[:len(rows)] = full[rows]
    return torch.from_numpy(out_r), torch.from_numpy(out_g)


def dump_gradients(model):
    """The gradients the optimizer step consumed: arena views as they are, row lists of the large tables as dense arrays."""
    out = {}
    for k, p in model.named_parameters():
        sg = getattr(p, "_swr_sparse_grad", None)
        if sg is not None:
            r, g = sg[0].cpu().numpy(), sg[1].cpu().numpy()
            full = np.zeros(tuple(p.shape), np.float64)
            np.add.at(full, r[r >= 0], g[r >= 0].astype(np.float64))
            out[k] = full.astype(np.float32)
        elif p.grad is not None and getattr(p, "_swr_touched", True):
            out[k] = p.grad.cpu().numpy()
    return out


def main():
    mode, name, out_dir = sys.argv[1:4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("DP_WORKER_BACKEND", "gloo")      # "nccl": RCCL (one rank per GPU: world 1 on a one-GPU box)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "torch-only":
        # control experiment of tools/dp8_soak.py: the SAME process topology (`world` processes on cuda:0, gloo collectives
        # over device tensors) running nothing but PyTorch's own kernels -- libswr is never loaded.  A runtime GPU fault
        # here cannot come from this repository's kernels or stream edges.
        torch.cuda.set_device(0)
        g = torch.Generator(device="cuda").manual_seed(rank)
        a = torch.randn(1024, 1024, device="cuda", generator=g)
        acc = torch.zeros(1024, 1024, device="cuda")
        for _ in range(20):
            acc = torch.relu(acc + a @ a.t() * 1e-3)
            buf = torch.empty(world * 4096, device="cuda")
            dist.all_gather_into_tensor(buf, acc.reshape(-1)[:4096].contiguous())
            acc = acc + buf[:4096].sum() * 1e-9
        torch.cuda.synchronize()
        assert bool(torch.isfinite(acc).all())
        dist.barrier()
        dist.destroy_process_group()
        return
    case = Case(name)
    x, y = shard(case, rank, world)
    if mode == "exchange-cpu":
        from scenario_wise_rec.parallel import exchange_gradients
        om = make_oracle(case)
        _, _, grads = om.loss_and_grads(x, y)
        keys = sorted(grads)
        big = max((k for k in keys if "embed_dict" in k), key=lambda k: grads[k].size)
        dense_keys = [k for k in keys if k != big]
        flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in dense_keys]).astype(np.float32))
        gb = grads[big]
        rows = np.flatnonzero(np.abs(gb).sum(1) > 0).astype(np.int32)
        n = len(y)                                   # one entry per looked-up sample, -1 padded (product convention)
        urow = np.full(n, -1, np.int32); urow[:len(rows)] = rows
        ugrad = np.zeros((n, gb.shape[1]), np.float32); ugrad[:len(rows)] = gb[rows]
        merged = exchange_gradients(flat, [(torch.from_numpy(urow), torch.from_numpy(ugrad), gb.shape[0])], world,
                                    merge_rows=numpy_merge_rows)
        if rank == 0:
            out, off = {}, 0
            for k in dense_keys:
                out[k] = flat[off:off + grads[k].size].numpy().reshape(grads[k].shape); off += grads[k].size
            r, g = merged[0]
            full = np.zeros(gb.shape, np.float32)
            full[r.numpy()[r.numpy() >= 0]] = g.numpy()[r.numpy() >= 0]
            out[big] = full
            np.savez(os.path.join(out_dir, "exchanged.npz"), **out)
    else:
        from scenario_wise_rec import _hip as H
        from scenario_wise_rec.parallel import DataParallelStep
        from scenario_wise_rec.trainers import CTRTrainer
        from scenario_wise_rec.basic.module import SwrModule
        if len(sys.argv) > 4 and mode != "trainer-gpu":
            SwrModule.dense_table_limit_bytes = int(sys.argv[4])     # force the row-sparse path for the larger tables
        if mode == "trainer-gpu":
            SwrModule.dense_table_limit_bytes = 2048
            os.environ["SWR_DP_BACKEND"] = "gloo"
        torch.cuda.set_device(0)
        if mode == "trainer-gpu":
            # the reference's multi-GPU entry, CTRTrainer(gpus=[...]) (ctr_trainer.py:45-47): every process gets the WHOLE
            # batch from its (identical) loader and trains on its row chunk; 4 epochs over a one-batch loader = two eager
            # steps, the capture, one more replay
            model = build_product_model(case, device="cpu")
            tr = CTRTrainer(model, "dp", optimizer_params={"lr": case.meta["lr"], "weight_decay": case.meta["weight_decay"]},
                            gpus=[0] * world, n_epoch=1)
            tr.use_graph = os.environ.get("DP_EAGER_REFERENCE") is None
            xf, yf = case.batch(0)
            loader = [({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in xf.items()}, torch.from_numpy(yf))]
            model.train()
            n_epochs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
            for _ in range(n_epochs):
                tr.train_one_epoch(loader)
            torch.cuda.synchronize()
            H.check_errors()
            np.savez(os.path.join(out_dir, "state1.npz" if rank == 0 else f"state1_rank{rank}.npz"),
                     **{k: v.cpu().numpy() for k, v in model.state_dict().items()})
            dist.barrier()
            dist.destroy_process_group()
            return
        model = build_product_model(case, device="cuda:0")
        tr = CTRTrainer(model, "dp", optimizer_params={"lr": case.meta["lr"], "weight_decay": case.meta["weight_decay"]},
                        device="cuda:0")
        step = DataParallelStep(tr, world)
        model.train()
        if mode == "full-gpu":
            tr.optimizer.clear_grads = False       # the exchanged gradients are dumped AFTER the step
        if mode == "graph-gpu":
            # two captured graphs + eager collectives; the captured step must land where the eager one does.
            # capture() runs 2 eager warm-up steps, so compare after 3 steps in total on the same batch.
            xd, yd = to_device(x, "cuda:0"), torch.from_numpy(y).cuda()
            if os.environ.get("DP_EAGER_REFERENCE"):
                for _ in range(3):
                    step.train_step(xd, yd)
            else:
                step.capture(xd, yd, warmup=2)
                step.replay()
        else:
            step.train_step(to_device(x, "cuda:0"), torch.from_numpy(y).cuda())
        torch.cuda.synchronize()
        H.check_errors()
        # every rank dumps the state it holds and (eager mode) the exchanged gradients the optimizer consumed: the
        # test compares gradients -- not only the post-Adam state, which sees little more than their signs -- with the
        # reference's DataParallel result, and checks that all replicas are bitwise equal
        np.savez(os.path.join(out_dir, "state1.npz" if rank == 0 else f"state1_rank{rank}.npz"),
                 **{k: v.cpu().numpy() for k, v in model.state_dict().items()})
        if mode == "full-gpu":
            np.savez(os.path.join(out_dir, f"grads_rank{rank}.npz"), **dump_gradients(model))
            # what this rank RECEIVED: every rank's gradient arena (pre-exchange local gradients, by parameter name)
            xb, arena = step._xb, model.arena()
            if xb["allreduce"]:                      # (all-reduced arena: no per-rank copies exist on the receiver)
                dist.barrier()
                dist.destroy_process_group()
                return
            recv = xb["recv_d"].cpu().numpy().reshape(world, -1)
            names = {id(p): k for k, p in model.named_parameters()}
            local = {}
            for p, off, n in arena["spans"]:
                if p.requires_grad:
                    local[names[id(p)]] = recv[:, off:off + n].reshape((world,) + tuple(p.shape)).copy()
            np.savez(os.path.join(out_dir, f"received_rank{rank}.npz"), **local)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
