#!/usr/bin/env python3
"""Time the device half of the data-parallel exchange (swr_dp_finish) at world sizes 1..8 on ONE GPU, on synthetic
row lists shaped like config 2's large table (65 536 Zipf(1.05) lookups of 4.37 M rows per rank, E = 16, plus a
0.5 MB gradient arena): what the merge costs at the world size of a full node, without communication."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import numpy as np
import torch
from scenario_wise_rec import parallel

n, dim, vocab, A = 65536, 16, 4371900, 131072
rng = np.random.default_rng(0)
for world in (1, 2, 4, 8):
    total = (A + n + n * dim + 3) // 4 * 4
    recv = np.zeros((world, total), np.float32)
    for r in range(world):
        ids = np.sort((np.minimum(rng.zipf(1.05, size=n) - 1, vocab - 1) * 2654435761) % vocab).astype(np.int32)
        head = np.ones(n, bool); head[1:] = ids[1:] != ids[:-1]
        recv[r, A:A + n] = np.where(head, ids, ~ids).astype(np.int32).view(np.float32)
        recv[r, A + n:A + n + n * dim] = (rng.standard_normal((n, dim)) * head[:, None]).astype(np.float32).reshape(-1)
    R = torch.from_numpy(recv).cuda().reshape(-1)
    dense = torch.zeros(A, device="cuda")
    gathered = (R, A, [(A, A + n, n, dim, vocab)], total)
    for _ in range(3):
        parallel.finish(dense, gathered, world)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        merged = parallel.finish(dense, gathered, world)
    e1.record(); e1.synchronize()
    rows = merged[0][0]
    print(f"world {world}: swr_dp_finish {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  ({world * n} entries, {int((rows >= 0).sum())} distinct rows)")
