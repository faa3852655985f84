"""Dump the captured training step's hipGraph (config 2) and print, for chosen kernels, which kernels they wait for.
    python tools/graph_deps.py tower_bwd_stats gemm_rows_x6 direct_kernel adam_dense_rows"""
import os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
import bench
from scenario_wise_rec import ops
from scenario_wise_rec.trainers import CTRTrainer


def main():
    cfg = bench.CONFIGS[2]
    B = cfg["batch"]
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(cfg)
    tr = CTRTrainer(model, "deps", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda:0")
    tr.use_graph = False
    model.train()
    xh, yh = bench.synth_batch(cfg, B, seed=1)
    x = {n: torch.from_numpy(v).to(dev) for n, v in xh.items()}
    y = torch.from_numpy(yh).to(dev)
    for _ in range(3):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    snap = tr.optimizer.host_counts()
    with torch.cuda.graph(g):
        tr.train_step(x, y)
        ops.join_side_streams()
    tr.optimizer.restore_host_counts(snap)
    path = "/tmp/step_graph.dot"
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = g.raw_cuda_graph()
    rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(raw), path.encode(), ctypes.c_uint(1))      # verbose
    print("hipGraphDebugDotPrint rc", rc)
    txt = open(path).read()
    names = {}
    for m in re.finditer(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt):
        names[m.group(1)] = m.group(2).replace("\\n", " ")[:70]
    edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
    preds = {}
    for a, b in edges:
        preds.setdefault(b, []).append(a)
    print(len(names), "nodes", len(edges), "edges")
    for want in sys.argv[1:]:
        for nid, lab in names.items():
            if want in lab:
                print(f"{lab}\n    <- " + "\n    <- ".join(names.get(p, p) for p in preds.get(nid, [])))


if __name__ == "__main__":
    main()
