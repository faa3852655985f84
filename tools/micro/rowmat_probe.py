"""Stand-alone timing of swr_rowmat_fwd / _bwd at HAMUR's config-5 shape (B 32768, D 8, k 35), HIP events."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scenario-wise-rec_amd"))
from scenario_wise_rec import ops

B, D, k = 32768, 8, 35
sets = [(torch.randn(B, D, k, device="cuda"), torch.randn(B, k, k, device="cuda"), torch.randn(B, D, k, device="cuda")) for _ in range(4)]
def run(fn, n=40):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print("fwd us", round(run(lambda i: ops.RowMat.apply(sets[i % 4][0], sets[i % 4][1])), 1))
T, Hm, dO = sets[0]
T.requires_grad_(True); Hm.requires_grad_(True)
def fb(i):
    out = ops.RowMat.apply(T, Hm); out.backward(dO); T.grad = None; Hm.grad = None
print("fwd+bwd us", round(run(fb), 1))
