"""CPU oracle for the multi-domain CTR hot path — TEST INFRASTRUCTURE ONLY.

This package is a plain-numpy restatement of the arithmetic the reference
(Xiaopengli1/Scenario-Wise-Rec, `/root/reference`) runs for its training hot
path: EmbeddingLayer lookup -> shared/expert MLP stacks -> per-domain towers ->
domain select -> BCE -> backward -> Adam.

The reference itself is pure Python on top of PyTorch; the arithmetic lives in
PyTorch (third-party, un-vendored; `requirements.txt:5` pins torch==1.13.1,
this container runs 2.10.0).  The oracle therefore restates the *published*
semantics of `nn.Embedding`, `nn.Linear`, `nn.BatchNorm1d`, `Softmax`,
`sigmoid`, `BCELoss` and `Adam` in numpy (oracle/tape.py, oracle/nn.py,
oracle/optim.py) and the reference's own model code on top of them
(oracle/models.py), each function citing the reference file:line it follows.

Parity pinning: the reference ships no tests and no golden vectors for this
path, so the oracle is pinned against OUTPUTS OF THE
REFERENCE ITSELF, run in the build container: `tests/golden/make_golden.py`
imports `/root/reference`, runs every in-scope model family on small seeded
inputs and commits inputs + expected outputs as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks the oracle against every one of them.

Usage rule: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this package, and only as the checker.  The
product (`scenario-wise-rec_amd/`) never imports it and has no CPU fallback.
"""
