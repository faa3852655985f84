"""Multi-threaded CPU port of the reference's MMoE training step (oracle only; bench.py `cpu_baseline` leg).

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product path.

The numpy tape (oracle/tape.py) is the checker: exact, but its elementwise operations run on one thread, so timing it
says little about what the reference achieves on a many-core host.  This module restates the same step -- `MMOE.forward`
(`models/multi_domain/mmoe.py:33-56`), `BCELoss` (`trainers/ctr_trainer.py:56,70`), `loss.backward()`, dense
`Adam(lr, weight_decay)` over every parameter (`ctr_trainer.py:50-52,73`) -- on torch CPU operators under
`torch.set_num_threads(all cores)`, which is what the reference itself executes when run with `--device cpu`.  It is
checked against the numpy oracle (tests/test_oracle_golden.py::test_torch_port_matches_the_oracle) so that the number
bench.py reports belongs to the same computation the HIP path is compared with.  State keys are the reference's
(SURVEY.md Appendix A.9).
"""
import numpy as np
import torch
import torch.nn.functional as F


class MMoEPort(object):
    def __init__(self, features, hyper, state, threads=None):
        """features: list of oracle.nn.Sparse / Dense; state: reference-keyed arrays (fp32)."""
        from .nn import Sparse
        if threads:
            torch.set_num_threads(int(threads))
        self.sparse = [f for f in features if isinstance(f, Sparse)]
        self.dense = [f for f in features if not isinstance(f, Sparse)]
        self.h = hyper
        if len(hyper["expert_params"]["dims"]) != 1 or len(hyper["tower_params"]["dims"]) != 1:
            raise NotImplementedError("MMoEPort restates the BASELINE configuration: one-layer experts and towers")
        self.p, self.buf = {}, {}
        for k, v in state.items():
            t = torch.from_numpy(np.array(v))
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                self.buf[k] = t
            else:
                self.p[k] = t.requires_grad_(True)
        self.opt = None

    def _block(self, pre, x, softmax=False):
        """Linear -> BatchNorm1d (training mode) -> ReLU / Softmax(dim=1)  (`basic/layers.py:253-258`)."""
        z = F.linear(x, self.p[pre + ".mlp.0.weight"], self.p[pre + ".mlp.0.bias"])
        z = F.batch_norm(z, self.buf[pre + ".mlp.1.running_mean"], self.buf[pre + ".mlp.1.running_var"],
                         self.p[pre + ".mlp.1.weight"], self.p[pre + ".mlp.1.bias"], True, 0.1, 1e-5)
        self.buf[pre + ".mlp.1.num_batches_tracked"] += 1
        return torch.softmax(z, dim=1) if softmax else torch.relu(z)

    def forward(self, x):
        emb = [F.embedding(torch.as_tensor(x[f.name]).long(),
                           self.p[f"embedding.embed_dict.{f.shared_with or f.name}.weight"]) for f in self.sparse]
        dense = [torch.as_tensor(x[f.name]).float().unsqueeze(1) for f in self.dense]
        e = torch.cat(emb + dense, dim=1)                                    # sparse block first, dense last
        D, ne = self.h["domain_num"], self.h["n_expert"]
        experts = torch.stack([self._block(f"experts.{j}", e) for j in range(ne)], dim=1)           # [B, ne, H]
        dom = torch.as_tensor(x["domain_indicator"]).long()
        out = torch.zeros(e.shape[0], 1)
        for d in range(D):
            gate = self._block(f"gates.{d}", e, softmax=True).unsqueeze(-1)                          # [B, ne, 1]
            pooled = (gate * experts).sum(dim=1)
            h = self._block(f"towers.{d}", pooled)
            y = torch.sigmoid(F.linear(h, self.p[f"towers.{d}.mlp.4.weight"], self.p[f"towers.{d}.mlp.4.bias"]))
            out = torch.where((dom == d).unsqueeze(1), y, out)
        return out.squeeze(1)

    def step(self, x, y, lr=1e-3, weight_decay=1e-5):
        """One training step (`ctr_trainer.py:69-73`); returns (probabilities, loss)."""
        if self.opt is None:
            self.opt = torch.optim.Adam(list(self.p.values()), lr=lr, weight_decay=weight_decay)
        p = self.forward(x)
        loss = F.binary_cross_entropy(p, torch.as_tensor(y).float())
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return p.detach().numpy(), float(loss.detach())

    def loss_and_grads(self, x, y):
        for t in self.p.values():
            t.grad = None
        p = self.forward(x)
        loss = F.binary_cross_entropy(p, torch.as_tensor(y).float())
        loss.backward()
        return p.detach().numpy(), float(loss.detach()), {k: t.grad.numpy() for k, t in self.p.items() if t.grad is not None}
