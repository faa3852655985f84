"""Multi-threaded CPU port of the reference's training step for every in-scope model family (oracle only).

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product path.

The numpy tape (oracle/tape.py) is the checker of record: exact, but its elementwise operations run on one thread, so
(a) timing it says little about what the reference achieves on a many-core host and (b) it cannot evaluate a 32 768-row
batch of HAMUR / PPNet in test time.  This module restates the same step -- the model's `forward`, `BCELoss`
(`trainers/ctr_trainer.py:56,70`), `loss.backward()`, dense `Adam(lr, weight_decay)` over every parameter
(`ctr_trainer.py:50-52,73`) -- on torch CPU operators under `torch.set_num_threads(n)`, which is what the reference
itself executes when run with `--device cpu`.  Families and the reference forwards they restate:

    SharedBottom  models/multi_domain/sharebottom.py:28-50      MMOE   models/multi_domain/mmoe.py:33-56
    PLE           models/multi_domain/ple.py:41-64,107-136      Star   models/multi_domain/star.py:78-118
    PPNet         models/multi_domain/ppnet.py:21-29,47-67      EPNet  models/multi_domain/epnet.py:25-33
    HamurSmall / HamurLarge   models/multi_domain/hamur.py:308-378 / 101-244

Every family is pinned on the reference-generated golden vectors (tests/test_oracle_golden.py::
test_torch_port_matches_the_golden_vectors: probabilities, loss, every gradient, state after one Adam step), so the
numbers bench.py reports as `cpu_baseline` and the full-shard gradients tests/test_full_size_gpu.py compares with belong
to the computation the HIP path is checked against.  State keys are the reference's.

`dtype=torch.float64` turns the port into a fast fp64 oracle for the large shards; `materialize_adapter=False` evaluates
HAMUR's adapter as ((h U) H_b) V instead of building the per-sample weight U H_b V (`hamur.py:346,355`): the same
function, 36 x fewer flops and no [B, m, 32] tensor -- the form for 32 768-row oracles; the cpu_baseline leg keeps the
reference's materialised form.
"""
import numpy as np
import torch
import torch.nn.functional as F

BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


class TorchPort(object):
    def __init__(self, family, hyper, state, dtype=torch.float32, threads=None, materialize_adapter=True, track_buffers=True):
        """hyper: the family's constructor arguments with oracle.nn feature lists (`features`, or `id_features` +
        `agn_features`, or `sce_features` + `agn_features`); state: reference-keyed arrays."""
        if threads:
            torch.set_num_threads(int(threads))
        self.family, self.h, self.dtype = family, dict(hyper), dtype
        self.materialize_adapter, self.track = materialize_adapter, track_buffers
        self.training = True
        self.p, self.buf = {}, {}
        for k, v in state.items():
            t = torch.from_numpy(np.array(v))
            if k.endswith(BUFFER_SUFFIXES):
                self.buf[k] = t.to(dtype) if t.is_floating_point() else t
            else:
                self.p[k] = t.to(dtype).requires_grad_(True)
        self.opt = None

    # ---------------------------------------------------------------- layers (basic/layers.py)
    def _embed(self, prefix, x, features, detach=False):
        """`EmbeddingLayer.forward(squeeze_dim=True)` (`basic/layers.py:64-105`): sparse block first, dense block last."""
        from .nn import Sparse
        sparse = [F.embedding(torch.as_tensor(np.asarray(x[f.name])).long(),
                              self.p[f"{prefix}.embed_dict.{f.shared_with or f.name}.weight"])
                  for f in features if isinstance(f, Sparse)]
        dense = [torch.as_tensor(np.asarray(x[f.name])).float().to(self.dtype).unsqueeze(1)
                 for f in features if not isinstance(f, Sparse)]
        e = torch.cat(sparse + dense, dim=1)
        return e.detach() if detach else e

    def _bn(self, pre, z):
        """nn.BatchNorm1d(eps 1e-5, momentum 0.1) in the port's mode."""
        rm, rv = self.buf[pre + ".running_mean"], self.buf[pre + ".running_var"]
        if self.training and self.track:
            self.buf[pre + ".num_batches_tracked"] += 1
        if self.training and not self.track:
            rm = rv = None
        return F.batch_norm(z, rm, rv, self.p[pre + ".weight"], self.p[pre + ".bias"], self.training, 0.1, 1e-5)

    def _mlp(self, pre, x, dims, output_layer, activation="relu"):
        """`MLP` (`basic/layers.py:231-264`): [Linear, BatchNorm1d, act, Dropout(0)] per dim at `.mlp.{4i..}` (+ Linear(., 1))."""
        i = 0
        for _ in dims:
            x = self._bn(f"{pre}.mlp.{i + 1}", F.linear(x, self.p[f"{pre}.mlp.{i}.weight"], self.p[f"{pre}.mlp.{i}.bias"]))
            x = torch.softmax(x, dim=1) if activation == "softmax" else torch.relu(x)
            i += 4
        if output_layer:
            x = F.linear(x, self.p[f"{pre}.mlp.{i}.weight"], self.p[f"{pre}.mlp.{i}.bias"])
        return x

    def _gate_nu(self, pre, x):
        """`GateNU` (`basic/layers.py:307-320`): 2 sigmoid(W2 relu(W1 x + b1) + b2)."""
        h = torch.relu(F.linear(x, self.p[pre + ".network.0.weight"], self.p[pre + ".network.0.bias"]))
        return 2.0 * torch.sigmoid(F.linear(h, self.p[pre + ".network.2.weight"], self.p[pre + ".network.2.bias"]))

    @staticmethod
    def _select(ys, dom):
        """`final = where(domain_id == d, y_d, final)` from zeros (`mmoe.py:53-55`): ids outside [0, D) give 0.0."""
        out = torch.zeros_like(ys[0])
        for d, y in enumerate(ys):
            out = torch.where((dom == d).unsqueeze(1), y, out)
        return out

    @staticmethod
    def _mix(gate, experts):
        return (gate.unsqueeze(-1) * torch.stack(experts, dim=1)).sum(dim=1)

    # ---------------------------------------------------------------- families
    def _sharedbottom(self, x):
        h = self.h
        e = self._embed("embedding", x, h["features"])
        z = self._mlp("bottom_mlp", e, h["bottom_params"]["dims"], False)
        return [torch.sigmoid(self._mlp(f"towers.{d}", z, h["tower_params"]["dims"], True)) for d in range(h["domain_num"])]

    def _mmoe(self, x):
        h = self.h
        e = self._embed("embedding", x, h["features"])
        experts = [self._mlp(f"experts.{j}", e, h["expert_params"]["dims"], False) for j in range(h["n_expert"])]
        ys = []
        for d in range(h["domain_num"]):
            gate = self._mlp(f"gates.{d}", e, [h["n_expert"]], False, "softmax")
            ys.append(torch.sigmoid(self._mlp(f"towers.{d}", self._mix(gate, experts), h["tower_params"]["dims"], True)))
        return ys

    def _ple(self, x):
        h = self.h
        D, ns, nsh = h["domain_num"], h["n_expert_specific"], h["n_expert_shared"]
        e = self._embed("embedding", x, h["features"])
        inputs = [e] * (D + 1)
        dims = h["expert_params"]["dims"]
        for lvl in range(h["n_level"]):
            pre = f"cgc_layers.{lvl}"
            spec = [self._mlp(f"{pre}.experts_specific.{d * ns + i}", inputs[d], dims, False) for d in range(D) for i in range(ns)]
            shared = [self._mlp(f"{pre}.experts_shared.{i}", inputs[-1], dims, False) for i in range(nsh)]
            outs = []
            for d in range(D):
                g = self._mlp(f"{pre}.gates_specific.{d}", inputs[d], [ns + nsh], False, "softmax")
                outs.append(self._mix(g, spec[d * ns:(d + 1) * ns] + shared))
            if lvl + 1 < h["n_level"]:
                g = self._mlp(f"{pre}.gate_shared", inputs[-1], [ns * D + nsh], False, "softmax")
                outs.append(self._mix(g, spec + shared))
            inputs = outs
        return [torch.sigmoid(self._mlp(f"towers.{d}", inputs[d], h["tower_params"]["dims"], True)) for d in range(D)]

    def _star(self, x):
        """Partitioned norm over the whole batch (biased variance, eps 1e-6), factorised weights [in, out], BN + ReLU after
        EVERY layer including the 1-wide last one; sigmoid(select + aux) (`star.py:88-117`)."""
        h, p = self.h, self.p
        e = self._embed("embedding", x, h["features"])
        aux = self._mlp("auxnet", e, h["aux_dims"], True)
        mean = e.mean(dim=0)
        cen = e - mean
        n = cen / torch.sqrt((cen * cen).mean(dim=0) + 1e-6)
        outs = []
        for d in range(h["num_domains"]):
            z = (p["dn_share_gamma"] * p[f"domain_specific_dn_gamma.{d}"]) * n + p["dn_share_bias"] + p[f"domain_specific_dn_bias.{d}"]
            for l in range(len(h["fcn_dims"]) + 1):
                w = p[f"share_parm_w.{l}"] * p[f"domain_specific_w.{d}.{l}"]
                z = torch.relu(self._bn(f"domain_specific_bn.{d}.{l}", z @ w + p[f"share_parm_b.{l}"] + p[f"domain_specific_b.{d}.{l}"]))
            outs.append(z)
        return outs, aux

    def _ppnet(self, x):
        h, p = self.h, self.p
        g_in = torch.cat([self._embed("id_embedding", x, h["id_features"]),
                          self._embed("agn_embedding", x, h["agn_features"], detach=True)], dim=1)
        outs = []
        for d in range(h["domain_num"]):
            z = g_in
            for l, width in enumerate(h["fcn_dims"]):
                z = self._mlp(f"domain_tower.{d}.mlp_layers.{l}", z, [width], False) * self._gate_nu(f"domain_tower.{d}.gate_layers.{l}", g_in)
            outs.append(torch.sigmoid(F.linear(z, p[f"domain_tower.{d}.final_layer.weight"], p[f"domain_tower.{d}.final_layer.bias"])))
        return outs

    def _epnet(self, x):
        h, p = self.h, self.p
        sce = self._embed("sce_embedding", x, h["sce_features"])
        agn = self._embed("agn_embedding", x, h["agn_features"])
        gate = self._gate_nu("gatenu", torch.cat([sce, agn.detach()], dim=1))
        return torch.sigmoid(F.linear(agn * gate, p["mlp.mlp.0.weight"], p["mlp.mlp.0.bias"]))

    def _adapter(self, z, Hm, iu, gamma, bias):
        """Adapter cell (`hamur.py:344-367`, `175-198`): sigmoid(z (U0 H_b V0) + b0) (U1 H_b V1) + b1, domain norm over the
        batch (UNBIASED variance, eps 1e-5), scale / shift, residual."""
        p = self.p

        def proj(t, i):
            if self.materialize_adapter:
                w = torch.einsum("mi,bij,jn->bmn", p[f"u.{i}"], Hm, p[f"v.{i}"])
                return torch.einsum("bf,bfj->bj", t, w) + p[f"b_list.{i}"]
            return torch.einsum("bi,bij->bj", t @ p[f"u.{i}"], Hm) @ p[f"v.{i}"] + p[f"b_list.{i}"]
        t = proj(torch.sigmoid(proj(z, iu)), iu + 1)
        cen = t - t.mean(dim=0)
        var = (cen * cen).sum(dim=0) / float(t.shape[0] - 1)
        return p[gamma] * (cen / torch.sqrt(var + 1e-5)) + p[bias] + z

    def _hamur(self, x, large):
        h, p = self.h, self.p
        k = h["k"]
        e = self._embed("embedding", x, h["features"])
        n_hyper = sum(1 for key in p if key.startswith("hyper_net.") and key.endswith(".weight")) // 2   # Linear + BN per block
        n_blocks = 7 if large else 2
        outs = []
        for d in range(h["domain_num"]):
            z = e                                       # the shared hyper-net runs inside the domain loop (`hamur.py:315`): D BN updates
            for i in range(n_hyper):
                z = torch.relu(self._bn(f"hyper_net.{4 * i + 1}", F.linear(z, p[f"hyper_net.{4 * i}.weight"], p[f"hyper_net.{4 * i}.bias"])))
            Hm = z.reshape(-1, k, k)
            z = e
            for blk in range(n_blocks):
                pre = f"layer_list.{d}"
                z = torch.relu(self._bn(f"{pre}.{3 * blk + 1}", F.linear(z, p[f"{pre}.{3 * blk}.weight"], p[f"{pre}.{3 * blk}.bias"])))
                if large and blk == 5:
                    z = self._adapter(z, Hm, 0, "gamma1", "bias1")
                elif large and blk == 6:
                    z = self._adapter(z, Hm, 2, "gamma2", "bias2")
                elif not large and blk == 1:
                    z = self._adapter(z, Hm, 0, "gamma1", "bias1")
            last = 3 * n_blocks
            outs.append(torch.sigmoid(F.linear(z, p[f"layer_list.{d}.{last}.weight"], p[f"layer_list.{d}.{last}.bias"])))
        return outs

    # ---------------------------------------------------------------- step
    def forward(self, x):
        dom = torch.as_tensor(np.asarray(x["domain_indicator"])).long() if "domain_indicator" in x else None
        fam = self.family
        if fam == "Star":
            outs, aux = self._star(x)
            return torch.sigmoid(self._select(outs, dom) + aux).squeeze(1)
        if fam == "EPNet":
            return self._epnet(x).squeeze(1)
        ys = {"SharedBottom": self._sharedbottom, "MMOE": self._mmoe, "PLE": self._ple, "PPNet": self._ppnet,
              "HamurSmall": lambda q: self._hamur(q, False), "HamurLarge": lambda q: self._hamur(q, True)}[fam](x)
        return self._select(ys, dom).squeeze(1)

    def predict(self, x):
        self.training = False
        try:
            with torch.no_grad():
                return self.forward(x).numpy()
        finally:
            self.training = True

    def _loss(self, p, y):
        return F.binary_cross_entropy(p, torch.as_tensor(np.asarray(y)).float().to(self.dtype))

    def step(self, x, y, lr=1e-3, weight_decay=1e-5):
        """One training step (`ctr_trainer.py:69-73`); returns (probabilities, loss)."""
        if self.opt is None:
            self.opt = torch.optim.Adam(list(self.p.values()), lr=lr, weight_decay=weight_decay)
        p = self.forward(x)
        loss = self._loss(p, y)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return p.detach().numpy(), float(loss.detach())

    def loss_and_grads(self, x, y):
        for t in self.p.values():
            t.grad = None
        p = self.forward(x)
        loss = self._loss(p, y)
        loss.backward()
        return p.detach().numpy(), float(loss.detach()), {k: t.grad.numpy() for k, t in self.p.items() if t.grad is not None}


class MMoEPort(TorchPort):
    """The MMoE step with the feature list passed separately (bench.py's `cpu_baseline` leg of config 2)."""

    def __init__(self, features, hyper, state, threads=None):
        super().__init__("MMOE", dict(hyper, features=features), state, threads=threads)
