"""Progressive Layered Extraction (reference: `models/multi_domain/ple.py:13-136`).
paper: (RecSys'2020) PLE: A Novel Multi-Task Learning Model for Personalized Recommendations."""
import torch
import torch.nn as nn

from ... import ops
from ...basic.layers import MLP, EmbeddingLayer, LayerBank, mlp_bank_forward, mlp_bank_groups, mlp_bank_select
from ...basic.module import SwrModule


class PLE(SwrModule):
    """n_level CGC layers, then one tower per domain, sigmoid, domain select (`ple.py:41-64`)."""

    def __init__(self, features, domain_num, n_level, n_expert_specific, n_expert_shared, expert_params,
                 tower_params):
        super().__init__()
        self.features = features
        self.domain_num = domain_num
        self.n_level = n_level
        self.input_dims = sum([fea.embed_dim for fea in features])
        self.embedding = EmbeddingLayer(features)
        self.cgc_layers = nn.ModuleList(
            CGC(i + 1, n_level, self.domain_num, n_expert_specific, n_expert_shared, self.input_dims, expert_params)
            for i in range(n_level))
        self.towers = nn.ModuleList(
            MLP(expert_params["dims"][-1], output_layer=True, **tower_params) for i in range(self.domain_num))

    def _fused_groups(self):
        return mlp_bank_groups(list(self.towers))

    def forward(self, x):
        domain_id = x["domain_indicator"]
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        outs = None                       # level 1: every expert and gate reads the same embedding
        for i in range(self.n_level):
            outs = self.cgc_layers[i].forward_fused(embed_x if i == 0 else outs, shared=(i == 0))
        H_ = self.cgc_layers[-1].out_dim
        D = self.domain_num
        if not self.training and ops.routed_eval_ok(outs):
            # inference: a row's own tower only (the CGC levels mix experts across domains and stay whole-batch)
            route = ops.DomainRouting(domain_id, D)
            os_ = route.rows(outs)
            return ops.routed_probs(route, [self.towers[d](route.segment(os_, d)[:, d * H_:(d + 1) * H_]) if route.count(d) else None
                                            for d in range(D)])
        return mlp_bank_select(list(self.towers), outs[:, :D * H_], domain_id)                  # towers [B, D] -> select


class CGC(SwrModule):
    """Customized Gate Control layer (`ple.py:67-136`): per domain `n_expert_specific` experts fed by
    that domain's input, `n_expert_shared` experts fed by the shared input, one gate per domain over
    (its experts + the shared ones) and -- except on the last level -- a shared gate over ALL experts."""

    def __init__(self, cur_level, n_level, domain_num, n_expert_specific, n_expert_shared, input_dims, expert_params):
        super().__init__()
        self.cur_level = cur_level
        self.n_level = n_level
        self.domain_num = domain_num
        self.n_expert_specific = n_expert_specific
        self.n_expert_shared = n_expert_shared
        self.n_expert_all = n_expert_specific * self.domain_num + n_expert_shared
        input_dims = input_dims if cur_level == 1 else expert_params["dims"][-1]
        self.in_dim = input_dims
        self.out_dim = expert_params["dims"][-1]
        self.experts_specific = nn.ModuleList(
            MLP(input_dims, output_layer=False, **expert_params) for _ in range(self.domain_num * self.n_expert_specific))
        self.experts_shared = nn.ModuleList(
            MLP(input_dims, output_layer=False, **expert_params) for _ in range(self.n_expert_shared))
        self.gates_specific = nn.ModuleList(
            MLP(input_dims, **{"dims": [self.n_expert_specific + self.n_expert_shared], "activation": "softmax",
                               "output_layer": False}) for _ in range(self.domain_num))
        if cur_level < n_level:
            self.gate_shared = MLP(input_dims, **{"dims": [self.n_expert_all], "activation": "softmax",
                                                  "output_layer": False})

    def _members(self):
        ex = list(self.experts_specific) + list(self.experts_shared)
        gs = list(self.gates_specific) + ([self.gate_shared] if self.cur_level < self.n_level else [])
        return ex, gs

    def _bank_shared(self):
        """Level 1: first block of every expert + every gate, all on the same input."""
        ex, gs = self._members()
        e0 = ex[0]
        lin = [m.block(0)[0] for m in ex + gs]
        bns = [m.block(0)[1] for m in ex + gs]
        acts = [e0.act] * len(ex) + [("softmax", g.block(0)[0].out_features) for g in gs]
        return LayerBank(lin, bns, acts)

    def _fusable(self):
        e0 = self.experts_specific[0]
        return e0.n_blocks >= 1 and e0.act in ("relu", "sigmoid") and e0.dropout_p == 0

    def _fused_groups(self):
        ex, _ = self._members()
        g = []
        if self.cur_level == 1 and self._fusable():
            g += self._bank_shared().tensor_groups()
        return g + mlp_bank_groups(ex)

    def forward_fused(self, x, shared):
        """x: [B, K] (level 1, `shared`) or [B, (D+1)*K] with the previous level's D+1 outputs side by side.
        Returns [B, (D or D+1) * H]: the domain outputs, then the shared output on non-last levels."""
        D, ns, nsh = self.domain_num, self.n_expert_specific, self.n_expert_shared
        ex, gs = self._members()
        K, nE = self.in_dim, len(ex)
        g_sep = None
        if shared and self._fusable():
            y = self._bank_shared()(x, self.training)                  # [B, nE*H0 | gate columns]
            h0 = ex[0].block(0)[0].out_features
            xe, g_all = ops.split_cols(y, [nE * h0, y.shape[1] - nE * h0])     # (one gradient tensor in the backward)
            if ex[0].n_blocks > 1:
                xe = mlp_bank_forward(ex, xe, shared_input=False, first_block=1)
                g_sep = g_all          # the gates stay where the first layer wrote them (no concatenation with the experts' output)
                y = xe
        else:
            src = (lambda d: x) if shared else (lambda d: x[:, d * K:(d + 1) * K])
            xs = [m(src(i // ns if i < D * ns else D)) for i, m in enumerate(ex)]
            g_list = [g(src(d if d < D else D)) for d, g in enumerate(gs)]
            y = torch.cat(xs + g_list, dim=1)
        H_ = self.out_dim
        sel = [[d * ns + i for i in range(ns)] + [D * ns + j for j in range(nsh)] for d in range(D)]
        if g_sep is not None and not ops.moe_mix_separate_ok(y, g_sep, ops.make_mix_desc(D, ns + nsh, H_, 0, 0, ns + nsh, sel)):
            y, g_sep = torch.cat([y, g_sep], dim=1), None
        g0 = 0 if g_sep is not None else nE * H_                        # first gate column (in g_sep / in y)
        g_dom = g_shared = g_sep
        gs0 = g0 + D * (ns + nsh)                                       # first column of the shared gate
        if g_sep is not None and self.cur_level < self.n_level:
            # two mixes read the gate tensor: each gets its own block (a view with its own gradient slot)
            g_dom, g_shared = ops.split_cols(g_sep, [D * (ns + nsh), nE])
            gs0 = 0
        desc = ops.make_mix_desc(D, ns + nsh, H_, 0, g0, ns + nsh, sel)
        out = ops.MoeMix.apply(y, desc, y.shape[1], g_dom)              # [B, D*H]
        if self.cur_level < self.n_level:
            desc_s = ops.make_mix_desc(1, nE, H_, 0, gs0, nE, [list(range(nE))])
            out = torch.cat([out, ops.MoeMix.apply(y, desc_s, y.shape[1], g_shared)], dim=1)
        return out

    def forward(self, x_list):
        """Reference-style call: list of D+1 inputs -> list of D (+1) outputs."""
        same = all(t is x_list[0] for t in x_list)
        out = self.forward_fused(x_list[0] if same else torch.cat(list(x_list), dim=1), shared=same)
        H_ = self.out_dim
        return [out[:, i * H_:(i + 1) * H_] for i in range(out.shape[1] // H_)]
