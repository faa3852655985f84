"""PPNet (reference: `models/multi_domain/ppnet.py:8-67`)."""
import torch
import torch.nn as nn

from ... import ops
from ...basic.activation import activation_layer
from ...basic.layers import MLP, EmbeddingLayer, GateNU, LayerBank, fused_lookup
from ...basic.module import SwrModule


class PPTowerBlock(SwrModule):
    """h_0 = g;  h_{l+1} = ReLU(BN(Linear_l(h_l))) * GateNU_l(g);  out = sigmoid(Linear(h_L))
    (`ppnet.py:8-29`).  NB the tower input is the GATE input; `agn_emb` is accepted and unused."""

    def __init__(self, input_dim, fcn_dims):
        super().__init__()
        self.input_dim = input_dim
        self.dims = [self.input_dim] + fcn_dims
        self.gate_layers = nn.ModuleList()
        self.mlp_layers = nn.ModuleList()
        for i in range(len(self.dims) - 1):
            self.mlp_layers.append(MLP(input_dim=self.dims[i], dims=[self.dims[i + 1]], output_layer=False))
            self.gate_layers.append(GateNU(self.dims[0], self.dims[i + 1]))
        self.final_layer = nn.Linear(self.dims[-1], 1)
        self.sig = activation_layer("sigmoid")

    def _hidden(self, gate_input_emb):
        hidden = gate_input_emb
        for i in range(len(self.mlp_layers)):
            hidden = ops.mul(self.mlp_layers[i](hidden), self.gate_layers[i](gate_input_emb))
        return hidden

    def forward(self, agn_emb, gate_input_emb):
        return LayerBank([self.final_layer], None, ["sigmoid"])(self._hidden(gate_input_emb), self.training)

    def logits(self, gate_input_emb):
        """forward() before its sigmoid (the routed inference path selects logits)."""
        return LayerBank([self.final_layer], None, [None])(self._hidden(gate_input_emb), self.training)


class PPNet(SwrModule):
    """One PPTowerBlock per domain on the whole batch, domain select (`ppnet.py:32-67`).

    Fused evaluation: ONE lookup for id + agnostic features; the first MLP layer of all D towers and the
    first GateNU layer of all D*L gates are one stacked product on the shared gate input; everything
    deeper runs as launches grouped over the domains."""

    def __init__(self, id_features, agn_features, domain_num, fcn_dims):
        super().__init__()
        self.id_features = id_features
        self.agn_features = agn_features
        self.domain_num = domain_num
        self.id_embedding = EmbeddingLayer(id_features)
        self.agn_embedding = EmbeddingLayer(agn_features)
        self.id_dims = sum([fea.embed_dim for fea in id_features])
        self.agn_dims = sum([fea.embed_dim for fea in agn_features])
        self.input_dims = self.id_dims + self.agn_dims
        self.domain_tower = nn.ModuleList()
        for i in range(domain_num):
            self.domain_tower.append(PPTowerBlock(self.input_dims, fcn_dims))
        self.n_layers = len(fcn_dims)

    def _first_banks(self):
        T, L = list(self.domain_tower), self.n_layers
        mlp0 = LayerBank([t.mlp_layers[0].block(0)[0] for t in T], [t.mlp_layers[0].block(0)[1] for t in T],
                         ["relu"] * len(T))
        gate0 = LayerBank([t.gate_layers[l].network[0] for l in range(L) for t in T], None, ["relu"] * (L * len(T)))
        return mlp0, gate0

    def _fused_groups(self):
        T, L = list(self.domain_tower), self.n_layers
        mlp0, gate0 = self._first_banks()
        g = mlp0.tensor_groups() + gate0.tensor_groups()
        for l in range(L):
            if l > 0:
                g += LayerBank([t.mlp_layers[l].block(0)[0] for t in T], [t.mlp_layers[l].block(0)[1] for t in T]).tensor_groups()
            g += LayerBank([t.gate_layers[l].network[2] for t in T]).tensor_groups()
        g += LayerBank([t.final_layer for t in T]).tensor_groups()
        return g

    def forward(self, x):
        domain_id = x["domain_indicator"]
        T, L, D = list(self.domain_tower), self.n_layers, self.domain_num
        # cat(id_x, agn_x.detach()) (ppnet.py:54) as one lookup; the agnostic tables take no gradient
        gate_in = fused_lookup(x, [(self.id_embedding, self.id_features), (self.agn_embedding, self.agn_features, True)])
        if not self.training and ops.routed_eval_ok(gate_in):
            # inference: a row runs its own domain's tower only (PPTowerBlock.forward on that domain's rows)
            route = ops.DomainRouting(domain_id, D)
            gs = route.rows(gate_in)
            return ops.routed_probs(route, [T[d].logits(route.segment(gs, d)) if route.count(d) else None for d in range(D)])
        dims = T[0].dims
        # two stacked products on the shared input: the D first tower layers (BN + ReLU) and the hidden
        # layers of all D*L GateNUs (ReLU)
        mlp0, gate0 = self._first_banks()
        hidden = mlp0(gate_in, self.training)                        # [B, D*n_1]
        gh_all = gate0(gate_in, self.training)                       # [B, sum_l D*n_l]
        gh_blocks = ops.split_cols(gh_all, [D * dims[l + 1] for l in range(L)])      # one gradient tensor in the backward
        for l in range(L):
            n = dims[l + 1]
            gh = gh_blocks[l]
            gate_z = LayerBank([t.gate_layers[l].network[2] for t in T], None, [None] * D, grouped=True)(gh, self.training)
            if l > 0:
                hidden = LayerBank([t.mlp_layers[l].block(0)[0] for t in T], [t.mlp_layers[l].block(0)[1] for t in T],
                                   ["relu"] * D, grouped=True)(hidden, self.training)
            hidden = ops.mul_sigmoid(hidden, gate_z, T[0].gate_layers[l].gemma)      # hidden * (gamma * sigmoid(z)), one pass
        logits = LayerBank([t.final_layer for t in T], None, [None] * D, grouped=True)(hidden, self.training)   # [B, D]
        return ops.domain_select(logits, domain_id, apply_sigmoid=True)
