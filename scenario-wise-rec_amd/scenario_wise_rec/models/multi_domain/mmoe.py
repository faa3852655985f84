"""Multi-gate Mixture-of-Experts (reference: `models/multi_domain/mmoe.py:6-56`)."""
import os

import torch
import torch.nn as nn

from ... import ops
from ...basic.layers import MLP, EmbeddingLayer, LayerBank, _bn_dict, mlp_bank_forward, mlp_bank_groups, mlp_bank_select
from ...basic.module import SwrModule


class MMOE(SwrModule):
    """n_expert expert MLPs, one gate (Linear -> BatchNorm1d -> Softmax) and one tower per domain.

    Every branch runs on the whole batch and the domain is selected at the very end (`mmoe.py:44-56`),
    so in training mode the BatchNorm statistics of every expert, gate and tower see all rows.  Here:

      fused gather -> ONE stacked product for the first layer of all experts and all gates (its epilogue
      produces the BatchNorm statistics) -> BN + ReLU / BN + softmax -> gate mix -> grouped tower
      products -> domain select.
    """

    def __init__(self, features, domain_num, n_expert, expert_params, tower_params):
        super().__init__()
        self.features = features
        self.domain_num = domain_num
        self.n_expert = n_expert
        self.embedding = EmbeddingLayer(features)
        self.input_dims = sum([fea.embed_dim for fea in features])
        self.experts = nn.ModuleList(
            MLP(self.input_dims, output_layer=False, **expert_params) for i in range(self.n_expert))
        self.gates = nn.ModuleList(
            MLP(self.input_dims, output_layer=False, **{"dims": [self.n_expert], "activation": "softmax"})
            for i in range(self.domain_num))
        self.towers = nn.ModuleList(MLP(expert_params["dims"][-1], **tower_params) for i in range(self.domain_num))

    # experts' and gates' first layers read the same embedding: evaluate them together when possible
    def _fusable(self):
        e0 = self.experts[0]
        return e0.n_blocks >= 1 and e0.act is not None and e0.dropout_p == 0

    def _first_bank(self):
        ne, D = self.n_expert, self.domain_num
        e0 = self.experts[0]
        lin = [m.block(0)[0] for m in self.experts] + [g.block(0)[0] for g in self.gates]
        bns = [m.block(0)[1] for m in self.experts] + [g.block(0)[1] for g in self.gates]
        acts = [(e0.act, lin[0].out_features) if e0.act == "softmax" else e0.act] * ne + [("softmax", ne)] * D
        return LayerBank(lin, bns, acts)

    def _fused_groups(self):
        g = mlp_bank_groups(list(self.towers))
        if self._fusable():
            g += self._first_bank().tensor_groups()
            g += mlp_bank_groups(list(self.experts))        # deeper expert blocks (block 0 is already placed)
        return g

    def forward(self, x):
        domain_id = x["domain_indicator"]
        # [B, K0]; when the experts and gates are ONE stacked layer (the only reader of the lookup) the small tables ride
        # that layer's backward (ops.OneHotInfo)
        embed_x = self.embedding(x, self.features, squeeze_dim=True, onehot=self._fusable())
        ne, D = self.n_expert, self.domain_num
        experts, gates = list(self.experts), list(self.gates)
        if self._fusable():
            h0 = experts[0].block(0)[0].out_features
            if (self.training and experts[0].n_blocks == 1 and experts[0].act == "relu" and embed_x.is_cuda
                    and os.environ.get("SWR_BNMIX", "1") != "0" and (ne * h0 + D * ne) % 4 == 0
                    and ops.bnmix_supported(ne, h0, D)):
                # single-layer ReLU experts: BatchNorm + ReLU / softmax + gate mix in one pass, no [B, 148] activations
                pooled = self._first_bank()(embed_x, True, mix=(ne, h0, D))      # [B, D*H]
                return mlp_bank_select(list(self.towers), pooled, domain_id)
            y = self._first_bank()(embed_x, self.training)                       # [B, ne*H0 + D*ne]
            if experts[0].n_blocks > 1:
                y_ex, y_gate = ops.split_cols(y, [ne * h0, y.shape[1] - ne * h0])     # (one gradient tensor in the backward)
                ex = mlp_bank_forward(experts, y_ex, shared_input=False, first_block=1)
                Hx = experts[0].out_dim
                dsep = ops.make_mix_desc(D, ne, Hx, 0, 0, ne, [list(range(ne))] * D)
                if self.training and ops.moe_mix_separate_ok(ex, y_gate, dsep):
                    # the gates stay in the first layer's output (no concatenation pass, their gradient lands in split_cols' tensor)
                    return mlp_bank_select(list(self.towers), ops.MoeMix.apply(ex, dsep, ex.shape[1], y_gate), domain_id)
                y = torch.cat([ex, y_gate], dim=1)
        else:
            ex = torch.cat([m(embed_x) for m in experts], dim=1)
            y = torch.cat([ex, mlp_bank_forward(gates, embed_x, shared_input=True)], dim=1)
        H_ = experts[0].out_dim
        t0 = self.towers[0]
        if (not self.training and not torch.is_grad_enabled() and y.is_cuda and os.environ.get("SWR_ROUTED_EVAL", "1") != "0"
                and t0.n_blocks == 1 and t0.act == "relu" and t0.has_output_layer and t0.dropout_p == 0
                and ops.routed_mmoe_eval_supported(ne, H_, D, t0.block(0)[0].out_features)):
            # inference: a row needs only its own domain's gate mix and tower (BatchNorm is a fixed affine in eval mode)
            blocks = [t.block(0) for t in self.towers]
            outs = [t.output_linear() for t in self.towers]
            return ops.routed_mmoe_eval(y, ne, H_, [b[0].weight for b in blocks], [b[0].bias for b in blocks],
                                        _bn_dict([b[1] for b in blocks]), [o.weight for o in outs], [o.bias for o in outs],
                                        domain_id)
        desc = ops.make_mix_desc(D, ne, H_, 0, ne * H_, ne, [list(range(ne))] * D)
        pooled = ops.MoeMix.apply(y, desc, y.shape[1])                           # [B, D*H]
        return mlp_bank_select(list(self.towers), pooled, domain_id)              # towers [B, D] -> sigmoid -> select
