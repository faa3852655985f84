"""Shared-Bottom (reference: `models/multi_domain/sharebottom.py:6-50`)."""
import torch.nn as nn

from ... import ops
from ...basic.layers import MLP, EmbeddingLayer, mlp_bank_forward, mlp_bank_groups
from ...basic.module import SwrModule


class SharedBottom(SwrModule):
    """emb -> bottom MLP -> one tower per domain on the WHOLE batch -> sigmoid -> domain select.

    Args as the reference: features, domain_num, bottom_params {"dims", "activation", "dropout"},
    tower_params (same keys)."""

    def __init__(self, features, domain_num, bottom_params, tower_params):
        super().__init__()
        self.features = features
        self.embedding = EmbeddingLayer(features)
        self.bottom_dims = sum([fea.embed_dim for fea in features])
        self.domain_num = domain_num
        self.bottom_mlp = MLP(self.bottom_dims, **{**bottom_params, **{"output_layer": False}})
        self.towers = nn.ModuleList(MLP(bottom_params["dims"][-1], **tower_params) for i in range(self.domain_num))

    def _fused_groups(self):
        return mlp_bank_groups(list(self.towers))

    def forward(self, x):
        domain_id = x["domain_indicator"]
        # (the bottom MLP's first layer is the only reader of the lookup: ops.OneHotInfo)
        h = self.bottom_mlp(self.embedding(x, self.features, squeeze_dim=True, onehot=self.bottom_mlp.n_blocks > 0))
        if not self.training and ops.routed_eval_ok(h):
            # inference: every row through its own domain's tower only (BatchNorm is a fixed affine in eval mode)
            route = ops.DomainRouting(domain_id, self.domain_num)
            hs = route.rows(h)
            return ops.routed_probs(route, [self.towers[d](route.segment(hs, d)) if route.count(d) else None
                                            for d in range(self.domain_num)])
        # all towers read the same h: their first layers are one stacked product, the rest grouped
        logits = mlp_bank_forward(list(self.towers), h, shared_input=True)          # [B, D]
        return ops.domain_select(logits, domain_id, apply_sigmoid=True)
