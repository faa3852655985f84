"""EPNet (reference: `models/multi_domain/epnet.py:6-33`)."""
import torch.nn as nn

from ... import ops
from ...basic.layers import MLP, EmbeddingLayer, GateNU, LayerBank, fused_lookup
from ...basic.module import SwrModule


class EPNet(SwrModule):
    """sigmoid(Linear(agn_e * GateNU(cat(sce_e, stopgrad(agn_e))))).

    `MLP(self.agn_dims, fcn_dims)` binds `fcn_dims` to MLP's 2nd positional parameter `output_layer`
    (`epnet.py:21`, `basic/layers.py:248`): the "MLP" is ONE Linear(agn_dims, 1) at `mlp.mlp.0` and
    `fcn_dims` is otherwise unused.  Kept as is -- checkpoints must interchange."""

    def __init__(self, sce_features, agn_features, fcn_dims):
        super().__init__()
        self.sce_features = sce_features
        self.agn_features = agn_features
        self.sce_embedding = EmbeddingLayer(sce_features)
        self.agn_embedding = EmbeddingLayer(agn_features)
        self.sce_dims = sum([fea.embed_dim for fea in sce_features])
        self.agn_dims = sum([fea.embed_dim for fea in agn_features])
        self.dims = self.sce_dims + self.agn_dims
        self.gatenu = GateNU(self.dims, self.agn_dims)
        self.mlp = MLP(self.agn_dims, fcn_dims)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        # one fused lookup for both feature groups: [sce_e | agn_e]
        both = fused_lookup(x, [(self.sce_embedding, self.sce_features), (self.agn_embedding, self.agn_features)])
        agn_x = both[:, self.sce_dims:]
        gate_in = ops.StopGradCols.apply(both, self.sce_dims, self.dims)        # cat(sce_x, agn_x.detach())
        gated = ops.mul_sigmoid(agn_x, self.gatenu.logits(gate_in), self.gatenu.gemma)      # agn_x * (gamma * sigmoid(.))
        out = LayerBank([self.mlp.mlp[0]], None, ["sigmoid"])(gated, self.training)
        return out.squeeze()
