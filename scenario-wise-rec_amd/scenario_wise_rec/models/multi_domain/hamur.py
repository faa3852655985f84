"""HAMUR: hyper adapter for multi-domain recommendation (reference: `models/multi_domain/hamur.py`)."""
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from ... import ops
from ...basic.activation import activation_layer
from ...basic.layers import EmbeddingLayer, LayerBank, _bn_dict
from ...basic.module import SwrModule


class _Hamur(SwrModule):
    """Per-domain backbone [Linear, BN, ReLU] x n_blocks + Linear(., 1); one SHARED hyper-network
    emb -> ... -> k*k whose output, reshaped to a per-sample matrix H_b, parameterises adapter cells
    inserted after chosen backbone blocks (`hamur.py:101-244`, `308-378`).

    Fidelity notes (all reproduced): the reference evaluates the hyper-network INSIDE the domain loop, so
    in training its BatchNorm running statistics take D momentum updates per forward and
    `num_batches_tracked` grows by D; `hyper_dims += [k*k]` mutates the caller's list; U, V start as ones;
    the adapter bottleneck is hard-coded to 32.  Here the hyper-network runs once per forward (same
    output, gradients sum over its D uses) and the D-fold statistics update is applied in closed form;
    the adapter is applied in the re-associated form ((h U) H_b) V, which never materialises the
    per-sample [m, 32] weight."""

    n_blocks = 2
    adapter_after = (1,)          # block indices followed by an adapter cell

    def __init__(self, features, domain_num, fcn_dims, hyper_dims, k):
        super().__init__()
        self.features = features
        self.input_dim = sum([fea.embed_dim for fea in features])
        self.layer_num = len(fcn_dims) + 1
        self.fcn_dim = [self.input_dim] + fcn_dims
        self.domain_num = domain_num
        self.embedding = EmbeddingLayer(features)
        self.relu = activation_layer("relu")
        self.sig = activation_layer("sigmoid")

        self.layer_list = nn.ModuleList()
        for d in range(domain_num):
            ds = nn.ModuleList()
            for i in range(self.n_blocks):
                ds.append(nn.Linear(self.fcn_dim[i], self.fcn_dim[i + 1]))
                ds.append(nn.BatchNorm1d(self.fcn_dim[i + 1]))
                ds.append(nn.ReLU())
            ds.append(nn.Linear(self.fcn_dim[self.n_blocks], 1))
            self.layer_list.append(ds)

        self.k = k
        self.u = nn.ParameterList()
        self.v = nn.ParameterList()
        for blk in self.adapter_after:
            m = self.fcn_dim[blk + 1]
            self.u.append(Parameter(torch.ones((m, self.k)), requires_grad=True))
            self.u.append(Parameter(torch.ones((32, self.k)), requires_grad=True))
        for blk in self.adapter_after:
            m = self.fcn_dim[blk + 1]
            self.v.append(Parameter(torch.ones((self.k, 32)), requires_grad=True))
            self.v.append(Parameter(torch.ones((self.k, m)), requires_grad=True))

        hyper_dims += [self.k * self.k]                  # in place, like the reference (hamur.py:77,288)
        input_dim = self.input_dim
        hyper_layers = []
        for i_dim in hyper_dims:
            hyper_layers += [nn.Linear(input_dim, i_dim), nn.BatchNorm1d(i_dim), nn.ReLU(), nn.Dropout(p=0)]
            input_dim = i_dim
        self.hyper_net = nn.Sequential(*hyper_layers)
        self.n_hyper = len(hyper_dims)

        self.b_list = nn.ParameterList()
        for blk in self.adapter_after:
            self.b_list.append(Parameter(torch.zeros((32)), requires_grad=True))
            self.b_list.append(Parameter(torch.zeros((self.fcn_dim[blk + 1])), requires_grad=True))
        self.gamma1 = nn.Parameter(torch.ones(self.fcn_dim[self.adapter_after[0] + 1]))
        self.bias1 = nn.Parameter(torch.zeros(self.fcn_dim[self.adapter_after[0] + 1]))
        if len(self.adapter_after) > 1:
            self.gamma2 = nn.Parameter(torch.ones(self.fcn_dim[self.adapter_after[1] + 1]))
            self.bias2 = nn.Parameter(torch.zeros(self.fcn_dim[self.adapter_after[1] + 1]))
        self.eps = 1e-5

    def _fused_groups(self):
        g = []
        for i in range(self.n_blocks):
            g += LayerBank([ds[3 * i] for ds in self.layer_list], [ds[3 * i + 1] for ds in self.layer_list]).tensor_groups()
        g += LayerBank([ds[3 * self.n_blocks] for ds in self.layer_list]).tensor_groups()
        return g

    def _hyper(self, emb):
        """hyper_net(emb) once; in training the BN buffers advance as if it had run D times."""
        D = self.domain_num
        h = emb
        for i in range(self.n_hyper):
            lin, bn = self.hyper_net[4 * i], self.hyper_net[4 * i + 1]
            bnd = _bn_dict([bn])
            if self.training:
                bnd["momentum"] = 1.0 - (1.0 - bnd["momentum"]) ** D      # D momentum updates with the same batch stats
            h = ops.linear_bn_act(h, [lin.weight], [lin.bias], bn=bnd, acts="relu", training=self.training)
            if self.training and D > 1:
                bn.num_batches_tracked += D - 1
        return h.reshape(-1, self.k, self.k)

    def _adapter(self, h, Hm, iu, gamma, bias):
        """Adapter cell on [B, D*m] (all domains side by side), `hamur.py:175-198 / 344-367`."""
        D = self.domain_num
        B = h.shape[0]
        m = h.shape[1] // D
        hd = h.reshape(B, D, m)
        # down projection with W1_b = U0 H_b V0, applied as ((h U0) H_b) V0; the shared factors are plain products on
        # [B*D, .], the per-sample factor is swr_rowmat
        k = self.k
        t = ops.MatmulIO.apply(hd.reshape(B * D, m), self.u[iu], None, None)                     # [B*D, k]
        t = ops.RowMat.apply(t.reshape(B, D, k), Hm)
        t = ops.MatmulIO.apply(t.reshape(B * D, k), self.v[iu], self.b_list[iu], "sigmoid")   # [B*D, 32], sigmoid fused in
        t = ops.MatmulIO.apply(t, self.u[iu + 1], None, None)
        t = ops.RowMat.apply(t.reshape(B, D, k), Hm)
        t = ops.MatmulIO.apply(t.reshape(B * D, k), self.v[iu + 1], self.b_list[iu + 1], None)  # [B*D, m]
        # domain norm over the batch: UNBIASED variance, eps 1e-5.  With n rows,
        #   (t - mean) / sqrt(M2 / (n - 1) + eps) = sqrt((n - 1) / n) * (t - mean) / sqrt(M2 / n + eps (n - 1) / n),
        # i.e. the biased-variance kernel with a rescaled eps and gamma
        c = ((B - 1) / B) ** 0.5 if B > 1 else 0.0
        eps_b = self.eps * (B - 1) / B if B > 1 else self.eps
        out = ops.batch_standardize(t.reshape(B, D * m), eps_b, (gamma * c).repeat(D), bias.repeat(D)).reshape(B, D, m)
        return ops.add(out.reshape(B, D * m), h)

    def forward(self, x):
        domain_id = x["domain_indicator"]
        emb = self.embedding(x, self.features, squeeze_dim=True)
        D = self.domain_num
        Hm = self._hyper(emb)                                             # [B, k, k]
        h = emb
        ia = 0
        for i in range(self.n_blocks):
            bank = LayerBank([ds[3 * i] for ds in self.layer_list], [ds[3 * i + 1] for ds in self.layer_list],
                             ["relu"] * D, grouped=(i > 0))
            h = bank(h, self.training)                                    # [B, D * fcn_dim[i+1]]
            if i in self.adapter_after:
                g, b = (self.gamma1, self.bias1) if ia == 0 else (self.gamma2, self.bias2)
                h = self._adapter(h, Hm, 2 * ia, g, b)
                ia += 1
        logits = LayerBank([ds[3 * self.n_blocks] for ds in self.layer_list], None, [None] * D, grouped=True)(h, self.training)
        return ops.domain_select(logits, domain_id, apply_sigmoid=True)


class HamurSmall(_Hamur):
    """2-block backbone, one adapter cell after block 2 (`hamur.py:247-378`)."""
    n_blocks = 2
    adapter_after = (1,)


class HamurLarge(_Hamur):
    """7-block backbone, adapter cells after blocks 6 and 7 (`hamur.py:9-244`)."""
    n_blocks = 7
    adapter_after = (5, 6)
