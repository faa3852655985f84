"""STAR: star topology adaptive recommender (reference: `models/multi_domain/star.py:10-118`)."""
import os

import torch
import torch.nn as nn
from torch.nn import init
from torch.nn.parameter import Parameter

from ... import ops
from ...basic.activation import activation_layer
from ...basic.layers import MLP, EmbeddingLayer, _bn_dict
from ...basic.module import SwrModule


class Star(SwrModule):
    """Partitioned normalisation + factorised FCN per domain + auxiliary network.

    Per domain d (on the WHOLE batch): h = (g_s*g_d) * PN(e) + b_s + b_d;  for every layer l (the final
    1-wide one included): h = ReLU(BN_{d,l}(h @ (W_s,l * W_d,l) + b_s,l + b_d,l));
    out = sigmoid(select_d(h) + aux(e)) (`star.py:78-118`).  Weights are stored [in, out].

    Evaluation here: PN(e) does not depend on d and is computed once; the per-domain affine of the
    partitioned norm is folded into the first layer's (small) weights; layer l of ALL domains is one
    stacked (l = 0) or grouped (l > 0) f32-MFMA product whose epilogue yields the BatchNorm statistics."""

    def __init__(self, features, num_domains, fcn_dims, aux_dims):
        super().__init__()
        self.features = features
        self.input_dim = sum([fea.embed_dim for fea in features])
        self.layer_num = len(fcn_dims) + 1
        self.fcn_dim = [self.input_dim] + fcn_dims + [1]
        self.num_domains = num_domains
        self.aux_dims = aux_dims
        self.embedding = EmbeddingLayer(features)

        self.dn_share_gamma = Parameter(torch.ones(self.input_dim))
        self.dn_share_bias = Parameter(torch.zeros(self.input_dim))
        self.eps = 1e-6

        self.auxnet = MLP(self.input_dim, dims=self.aux_dims)
        self.relu = activation_layer("relu")
        self.sig = activation_layer("sigmoid")

        self.share_parm_w = nn.ParameterList()
        self.share_parm_b = nn.ParameterList()
        for i in range(self.layer_num):
            self.share_parm_w.append(Parameter(torch.empty((self.fcn_dim[i], self.fcn_dim[i + 1])), requires_grad=True))
            self.share_parm_b.append(Parameter(torch.empty(self.fcn_dim[i + 1]), requires_grad=True))

        self.domain_specific_dn_gamma = nn.ParameterList()
        self.domain_specific_dn_bias = nn.ParameterList()
        self.domain_specific_w = nn.ParameterList()
        self.domain_specific_b = nn.ParameterList()
        self.domain_specific_bn = nn.ModuleList()
        for d in range(self.num_domains):
            self.domain_specific_dn_gamma.append(Parameter(torch.ones(self.input_dim)))
            self.domain_specific_dn_bias.append(Parameter(torch.zeros(self.input_dim)))
            lay_weight, lay_bias, lay_bn = nn.ParameterList(), nn.ParameterList(), nn.ModuleList()
            for i in range(self.layer_num):
                lay_weight.append(Parameter(torch.empty((self.fcn_dim[i], self.fcn_dim[i + 1])), requires_grad=True))
                lay_bias.append(Parameter(torch.empty(self.fcn_dim[i + 1]), requires_grad=True))
                lay_bn.append(nn.BatchNorm1d(self.fcn_dim[i + 1]))
            self.domain_specific_w.append(lay_weight)
            self.domain_specific_b.append(lay_bias)
            self.domain_specific_bn.append(lay_bn)
        self.reset_parameters()

    def reset_parameters(self):
        """kaiming-uniform weights, U(0, 1) biases (`star.py:68-76`)."""
        with torch.no_grad():
            for i in range(len(self.share_parm_w)):
                init.kaiming_uniform_(self.share_parm_w[i])
                init.uniform_(self.share_parm_b[i], 0, 1)
            for d in range(len(self.domain_specific_w)):
                for i in range(len(self.domain_specific_w[d])):
                    init.kaiming_uniform_(self.domain_specific_w[d][i])
                    init.uniform_(self.domain_specific_b[d][i], 0, 1)

    def _fused_groups(self):
        D = self.num_domains
        g = []
        for l in range(self.layer_num):
            bns = [self.domain_specific_bn[d][l] for d in range(D)]
            g += [[b.weight for b in bns], [b.bias for b in bns], [b.running_mean for b in bns],
                  [b.running_var for b in bns], [b.num_batches_tracked for b in bns]]
        return g

    def _layer_params(self, l):
        """What ops.star_layer_weights takes for layer l (the first layer also carries the partitioned norm's affines)."""
        D, first = self.num_domains, l == 0
        params = [self.share_parm_w[l], self.share_parm_b[l]] + ([self.dn_share_gamma, self.dn_share_bias] if first else [])
        params += [self.domain_specific_w[d][l] for d in range(D)] + [self.domain_specific_b[d][l] for d in range(D)]
        if first:
            params += list(self.domain_specific_dn_gamma) + list(self.domain_specific_dn_bias)
        return params

    def _routed_eval(self, h, domain_id, aux_out):
        """Inference: the partitioned norm's statistics are whole-batch (computed above); everything after it is per row,
        so a row runs the FCN stack of its own domain only -- 1/D of the dense evaluation's products."""
        D = self.num_domains
        route = ops.DomainRouting(domain_id, D)
        hs = route.rows(h)
        xs = [route.segment(hs, d) for d in range(D)]
        effs = ops.star_stack_weights(D, [self._layer_params(l) for l in range(self.layer_num)])
        for l in range(self.layer_num):
            eff = effs[l]
            for d in range(D):
                if route.count(d):
                    xs[d] = ops.linear_bn_act(xs[d], [eff[d]], [eff[D + d]], bn=_bn_dict([self.domain_specific_bn[d][l]]),
                                              acts="relu", groups=1, training=False)
        sel = route.scatter([xs[d] if route.count(d) else None for d in range(D)])
        return torch.sigmoid(sel + aux_out.reshape(-1))

    def forward(self, x):
        domain_id = x["domain_indicator"]
        emb = self.embedding(x, self.features, squeeze_dim=True)
        aux_out = self.auxnet(emb)                                        # [B, 1]
        D = self.num_domains
        # partitioned norm, shared part (identical for every domain, star.py:95-98): biased variance, eps 1e-6
        h = ops.batch_standardize(emb, self.eps)
        if not self.training and D <= 8 and ops.routed_eval_ok(h):
            return self._routed_eval(h, domain_id, aux_out)
        fused = D <= 8 and os.environ.get("SWR_STAR_FUSED", "1") != "0"
        if fused:
            # effective weights of EVERY layer for all domains before the first product, every parameter gradient after the
            # last one: two launches each way for the whole stack (csrc/star.hip, ops.StarStackWeights)
            effs = ops.star_stack_weights(D, [self._layer_params(l) for l in range(self.layer_num)])
        for l in range(self.layer_num):
            if fused:
                eff = effs[l]
                bns = [self.domain_specific_bn[d][l] for d in range(D)]
                h = ops.linear_bn_act(h, list(eff[:D]), list(eff[D:]), bn=_bn_dict(bns), acts="relu",
                                      groups=(1 if l == 0 else D), training=self.training)
                continue
            ws, bs = [], []
            for d in range(D):
                w = self.share_parm_w[l] * self.domain_specific_w[d][l]            # [in, out]
                b = self.share_parm_b[l] + self.domain_specific_b[d][l]
                if l == 0:
                    # fold the domain affine of the partitioned norm into the layer (star.py:99-100)
                    a = self.dn_share_gamma * self.domain_specific_dn_gamma[d]
                    c = self.dn_share_bias + self.domain_specific_dn_bias[d]
                    b = b + c @ w
                    w = a.unsqueeze(1) * w
                ws.append(w.t().contiguous())                                       # Linear layout [out, in]
                bs.append(b)
            bns = [self.domain_specific_bn[d][l] for d in range(D)]
            h = ops.linear_bn_act(h, ws, bs, bn=_bn_dict(bns), acts="relu", groups=(1 if l == 0 else D),
                                  training=self.training)                           # [B, D * out_l]
        return ops.domain_select(h, domain_id, apply_sigmoid=False, extra=aux_out)
