"""M3oE (reference: `models/multi_domain/m3oe.py:8-198`): STAR front + MMoE body + per-domain experts, LayerNorm blocks.

Module tree, parameter names, creation order (= RNG order) and constructor signature are the reference's.  Unlike
the BatchNorm families nothing couples the rows of a batch (LayerNorm is per row), so the forward is the same in train
and eval mode.  Evaluation on the HIP path:

  * one lookup (K1); `skip_conn` Linear and the D factorised STAR products `e @ (W_slot[d] (.) W_shared) + b_slot[d] +
    b_shared` (`m3oe.py:141-146`) as stacked products, the effective weights by `csrc/star.hip`, the row's own domain
    block by `swr_block_select` (rows with an id outside [0, D) stay zero, like the reference's chain of `where`);
  * all `expert_num` shared experts and `domain_num` domain experts read the same `emb`: one stacked product + one
    LayerNorm+ReLU launch over the G = expert_num + domain_num column groups (`csrc/layernorm.hip`);
  * gates = softmax(Linear(emb.detach())) (no BatchNorm here, `m3oe.py:117-119,150-151`), gate mix by `swr_moe_mix`;
  * the balance between domain experts (`m3oe.py:172-178`) is a [D*H, D*H] Kronecker weight built from the two scalar
    parameters (parameter-sized torch ops) and applied as one product;
  * towers [Linear, LayerNorm, ReLU, Linear(., 1)] grouped over the domains, domain select (+ BCE) fused.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ...basic.layers import EmbeddingLayer, LayerBank
from ...basic.module import SwrModule


class Weights(torch.nn.Module):
    """`m3oe.py:8-42`: a learnable scalar passed through a sigmoid (softmax_type 3, the only live branch); every call
    also anneals `tau` on the host, as the reference does."""

    def __init__(self, weight_shape, tau, tau_step, initial_deep, softmax_type=2):
        super().__init__()
        assert isinstance(weight_shape, (int, list))
        norm = weight_shape[-1] if isinstance(weight_shape, list) else weight_shape
        if initial_deep is None:
            initial_deep = np.ones(weight_shape, dtype=np.float32) / norm
            print(f'initial_deep: {initial_deep}')
        else:
            initial_deep = np.ones(weight_shape, dtype=np.float32) * initial_deep
        self.deep_weights = torch.nn.Parameter(torch.from_numpy(initial_deep), requires_grad=True)
        self.softmax_type = softmax_type
        self.tau = tau
        self.tau_step = tau_step

    def forward(self):
        if self.tau > 0.01:
            self.tau -= self.tau_step
        assert self.softmax_type == 3, "only softmax_type 3 (sigmoid) is reachable in the reference (m3oe.py:27-42)"
        return torch.sigmoid(self.deep_weights)


class Mlp_N(nn.Module):
    """[Linear, LayerNorm, ReLU] per consecutive pair of `fcn_dim` (`m3oe.py:45-68`); a parameter holder here."""

    def __init__(self, fcn_dim):
        super().__init__()
        self.fcn_dim = fcn_dim
        self.n = len(fcn_dim)
        self.domain_specific = nn.ModuleList()
        for i in range(self.n - 1):
            self.domain_specific.append(nn.Linear(self.fcn_dim[i], self.fcn_dim[i + 1]))
            self.domain_specific.append(nn.LayerNorm(self.fcn_dim[i + 1]))
            self.domain_specific.append(nn.ReLU())

    def blocks(self):
        return [(self.domain_specific[3 * i], self.domain_specific[3 * i + 1]) for i in range(self.n - 1)]


def mlp_n_bank(mlps, x, shared_input):
    """Structurally identical Mlp_N stacks evaluated together: block 0 on a shared x (stacked outputs) or on per-member
    column slices, deeper blocks grouped.  Returns [M, len(mlps) * out_dim]."""
    for i in range(mlps[0].n - 1):
        lins = [m.blocks()[i][0] for m in mlps]
        norms = [m.blocks()[i][1] for m in mlps]
        z = LayerBank(lins, grouped=not (shared_input and i == 0))(x, False)
        x = ops.layer_norm_act(z, norms, relu=True)
    return x


class M3oE(SwrModule):
    def __init__(self, features, domain_num, fcn_dims, expert_num, exp_d, exp_t, bal_d, bal_t, tau=1, task_num=1,
                 tau_step=0.00005, softmax_type=3, device="cpu"):
        super().__init__()
        self.features = features
        self.input_dim = sum([fea.embed_dim for fea in features])
        self.layer_num = len(fcn_dims) + 1
        self.fcn_dim = [self.input_dim] + fcn_dims
        self.domain_num = domain_num
        self.task_num = task_num
        self.expert_num = expert_num
        self.embedding = EmbeddingLayer(features)
        self.device = device
        self._weight_exp_d = Weights(1, tau, tau_step, exp_d, softmax_type)
        self._weight_exp_t = Weights(1, tau, tau_step, exp_t, softmax_type)
        self._weight_bal_d = Weights(1, tau, tau_step, bal_d, softmax_type)
        self._weight_bal_t = Weights(1, tau, tau_step, bal_t, softmax_type)
        assert len(self.fcn_dim) > 3, f'too few layers assigned, must larger than 3. Star owns 3 layers, mmoe owns the rest.'
        self.star_dim = self.fcn_dim[:3]
        self.fcn_dim = self.fcn_dim[3:]
        self.skip_conn = Mlp_N([self.star_dim[0], self.star_dim[2]])
        self.shared_weight = nn.Parameter(torch.empty(self.star_dim[0], self.star_dim[1]))
        self.shared_bias = nn.Parameter(torch.zeros(self.star_dim[1]))
        self.slot_weight = nn.ParameterList(
            [nn.Parameter(torch.empty(self.star_dim[0], self.star_dim[1])) for i in range(self.domain_num)])
        self.slot_bias = nn.ParameterList([nn.Parameter(torch.zeros(self.star_dim[1])) for i in range(self.domain_num)])
        self.star_mlp = Mlp_N([self.star_dim[1], self.star_dim[2]])
        torch.nn.init.xavier_uniform_(self.shared_weight.data)
        for m in self.slot_weight:
            torch.nn.init.xavier_uniform_(m.data)
        self.expert = nn.ModuleList()
        for d in range(expert_num):
            self.expert.append(Mlp_N(self.fcn_dim))
        self.domain_expert = nn.ModuleList()
        for d in range(domain_num):
            self.domain_expert.append(Mlp_N(self.fcn_dim))
        self.gate = torch.nn.ModuleList(
            [torch.nn.Sequential(torch.nn.Linear(self.fcn_dim[0], expert_num), torch.nn.Softmax(dim=1)) for i in
             range(domain_num)])
        self.tower = nn.ModuleList()
        for d in range(domain_num):
            self.tower.append(nn.Sequential(nn.Linear(self.fcn_dim[-1], self.fcn_dim[-1]), nn.LayerNorm(self.fcn_dim[-1]),
                                            nn.ReLU(), nn.Linear(self.fcn_dim[-1], 1)))

    def _fused_groups(self):
        everyone = list(self.expert) + list(self.domain_expert)
        g = []
        for i in range(everyone[0].n - 1):
            lins = [m.blocks()[i][0] for m in everyone]
            norms = [m.blocks()[i][1] for m in everyone]
            g += LayerBank(lins).tensor_groups() + [[n.weight for n in norms], [n.bias for n in norms]]
        g += LayerBank([s[0] for s in self.gate]).tensor_groups()
        g += LayerBank([t[0] for t in self.tower]).tensor_groups()
        g += [[t[1].weight for t in self.tower], [t[1].bias for t in self.tower]]
        g += LayerBank([t[3] for t in self.tower]).tensor_groups()
        return g

    def forward(self, x, test_flag=False):
        domain_id = x["domain_indicator"]
        D, ne = self.domain_num, self.expert_num
        H_ = self.fcn_dim[-1]
        e = self.embedding(x, self.features, squeeze_dim=True)
        skip = mlp_n_bank([self.skip_conn], e, True)                                       # [B, star2]
        # STAR front: the D factorised products on the whole batch, then each row keeps its own domain's block
        eff = ops.star_layer_weights(False, D, self.shared_weight, self.shared_bias, *self.slot_weight, *self.slot_bias)
        z = ops.linear_bn_act(e, list(eff[:D]), list(eff[D:]), bn=None, acts=None, groups=1, training=False)   # [B, D*star1]
        emb = ops.block_select(z, domain_id, D, self.star_dim[1])
        emb = ops.add(mlp_n_bank([self.star_mlp], emb, True), skip)
        # shared experts + domain experts on the same input, one stacked evaluation
        both = mlp_n_bank(list(self.expert) + list(self.domain_expert), emb, True)         # [B, (ne + D) * H]
        gates = LayerBank([s[0] for s in self.gate], None, [("softmax", ne)] * D)(emb.detach(), False)   # [B, D*ne]
        desc = ops.make_mix_desc(D, ne, H_, 0, ne * H_, ne, [list(range(ne)) for _ in range(D)])
        mixed = ops.MoeMix.apply(torch.cat([both[:, :ne * H_], gates], dim=1), desc, ne * H_ + D * ne)   # [B, D*H]
        # balance between the domain experts (`m3oe.py:172-178,187-189`): out_i = we * (wd dom_i + (1 - wd)/(D-1) sum_{j != i} dom_j)
        wd, we = self._weight_bal_d(), self._weight_exp_d()
        for _ in range(D * D - 1):
            self._weight_bal_d()                     # (the reference calls it D*D times per forward: tau annealing)
        for _ in range(D - 1):
            self._weight_exp_d()
        eye = torch.eye(D, device=e.device, dtype=torch.float32)
        # (D == 1: the reference's `j != i` loop is empty -- no off-diagonal term, and no division by D - 1 = 0)
        off = (1.0 - wd) / (D - 1) * (1.0 - eye) if D > 1 else torch.zeros_like(eye)
        M = we * (wd * eye + off)                                                           # [D, D], M[i, j]
        Wk = torch.kron(M, torch.eye(H_, device=e.device, dtype=torch.float32))            # [D*H, D*H] Linear layout [out, in]
        dom = both[:, ne * H_:]
        fused = ops.add(mixed, ops.linear_bn_act(dom, [Wk], None, bn=None, acts=None, groups=1, training=False))
        # towers: Linear(H, H) -> LayerNorm -> ReLU -> Linear(H, 1), grouped over the domains
        t = LayerBank([t[0] for t in self.tower], grouped=True)(fused, False)
        t = ops.layer_norm_act(t, [t_[1] for t_ in self.tower], relu=True)
        logits = LayerBank([t_[3] for t_ in self.tower], grouped=True)(t, False)           # [B, D]
        return ops.domain_select(logits, domain_id, apply_sigmoid=True)
