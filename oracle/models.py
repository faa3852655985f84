"""Model restatements (oracle only): SharedBottom, MMOE, PLE, Star, PPNet,
EPNet, HamurSmall, HamurLarge -- SURVEY.md Appendix A.

Each `*_forward(ctx, x, **hyper)` follows the cited reference forward line by
line in meaning (not in code): every domain branch runs on the whole batch and
the domain is selected at the very end, so in training mode the BatchNorm
statistics of each branch see every row.
Returns the probabilities `[B]` as a tape value.
"""
import numpy as np

from . import tape as T
from .nn import Ctx, Sparse, batchnorm, embedding_layer, feature_dim, gate_nu, mlp


def _squeeze1(v):
    return T.reshape(v, (v.v.shape[0],))


def sharedbottom_forward(ctx, x, features, domain_num, bottom_params, tower_params):
    """`SharedBottom.forward` (`models/multi_domain/sharebottom.py:28-50`)."""
    dom = x["domain_indicator"]
    e = embedding_layer(ctx, "embedding", x, features)
    h = mlp(ctx, "bottom_mlp", e, bottom_params["dims"], output_layer=False,
            activation=bottom_params.get("activation", "relu"))
    ys = [T.sigmoid(mlp(ctx, f"towers.{d}", h, tower_params["dims"], True,
                        tower_params.get("activation", "relu"))) for d in range(domain_num)]
    return _squeeze1(T.select_domain(ys, dom))


def _mix(gate, experts):
    """sum_j gate[:, j] * experts[j]  (`mmoe.py:48-49`, `ple.py:121-126`)."""
    out = None
    for j, xj in enumerate(experts):
        term = T.slice1(gate, j, j + 1) * xj
        out = term if out is None else out + term
    return out


def mmoe_forward(ctx, x, features, domain_num, n_expert, expert_params, tower_params):
    """`MMOE.forward` (`models/multi_domain/mmoe.py:33-56`); gates are
    MLP(dims=[n_expert], activation='softmax') = Linear -> BN -> Softmax
    (`mmoe.py:26-30`)."""
    dom = x["domain_indicator"]
    e = embedding_layer(ctx, "embedding", x, features)
    experts = [mlp(ctx, f"experts.{j}", e, expert_params["dims"], False,
                   expert_params.get("activation", "relu")) for j in range(n_expert)]
    gates = [mlp(ctx, f"gates.{d}", e, [n_expert], False, "softmax") for d in range(domain_num)]
    ys = []
    for d in range(domain_num):
        pooled = _mix(gates[d], experts)
        ys.append(T.sigmoid(mlp(ctx, f"towers.{d}", pooled, tower_params["dims"], True,
                                tower_params.get("activation", "relu"))))
    return _squeeze1(T.select_domain(ys, dom))


def ple_forward(ctx, x, features, domain_num, n_level, n_expert_specific, n_expert_shared,
                expert_params, tower_params):
    """`PLE.forward` + `CGC.forward` (`models/multi_domain/ple.py:41-64,107-136`)."""
    dom = x["domain_indicator"]
    e = embedding_layer(ctx, "embedding", x, features)
    inputs = [e] * (domain_num + 1)
    ea = expert_params.get("activation", "relu")
    for lvl in range(n_level):
        pre = f"cgc_layers.{lvl}"
        spec = []
        for d in range(domain_num):
            for i in range(n_expert_specific):
                k = d * n_expert_specific + i
                spec.append(mlp(ctx, f"{pre}.experts_specific.{k}", inputs[d], expert_params["dims"], False, ea))
        shared = [mlp(ctx, f"{pre}.experts_shared.{i}", inputs[-1], expert_params["dims"], False, ea)
                  for i in range(n_expert_shared)]
        outs = []
        for d in range(domain_num):
            g = mlp(ctx, f"{pre}.gates_specific.{d}", inputs[d], [n_expert_specific + n_expert_shared],
                    False, "softmax")
            outs.append(_mix(g, spec[d * n_expert_specific:(d + 1) * n_expert_specific] + shared))
        if lvl + 1 < n_level:
            g = mlp(ctx, f"{pre}.gate_shared", inputs[-1],
                    [n_expert_specific * domain_num + n_expert_shared], False, "softmax")
            outs.append(_mix(g, spec + shared))
        inputs = outs
    ys = [T.sigmoid(mlp(ctx, f"towers.{d}", inputs[d], tower_params["dims"], True,
                        tower_params.get("activation", "relu"))) for d in range(domain_num)]
    return _squeeze1(T.select_domain(ys, dom))


def star_forward(ctx, x, features, num_domains, fcn_dims, aux_dims):
    """`Star.forward` (`models/multi_domain/star.py:78-118`): partitioned norm
    (whole-batch mean / biased var, eps 1e-6) with gamma_s*gamma_d, beta_s+beta_d;
    layers h @ (W_s * W_d) + b_s + b_d -> BN_{d,l} -> ReLU including the final
    1-wide layer; out = sigmoid(select(h) + aux(e)).  Weights are [in, out]."""
    dom = x["domain_indicator"]
    e = embedding_layer(ctx, "embedding", x, features)
    aux = mlp(ctx, "auxnet", e, aux_dims, True, "relu")
    n_layer = len(fcn_dims) + 1
    outs = []
    for d in range(num_domains):
        mean = T.mean0(e)
        cen = e - mean
        var = T.mean0(cen * cen)
        h = cen / T.sqrt(var + 1e-6)
        h = (ctx.p("dn_share_gamma") * ctx.p(f"domain_specific_dn_gamma.{d}")) * h \
            + ctx.p("dn_share_bias") + ctx.p(f"domain_specific_dn_bias.{d}")
        for l in range(n_layer):
            w = ctx.p(f"share_parm_w.{l}") * ctx.p(f"domain_specific_w.{d}.{l}")
            b = ctx.p(f"share_parm_b.{l}") + ctx.p(f"domain_specific_b.{d}.{l}")
            h = T.matmul(h, w) + b
            h = T.relu(batchnorm(ctx, f"domain_specific_bn.{d}.{l}", h))
        outs.append(h)
    return _squeeze1(T.sigmoid(T.select_domain(outs, dom) + aux))


def ppnet_forward(ctx, x, id_features, agn_features, domain_num, fcn_dims):
    """`PPNet.forward` + `PPTowerBlock.forward` (`models/multi_domain/ppnet.py:47-67,21-29`).
    The tower input is the gate input cat(id_e, stopgrad(agn_e)); `agn_emb` is
    unused, so the agn tables receive no gradient at all."""
    dom = x["domain_indicator"]
    id_x = embedding_layer(ctx, "id_embedding", x, id_features)
    agn_x = embedding_layer(ctx, "agn_embedding", x, agn_features)
    g_in = T.cat1([id_x, T.detach(agn_x)])
    outs = []
    for d in range(domain_num):
        h = g_in
        for l in range(len(fcn_dims)):
            gate = gate_nu(ctx, f"domain_tower.{d}.gate_layers.{l}", g_in)
            h = mlp(ctx, f"domain_tower.{d}.mlp_layers.{l}", h, [fcn_dims[l]], False, "relu") * gate
        h = T.linear(h, ctx.p(f"domain_tower.{d}.final_layer.weight"), ctx.p(f"domain_tower.{d}.final_layer.bias"))
        outs.append(T.sigmoid(h))
    return _squeeze1(T.select_domain(outs, dom))


def epnet_forward(ctx, x, sce_features, agn_features, fcn_dims):
    """`EPNet.forward` (`models/multi_domain/epnet.py:25-33`).  `MLP(agn_dims,
    fcn_dims)` binds fcn_dims to `output_layer`, so the MLP is ONE
    Linear(agn_dims, 1) at `mlp.mlp.0` (`epnet.py:21`, `layers.py:248`)."""
    sce_x = embedding_layer(ctx, "sce_embedding", x, sce_features)
    agn_x = embedding_layer(ctx, "agn_embedding", x, agn_features)
    gate = gate_nu(ctx, "gatenu", T.cat1([sce_x, T.detach(agn_x)]))
    out = T.linear(agn_x * gate, ctx.p("mlp.mlp.0.weight"), ctx.p("mlp.mlp.0.bias"))
    return _squeeze1(T.sigmoid(out))


def _hyper_net(ctx, e, n_hyper):
    h = e
    for i in range(n_hyper):
        h = T.linear(h, ctx.p(f"hyper_net.{4 * i}.weight"), ctx.p(f"hyper_net.{4 * i}.bias"))
        h = T.relu(batchnorm(ctx, f"hyper_net.{4 * i + 1}", h))
    return h


def _adapter(ctx, h, H, iu, gamma, bias):
    """Adapter cell (`hamur.py:344-367` / `175-198`): down-projection with the
    per-sample matrix U0 H_b V0, sigmoid, up-projection with U1 H_b V1, domain
    norm over the batch (UNBIASED variance, eps 1e-5), residual."""
    w1 = T.einsum("mi,bij,jn->bmn", ctx.p(f"u.{iu}"), H, ctx.p(f"v.{iu}"))
    t = T.einsum("bf,bfj->bj", h, w1) + ctx.p(f"b_list.{iu}")
    t = T.sigmoid(t)
    w2 = T.einsum("mi,bij,jn->bmn", ctx.p(f"u.{iu + 1}"), H, ctx.p(f"v.{iu + 1}"))
    t = T.einsum("bf,bfj->bj", t, w2) + ctx.p(f"b_list.{iu + 1}")
    n = t.v.shape[0]
    mean = T.mean0(t)
    cen = t - mean
    var = T.sum_axis(cen * cen, 0) / float(n - 1)
    return ctx.p(gamma) * (cen / T.sqrt(var + 1e-5)) + ctx.p(bias) + h


def hamur_forward(ctx, x, features, domain_num, fcn_dims, hyper_dims, k, large=False):
    """`HamurSmall.forward` (`models/multi_domain/hamur.py:308-378`) and
    `HamurLarge.forward` (`101-244`).  The shared hyper-net is evaluated inside
    the domain loop: in training its BN running stats take D momentum updates
    per forward and its parameter gradients are the sum over the D uses.
    `hyper_dims` here is the caller's list WITHOUT the k*k the reference
    appends in place (`hamur.py:77,288`)."""
    dom = x["domain_indicator"]
    e = embedding_layer(ctx, "embedding", x, features)
    n_blocks = 7 if large else 2
    outs = []
    for d in range(domain_num):
        H = T.reshape(_hyper_net(ctx, e, len(hyper_dims) + 1), (-1, k, k))
        h = e
        for blk in range(n_blocks):
            pre = f"layer_list.{d}"
            h = T.linear(h, ctx.p(f"{pre}.{3 * blk}.weight"), ctx.p(f"{pre}.{3 * blk}.bias"))
            h = T.relu(batchnorm(ctx, f"{pre}.{3 * blk + 1}", h))
            if large and blk == 5:
                h = _adapter(ctx, h, H, 0, "gamma1", "bias1")
            elif large and blk == 6:
                h = _adapter(ctx, h, H, 2, "gamma2", "bias2")
            elif not large and blk == 1:
                h = _adapter(ctx, h, H, 0, "gamma1", "bias1")
        last = 3 * n_blocks
        h = T.linear(h, ctx.p(f"layer_list.{d}.{last}.weight"), ctx.p(f"layer_list.{d}.{last}.bias"))
        outs.append(T.sigmoid(h))
    return _squeeze1(T.select_domain(outs, dom))


def _mlp_n(ctx, prefix, x, n_blocks=1):
    """`Mlp_N` (`m3oe.py:45-67`): [Linear, LayerNorm(eps 1e-5), ReLU] per block at indices 3i..3i+2 of `.domain_specific`."""
    for i in range(n_blocks):
        x = T.linear(x, ctx.p(f"{prefix}.domain_specific.{3 * i}.weight"), ctx.p(f"{prefix}.domain_specific.{3 * i}.bias"))
        x = T.relu(T.layernorm(x, ctx.p(f"{prefix}.domain_specific.{3 * i + 1}.weight"),
                               ctx.p(f"{prefix}.domain_specific.{3 * i + 1}.bias"), 1e-5))
    return x


def m3oe_forward(ctx, x, features, domain_num, fcn_dims, expert_num, exp_d=1, exp_t=1, bal_d=1, bal_t=1, **_unused):
    """`M3oE.forward` (`models/multi_domain/m3oe.py:131-198`).  STAR front: per domain `e @ (W_slot[d] * W_shared) +
    b_slot[d] + b_shared`, rows keep their own domain's result (chain of `where` from zeros, 141-146); star_mlp + skip;
    gates = softmax(Linear(emb.detach())) (150-151); shared experts gate-mixed by bmm (187); domain experts balanced
    with the sigmoid scalars `_weight_bal_d`, `_weight_exp_d` (172-178, 187-189); towers Linear-LayerNorm-ReLU-Linear;
    sigmoid; domain select.  The same in train and eval mode (no BatchNorm)."""
    dom = x["domain_indicator"]
    D, ne = domain_num, expert_num
    n_body = len(fcn_dims) - 3                      # blocks of every expert: fcn_dim[3:] of [input_dim] + fcn_dims
    e = embedding_layer(ctx, "embedding", x, features)
    skip = _mlp_n(ctx, "skip_conn", e)
    outs = []
    for d in range(D):
        w = ctx.p(f"slot_weight.{d}") * ctx.p("shared_weight")
        outs.append(T.matmul(e, w) + ctx.p(f"slot_bias.{d}") + ctx.p("shared_bias"))
    emb = _mlp_n(ctx, "star_mlp", T.select_domain(outs, dom)) + skip
    emb_d = T.detach(emb)
    gates = [T.softmax_rows(T.linear(emb_d, ctx.p(f"gate.{d}.0.weight"), ctx.p(f"gate.{d}.0.bias"))) for d in range(D)]
    shared = [_mlp_n(ctx, f"expert.{j}", emb, n_body) for j in range(ne)]
    domexp = [_mlp_n(ctx, f"domain_expert.{d}", emb, n_body) for d in range(D)]
    wd = T.sigmoid(ctx.p("_weight_bal_d.deep_weights"))
    we = T.sigmoid(ctx.p("_weight_exp_d.deep_weights"))
    ys = []
    for i in range(D):
        bal = wd * domexp[i]
        for j in range(D):
            if j != i:
                bal = bal + (1 - wd) / (D - 1) * domexp[j]
        fused = _mix(gates[i], shared) + we * bal
        t = T.linear(fused, ctx.p(f"tower.{i}.0.weight"), ctx.p(f"tower.{i}.0.bias"))
        t = T.relu(T.layernorm(t, ctx.p(f"tower.{i}.1.weight"), ctx.p(f"tower.{i}.1.bias"), 1e-5))
        ys.append(T.sigmoid(T.linear(t, ctx.p(f"tower.{i}.3.weight"), ctx.p(f"tower.{i}.3.bias"))))
    return _squeeze1(T.select_domain(ys, dom))


FORWARDS = {
    "SharedBottom": sharedbottom_forward,
    "MMOE": mmoe_forward,
    "PLE": ple_forward,
    "Star": star_forward,
    "PPNet": ppnet_forward,
    "EPNet": epnet_forward,
    "M3oE": m3oe_forward,
    "HamurSmall": lambda ctx, x, **kw: hamur_forward(ctx, x, large=False, **kw),
    "HamurLarge": lambda ctx, x, **kw: hamur_forward(ctx, x, large=True, **kw),
}


class OracleModel:
    """A model family + hyper-parameters bound to a reference-keyed state dict."""

    def __init__(self, family, hyper, state, dtype=np.float32):
        self.family = family
        self.hyper = hyper
        self.state = {k: np.array(v) for k, v in state.items()}
        self.dtype = dtype

    def forward(self, x, training):
        """-> (ctx, probs Var[B])."""
        ctx = Ctx(self.state, training, self.dtype)
        return ctx, FORWARDS[self.family](ctx, x, **self.hyper)

    def predict(self, x):
        return self.forward(x, training=False)[1].v

    def loss_and_grads(self, x, y):
        """One training forward + BCE + backward (`ctr_trainer.py:69-72`).
        -> (probs, loss, grads-by-name).  BN buffers in `self.state` advance."""
        ctx, p = self.forward(x, training=True)
        loss = T.bce_mean(p, np.asarray(y, dtype=np.float32))
        T.backward(loss)
        return p.v, float(loss.v), ctx.grads()


def input_dims(features):
    return sum(feature_dim(f) for f in features)
