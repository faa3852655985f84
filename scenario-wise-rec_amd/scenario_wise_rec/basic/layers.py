"""EmbeddingLayer, MLP, GateNU -- the layer toolkit of the hot path (reference: `basic/layers.py`).

Module trees, constructor signatures and state_dict keys are the reference's (nn.Linear /
nn.BatchNorm1d / nn.Embedding objects are kept as PARAMETER HOLDERS, created in the reference's order so
a given torch seed yields the same initial weights); `forward` never calls them -- it launches the fused
HIP ops of `scenario_wise_rec.ops`.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .activation import activation_layer, activation_name
from .features import DenseFeature, SequenceFeature, SparseFeature
from .module import SwrModule


class EmbeddingLayer(SwrModule):
    """All feature lookups of a model in one fused gather (`basic/layers.py:27-114`).

    `forward(x, features, squeeze_dim=True)` returns `[B, sum(embed_dim)]`: the sparse embeddings in
    feature-list order first, the dense columns last -- whatever the order of the list
    (`layers.py:98-105`).  The returned tensor may be a column view of a 16-byte-aligned wider buffer.
    """

    def __init__(self, features):
        super().__init__()
        self.features = features
        self.embed_dict = nn.ModuleDict()
        self.n_dense = 0
        for fea in features:
            if fea.name in self.embed_dict:
                continue
            if isinstance(fea, (SparseFeature, SequenceFeature)) and fea.shared_with is None:
                self.embed_dict[fea.name] = fea.get_embedding_layer()
            elif isinstance(fea, DenseFeature):
                self.n_dense += 1

    def forward(self, x, features, squeeze_dim=False, onehot=False):
        """`onehot=True` (training, squeeze_dim): the caller promises that the returned tensor is consumed by exactly ONE
        ops.linear_bn_act / LayerBank call -- the lookup may then append a one-hot block of its small tables and that
        layer's backward produces their gradients itself (ops.OneHotInfo)."""
        sparse, dense = [], []
        for fea in features:
            if isinstance(fea, SequenceFeature):
                if fea.pooling not in ("sum", "mean", "concat"):
                    raise ValueError("Sequence pooling method supports only pooling in %s, got %s." %
                                     (["sum", "mean"], fea.pooling))
                sparse.append(fea)               # pooled lookup (`layers.py:73-87`): a column block like any sparse feature
            elif isinstance(fea, SparseFeature):
                sparse.append(fea)
            else:
                dense.append(fea)
        if not sparse and not dense:
            raise ValueError("The input features can note be empty")
        if not squeeze_dim and not sparse:
            raise ValueError(
                "If keep the original shape:[batch_size, num_features, embed_dim], expected %s in feature list, got %s" %
                ("SparseFeatures", features))
        plan, weights = _new_plan(self)
        # (onehot="layout": the training layout without a backward to follow -- bench.py times the lookup alone that way)
        plan.onehot = (onehot if onehot == "layout" else bool(onehot)) if (squeeze_dim and self.training) else False
        _plan_part(plan, weights, {}, self, x, sparse, dense if squeeze_dim else [])
        out = _run_plan(plan, weights)
        if squeeze_dim:
            return out
        if any(b["mode"] == 2 for b in plan.bags):
            raise NotImplementedError("squeeze_dim=False with pooling='concat' ([B, F, L, E]) is not built")
        dims = {s[3] for s in plan.sparse} | {b["dim"] for b in plan.bags}
        if len(dims) != 1:
            raise RuntimeError("squeeze_dim=False needs equal embed_dim for all sparse features")
        return out.reshape(out.shape[0], len(plan.sparse) + len(plan.bags), dims.pop())


def _new_plan(layer):
    plan = ops._GatherPlan()
    plan.sparse, plan.dense, plan.width = [], [], 0
    plan.bags = []
    plan.onehot, plan.oh, plan.ctx, plan.fold, plan.wide, plan.fl = False, None, None, None, None, None
    plan.lazy = {}
    plan.dense_limit_bytes = layer.dense_table_limit_bytes
    return plan, []


def _plan_part(plan, weights, wpos, layer, x, sparse, dense):
    """Append one EmbeddingLayer's lookups: its sparse embeddings, then its dense columns."""
    col = plan.width
    for fea in sparse:
        owner = fea.shared_with if fea.shared_with is not None else fea.name
        key = (id(layer), owner)
        if key not in wpos:
            wpos[key] = len(weights)
            weights.append(layer.embed_dict[owner].weight)
            lazy = getattr(layer.embed_dict[owner].weight, "_swr_lazy", None)
            if lazy is not None:
                plan.lazy[wpos[key]] = lazy
        w = weights[wpos[key]]
        if isinstance(fea, SequenceFeature):
            idx = x[fea.name]
            if idx.dim() != 2:
                raise ValueError(f"sequence feature {fea.name!r}: expected ids of shape (batch_size, seq_len), got {tuple(idx.shape)}")
            mode = {"sum": 0, "mean": 1, "concat": 2}[fea.pooling]
            plan.bags.append({"wpos": wpos[key], "idx": idx, "vocab": w.shape[0], "dim": w.shape[1], "col": col,
                              "L": idx.shape[1], "mode": mode, "pad": fea.padding_idx, "seed": getattr(fea, "hash_seed", 0)})
            col += w.shape[1] * (idx.shape[1] if mode == 2 else 1)
            continue
        plan.sparse.append((wpos[key], x[fea.name], w.shape[0], w.shape[1], col, getattr(fea, "hash_seed", 0)))
        col += w.shape[1]
    for fea in dense:
        plan.dense.append((x[fea.name], col))
        col += 1
    plan.width = col


def _run_plan(plan, weights):
    plan.ld = (plan.width + 3) // 4 * 4
    plan.want_grad = torch.is_grad_enabled()       # Function.forward itself always runs with grad mode off
    out = ops.EmbedGather.apply(plan, *weights)
    if plan.oh:
        # hand the one-hot block to the consuming layer (ops.OneHotInfo): which tables it covers, and the compact column
        # layout of the dX that layer owes the remaining (K3) slots
        info = ops.OneHotInfo()
        info.ctx = plan.ctx
        info.oh_col = plan.fold["col0"] + plan.fold["Kp"] if plan.fold is not None else (plan.width + 3) // 4 * 4
        info.oh_width = plan.ld - info.oh_col
        info.tables_p = [(weights[wpos], vocab, dim, off, col) for _i, wpos, vocab, dim, off, col in plan.oh]
        info.params = tuple({id(t[0]): t[0] for t in info.tables_p}.values())
        info.tables = None
        cols, compact, pos = [], [], 0
        for _wpos, _idx, _vocab, dim, col, _seed in plan.sparse[:plan.ctx.n_k3_slots]:
            compact.append(pos)
            cols.extend(range(col, col + dim))
            pos += dim
        pad = (-pos) % 4
        cols.extend([cols[-1]] * pad if cols else [])              # (16-byte rows for the product; the pad columns are never read)
        info.compact, info.n_sel = compact, len(cols)
        info.sel = _sel_tensor(tuple(cols), out.device) if cols else None
        info.fold, info.wide, info.K = plan.fold is not None, plan.wide, plan.width
        info.fl = getattr(plan, "fl", None)
        if plan.fold is not None:
            f = plan.fold
            info.col0, info.Kp = f["col0"], f["Kp"]
            inv = [0] * plan.width
            for j, c in enumerate(f["src"]):
                if c >= 0:
                    inv[c] = j
            for t, (_i, _wpos, _vocab, dim, _off, col) in enumerate(plan.oh):
                for e in range(dim):
                    inv[col + e] = -1 - t
            ohtab = [-1] * info.oh_width
            for t, (_i, _wpos, vocab, _dim, off, _col) in enumerate(plan.oh):
                ohtab[off:off + vocab] = [t] * vocab
            info.ohtab = _sel_tensor(tuple(ohtab), out.device, torch.int32)
            info.src = _sel_tensor(f["src"], out.device, torch.int32)
            info.inv = _sel_tensor(tuple(inv), out.device, torch.int32)
        out._swr_onehot = info
    # columns that can take a gradient: everything up to the end of the last embedding column (dense-feature columns are
    # inputs).  A layer that reads this tensor need not compute d/dx beyond it (ops.LinearBNAct: `n_compute` of dX).
    # (tables looked up detached -- PPNet's agnostic group, ppnet.py:54 -- take none either: at config 6 only the first 128
    # of the 452 columns do, and the two first-layer dX products skip the rest)
    out._swr_grad_cols = max([col + dim for w_, _i, _v, dim, col, _s in plan.sparse if weights[w_].requires_grad] +
                             [b["col"] + b["dim"] * (b["L"] if b["mode"] == 2 else 1) for b in plan.bags
                              if weights[b["wpos"]].requires_grad], default=0)
    return out


_SEL_CACHE = {}


def _sel_tensor(cols, device, dtype=torch.int64):
    """Column index list on the device, cached: built once per (layout, device) -- an upload per step would be a host
    sync and cannot be captured into a hipGraph."""
    key = (cols, str(device), dtype)
    t = _SEL_CACHE.get(key)
    if t is None:
        t = _SEL_CACHE[key] = torch.tensor(cols, dtype=dtype, device=device)
    return t


def fused_lookup(x, parts):
    """`torch.cat([layer(x, feats, squeeze_dim=True) for layer, feats, ... in parts], dim=1)` as ONE gather
    launch (PPNet / EPNet look two feature groups up, ppnet.py:51-54, epnet.py:26-28).  A part given as
    `(layer, feats, True)` is looked up detached: its tables take no gradient at all (PPNet's agnostic
    group is only ever used through `.detach()`, ppnet.py:54)."""
    plan, weights = _new_plan(parts[0][0])
    wpos = {}
    for part in parts:
        layer, feats = part[0], part[1]
        n0 = len(weights)
        sparse = [f for f in feats if isinstance(f, (SparseFeature, SequenceFeature))]
        dense = [f for f in feats if not isinstance(f, (SparseFeature, SequenceFeature))]
        _plan_part(plan, weights, wpos, layer, x, sparse, dense)
        if len(part) > 2 and part[2]:
            weights[n0:] = [w.detach() for w in weights[n0:]]
    return _run_plan(plan, weights)


def _bn_dict(bns):
    return {"gamma": [b.weight for b in bns], "beta": [b.bias for b in bns],
            "running_mean": [b.running_mean for b in bns], "running_var": [b.running_var for b in bns],
            "nbt": [b.num_batches_tracked for b in bns], "eps": bns[0].eps,
            "momentum": bns[0].momentum if bns[0].momentum is not None else 0.1}


class LayerBank(object):
    """Several [Linear (-> BatchNorm1d) (-> activation)] layers evaluated by ONE set of launches.

    shared input  (grouped=False): every member reads the same x; outputs side by side.
    own inputs    (grouped=True) : member g reads x[:, g*K:(g+1)*K]; all members have the same shape.
    `acts[i]` is None / 'relu' / 'sigmoid' / ('softmax', width)."""

    def __init__(self, linears, bns=None, acts=None, grouped=False):
        self.linears, self.bns, self.grouped = list(linears), (list(bns) if bns is not None else None), grouped
        if acts is None:
            acts = [None] * len(self.linears)
        assert len(acts) == len(self.linears), "one activation entry per layer"
        self.ranges, col = [], 0
        for lin, a in zip(self.linears, acts):
            n = lin.out_features
            if isinstance(a, tuple):
                self.ranges.append((col, col + n, a[0], a[1]))
            else:
                self.ranges.append((col, col + n, a, n if a == "softmax" else 1))
            col += n
        self.width = col
        # merge equal neighbours: the kernel takes at most 4 ranges
        merged = []
        for r in self.ranges:
            if merged and merged[-1][2] == r[2] and merged[-1][3] == r[3] and merged[-1][1] == r[0] and r[2] != "softmax":
                merged[-1] = (merged[-1][0], r[1], r[2], r[3])
            elif merged and r[2] == "softmax" and merged[-1][2] == "softmax" and merged[-1][3] == r[3] and merged[-1][1] == r[0]:
                merged[-1] = (merged[-1][0], r[1], r[2], r[3])
            else:
                merged.append(r)
        self.acts = merged

    def tensor_groups(self):
        g = [[l.weight for l in self.linears]]
        if self.linears[0].bias is not None:
            g.append([l.bias for l in self.linears])
        if self.bns is not None:
            g += [[b.weight for b in self.bns], [b.bias for b in self.bns], [b.running_mean for b in self.bns],
                  [b.running_var for b in self.bns], [b.num_batches_tracked for b in self.bns]]
        return g

    def __call__(self, x, training, mix=None):
        """`mix` = (n_expert, H, n_out): the members are n_expert ReLU experts of width H followed by n_out softmax
        gates over them, and the caller wants the gate-mixed outputs [M, n_out * H] (training mode only)."""
        biases = [l.bias for l in self.linears] if self.linears[0].bias is not None else None
        return ops.linear_bn_act(x, [l.weight for l in self.linears], biases,
                                 bn=_bn_dict(self.bns) if self.bns is not None else None, acts=self.acts,
                                 groups=len(self.linears) if self.grouped else 1, training=training, mix=mix)


class MLP(SwrModule):
    """[Linear -> BatchNorm1d -> activation -> Dropout] per entry of `dims`, plus a final Linear(., 1)
    when `output_layer` (`basic/layers.py:231-264`).  NB the reference's positional order
    `(input_dim, output_layer, dims, ...)`: the 2nd positional argument is `output_layer`."""

    def __init__(self, input_dim, output_layer=True, dims=None, dropout=0, activation="relu"):
        super().__init__()
        if dims is None:
            dims = []
        layers = []
        for i_dim in dims:
            layers.append(nn.Linear(input_dim, i_dim))
            layers.append(nn.BatchNorm1d(i_dim))
            layers.append(activation_layer(activation))
            layers.append(nn.Dropout(p=dropout))
            input_dim = i_dim
        if output_layer:
            layers.append(nn.Linear(input_dim, 1))
        self.mlp = nn.Sequential(*layers)
        self.n_blocks = len(dims)
        self.has_output_layer = bool(output_layer)
        self.dropout_p = dropout
        self.act = activation_name(activation)          # None: not a fused activation
        self.out_dim = 1 if output_layer else input_dim

    def block(self, i):
        """(Linear, BatchNorm1d, activation module) of block i."""
        return self.mlp[4 * i], self.mlp[4 * i + 1], self.mlp[4 * i + 2]

    def output_linear(self):
        return self.mlp[4 * self.n_blocks] if self.has_output_layer else None

    def _post(self, y, i):
        if self.act is None:
            y = self.mlp[4 * i + 2](y)                   # non-fused activation (dice / prelu / leakyrelu)
        if self.dropout_p > 0 and self.training:
            y = F.dropout(y, self.dropout_p, True)
        return y

    def forward(self, x):
        for i in range(self.n_blocks):
            lin, bn, _ = self.block(i)
            act = (self.act, lin.out_features) if self.act == "softmax" else self.act
            x = self._post(LayerBank([lin], [bn], [act])(x, self.training), i)
        if self.has_output_layer:
            x = LayerBank([self.output_linear()])(x, self.training)
        return x


def mlp_bank_groups(mlps):
    """Arena adjacency for `mlp_bank_forward(mlps, ...)`."""
    g = []
    for i in range(mlps[0].n_blocks):
        g += LayerBank([m.block(i)[0] for m in mlps], [m.block(i)[1] for m in mlps]).tensor_groups()
    if mlps[0].has_output_layer:
        g += LayerBank([m.output_linear() for m in mlps]).tensor_groups()
    return g


def _tower_head_ok(mlps, x, shared_input, first_block):
    m0 = mlps[0]
    if (shared_input or not m0.training or m0.n_blocks - first_block != 1 or not m0.has_output_layer or m0.act != "relu"
            or m0.dropout_p > 0 or not x.is_cuda or os.environ.get("SWR_TOWER_HEAD", "1") == "0"):
        return False
    lin, out = m0.block(first_block)[0], m0.output_linear()
    if lin.bias is None or out.bias is None or out.out_features != 1:
        return False
    if x.dim() != 2 or x.shape[1] != len(mlps) * lin.in_features or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        return False
    return ops.tower_head_supported(lin.in_features, lin.out_features)


def mlp_bank_select(mlps, x, domain_id, first_block=0):
    """`domain_select(sigmoid(towers(x)))` for per-domain towers on their own column blocks of x (mmoe.py:50-55): with the
    fused tower kernels and the trainer's fused loss active, the output layer, the select and the BCE are one launch."""
    if _tower_head_ok(mlps, x, False, first_block):
        blocks = [m.block(first_block) for m in mlps]
        outs = [m.output_linear() for m in mlps]
        return ops.tower_head_select(x, [b[0].weight for b in blocks], [b[0].bias for b in blocks],
                                     _bn_dict([b[1] for b in blocks]), [o.weight for o in outs], [o.bias for o in outs], domain_id)
    return ops.domain_select(mlp_bank_forward(mlps, x, shared_input=False, first_block=first_block), domain_id, apply_sigmoid=True)


def mlp_bank_forward(mlps, x, shared_input, first_block=0):
    """Evaluate structurally identical MLPs together: block `first_block` reads a shared x (stacked
    outputs) or per-member column slices of x; later blocks and the output layer are grouped launches.
    Returns [M, n_mlps * out_dim]."""
    m0 = mlps[0]
    training = m0.training
    if _tower_head_ok(mlps, x, shared_input, first_block):
        # per-domain tower heads [Linear -> BN -> ReLU -> Linear(., 1)] in training mode: the fused kernels of
        # csrc/tower.hip (three launches forward instead of five, half the HBM passes of the backward)
        blocks = [m.block(first_block) for m in mlps]
        outs = [m.output_linear() for m in mlps]
        return ops.tower_head(x, [b[0].weight for b in blocks], [b[0].bias for b in blocks], _bn_dict([b[1] for b in blocks]),
                              [o.weight for o in outs], [o.bias for o in outs])
    for i in range(first_block, m0.n_blocks):
        act = (m0.act, m0.block(i)[0].out_features) if m0.act == "softmax" else m0.act
        bank = LayerBank([m.block(i)[0] for m in mlps], [m.block(i)[1] for m in mlps], [act] * len(mlps),
                         grouped=not (shared_input and i == first_block))
        x = bank(x, training)
        if m0.act is None or (m0.dropout_p > 0 and training):
            n = x.shape[1] // len(mlps)
            x = torch.cat([m._post(x[:, j * n:(j + 1) * n], i) for j, m in enumerate(mlps)], dim=1)
    if m0.has_output_layer:
        grouped = not (shared_input and m0.n_blocks == first_block)
        x = LayerBank([m.output_linear() for m in mlps], grouped=grouped)(x, training)
    return x


class GateNU(SwrModule):
    """gamma * sigmoid(W2 relu(W1 x + b1) + b2) (`basic/layers.py:307-320`), gamma = 2."""

    def __init__(self, input_dim, output_dim, hidden_dim=None, gemma=2.0):
        super().__init__()
        if hidden_dim is None:
            hidden_dim = output_dim
        self.gemma = gemma
        self.network = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, output_dim),
                                     nn.Sigmoid())

    def logits(self, inputs):
        """The output layer before its sigmoid: `ops.mul_sigmoid(x, gate.logits(g), gate.gemma)` is `x * gate(g)` in one
        pass each way (no gate tensor, no scaling pass)."""
        h = LayerBank([self.network[0]], None, ["relu"])(inputs, self.training)
        return LayerBank([self.network[2]], None, [None])(h, self.training)

    def forward(self, inputs):
        h = LayerBank([self.network[0]], None, ["relu"])(inputs, self.training)
        return LayerBank([self.network[2]], None, ["sigmoid"])(h, self.training) * self.gemma
