"""Layer restatements: EmbeddingLayer, MLP, GateNU (oracle only).

All layers read/write a flat `state` dict whose keys and shapes are the
reference's `state_dict()` keys, so golden fixtures
dumped from the reference feed the oracle directly and the state after a step
compares key by key.  Parameters live on a tape (`oracle/tape.py`); buffers
(`running_mean`, `running_var`, `num_batches_tracked`) are updated in place in
training mode exactly like torch's BatchNorm1d (momentum 0.1, unbiased
variance for the running estimate).
"""
from collections import namedtuple

import numpy as np

from . import tape as T

Sparse = namedtuple("Sparse", "name vocab_size embed_dim shared_with", defaults=(None,))
Dense = namedtuple("Dense", "name")
Seq = namedtuple("Seq", "name vocab_size embed_dim pooling shared_with padding_idx", defaults=("mean", None, None))

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def feature_dim(f):
    return f.embed_dim if isinstance(f, (Sparse, Seq)) else 1


class Ctx:
    """One forward/backward pass over a state dict."""

    def __init__(self, state, training, dtype=np.float32):
        self.state = state
        self.training = training
        self.dtype = dtype
        self.params = {}

    def p(self, key):
        """Parameter `key` as a tape leaf (one leaf per key: shared tables and
        the HAMUR hyper-net accumulate the gradients of all their uses)."""
        if key not in self.params:
            self.params[key] = T.param(np.asarray(self.state[key], dtype=self.dtype))
        return self.params[key]

    def grads(self):
        """name -> gradient; parameters that took no gradient are absent
        (torch leaves `.grad = None`, Adam then skips them: PPNet's agn tables)."""
        return {k: v.g for k, v in self.params.items() if v.g is not None}


def embedding_layer(ctx, prefix, x, features):
    """`EmbeddingLayer.forward(x, features, squeeze_dim=True)`
    (`basic/layers.py:64-105`): per sparse feature `embed(x[name].long())`,
    per dense feature `x[name].float()`; result = cat(sparse block, dense
    block) -- sparse first, dense last, whatever the feature-list order."""
    sparse, dense = [], []
    for f in features:
        if isinstance(f, Sparse):
            owner = f.shared_with if f.shared_with is not None else f.name
            w = ctx.p(f"{prefix}.embed_dict.{owner}.weight")
            sparse.append(T.embedding(w, np.asarray(x[f.name]).astype(np.int64)))
        elif isinstance(f, Seq):
            sparse.append(sequence_pooling(ctx, prefix, x, f))
        else:
            dense.append(T.const(np.asarray(x[f.name]).astype(np.float32).astype(ctx.dtype)[:, None]))
    if not sparse and not dense:
        raise ValueError("The input features can note be empty")
    return T.cat1(sparse + dense) if len(sparse) + len(dense) > 1 else (sparse + dense)[0]


def sequence_pooling(ctx, prefix, x, f):
    """SequenceFeature branch of `EmbeddingLayer.forward` (`basic/layers.py:73-87`): embed the padded id matrix
    [B, L] -> [B, L, E]; mask = ids != padding_idx (ids != -1 without one; `InputMask`, `layers.py:137-140`);
    sum = bmm(mask, emb) (`SumPooling`, `layers.py:225-228`); mean = sum / (mask.sum + 1e-16) (`AveragePooling`,
    `layers.py:202-206`); concat keeps [B, L, E], flattened by squeeze_dim (`ConcatPooling`, `layers.py:186-187`)."""
    ids = np.asarray(x[f.name]).astype(np.int64)
    if ids.ndim != 2:
        raise ValueError("sequence feature ids must be (batch_size, seq_len)")
    owner = f.shared_with if f.shared_with is not None else f.name
    emb = T.embedding(ctx.p(f"{prefix}.embed_dict.{owner}.weight"), ids)                  # [B, L, E]
    B, L, E = emb.v.shape
    if f.pooling == "concat":
        return T.reshape(emb, (B, L * E))
    if f.pooling not in ("sum", "mean"):
        raise ValueError("Sequence pooling method supports only pooling in ['sum', 'mean'], got %s." % f.pooling)
    mask = (ids != (f.padding_idx if f.padding_idx is not None else -1)).astype(ctx.dtype)
    pooled = T.sum_axis(emb * T.const(mask[:, :, None]), 1)                               # [B, E]
    if f.pooling == "mean":
        pooled = pooled / T.const((mask.sum(axis=1, keepdims=True) + 1e-16).astype(ctx.dtype))
    return pooled


def batchnorm(ctx, prefix, x):
    """nn.BatchNorm1d(eps=1e-5, momentum=0.1, affine, track_running_stats)."""
    g, b = ctx.p(prefix + ".weight"), ctx.p(prefix + ".bias")
    st = ctx.state
    if ctx.training:
        y, mu, var = T.batchnorm_train(x, g, b, BN_EPS)
        n = x.v.shape[0]
        unbiased = var * (n / (n - 1)) if n > 1 else var
        rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
        st[prefix + ".running_mean"] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mu).astype(rm.dtype)
        st[prefix + ".running_var"] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased).astype(rv.dtype)
        st[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
        return y
    return T.batchnorm_eval(x, g, b, st[prefix + ".running_mean"].astype(ctx.dtype),
                            st[prefix + ".running_var"].astype(ctx.dtype), BN_EPS)


def act(name, x):
    name = name.lower()
    if name == "relu":
        return T.relu(x)
    if name == "softmax":
        return T.softmax_rows(x)
    if name == "sigmoid":
        return T.sigmoid(x)
    raise NotImplementedError(name)


def mlp(ctx, prefix, x, dims, output_layer=True, activation="relu"):
    """`MLP` (`basic/layers.py:231-264`): [Linear, BatchNorm1d, act,
    Dropout(0)] per dim at indices 4i..4i+3 of `.mlp`, optional Linear(.,1)."""
    i = 0
    for _ in dims:
        x = T.linear(x, ctx.p(f"{prefix}.mlp.{i}.weight"), ctx.p(f"{prefix}.mlp.{i}.bias"))
        x = batchnorm(ctx, f"{prefix}.mlp.{i + 1}", x)
        x = act(activation, x)
        i += 4
    if output_layer:
        x = T.linear(x, ctx.p(f"{prefix}.mlp.{i}.weight"), ctx.p(f"{prefix}.mlp.{i}.bias"))
    return x


def gate_nu(ctx, prefix, x, gamma=2.0):
    """`GateNU` (`basic/layers.py:307-320`): gamma * sigmoid(W2 relu(W1 x + b1) + b2)."""
    h = T.relu(T.linear(x, ctx.p(prefix + ".network.0.weight"), ctx.p(prefix + ".network.0.bias")))
    o = T.sigmoid(T.linear(h, ctx.p(prefix + ".network.2.weight"), ctx.p(prefix + ".network.2.bias")))
    return o * gamma
