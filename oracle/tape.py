"""Minimal reverse-mode differentiation tape over numpy arrays (oracle only).

The reference obtains its backward pass from torch.autograd
(`trainers/ctr_trainer.py:72` `loss.backward()`); the oracle restates the
derivative rules of exactly the primitives the in-scope models use, so the
backward pass of every model is derived from its forward restatement the same
way the reference derives it.  Gradients accumulate (sum) over multiple uses of
a value, like autograd.

Every op runs in the dtype of its inputs (float32 by default, float64 when the
caller builds the model in float64 for a tighter truth).
"""
import numpy as np


class Var:
    """A value on the tape. `g` is filled by `backward`."""
    __slots__ = ("v", "g", "_bw", "_prev", "req")

    def __init__(self, v, prev=(), bw=None, req=None):
        self.v = v
        self.g = None
        self._bw = bw
        self._prev = prev
        self.req = any(p.req for p in prev) if req is None else req

    @property
    def shape(self):
        return self.v.shape

    def acc(self, g):
        if not self.req:
            return
        if self.g is None:
            self.g = np.array(g, dtype=self.v.dtype, copy=True)
        else:
            self.g = self.g + g

    # python operators -> tape ops
    def __add__(self, o): return add(self, o)
    __radd__ = __add__
    def __sub__(self, o): return sub(self, o)
    def __rsub__(self, o): return sub(o, self)
    def __mul__(self, o): return mul(self, o)
    __rmul__ = __mul__
    def __truediv__(self, o): return div(self, o)
    def __matmul__(self, o): return matmul(self, o)


def const(v, dtype=None):
    return Var(np.asarray(v, dtype=dtype), req=False)


def param(v):
    return Var(v, req=True)


def _wrap(x, like):
    if isinstance(x, Var):
        return x
    return Var(np.asarray(x, dtype=like.v.dtype), req=False)


def _unbroadcast(g, shape):
    """Sum `g` down to `shape` (inverse of numpy broadcasting)."""
    if g.shape == tuple(shape):
        return g
    while g.ndim > len(shape):
        g = g.sum(axis=0)
    for i, s in enumerate(shape):
        if s == 1 and g.shape[i] != 1:
            g = g.sum(axis=i, keepdims=True)
    return g


def backward(root, seed=None):
    """Reverse sweep from `root` (scalar unless `seed` is given)."""
    topo, seen = [], set()
    stack = [(root, False)]
    while stack:
        node, done = stack.pop()
        if done:
            topo.append(node)
            continue
        if id(node) in seen:
            continue
        seen.add(id(node))
        stack.append((node, True))
        for p in node._prev:
            if id(p) not in seen and p.req:
                stack.append((p, False))
    root.g = np.ones_like(root.v) if seed is None else np.asarray(seed, dtype=root.v.dtype)
    for node in reversed(topo):
        if node._bw is not None and node.g is not None:
            node._bw(node.g)


# ---------------------------------------------------------------- elementwise
def add(a, b):
    a = _wrap(a, b if isinstance(b, Var) else a); b = _wrap(b, a)
    def bw(g):
        a.acc(_unbroadcast(g, a.v.shape)); b.acc(_unbroadcast(g, b.v.shape))
    return Var(a.v + b.v, (a, b), bw)


def sub(a, b):
    a = _wrap(a, b if isinstance(b, Var) else a); b = _wrap(b, a)
    def bw(g):
        a.acc(_unbroadcast(g, a.v.shape)); b.acc(_unbroadcast(-g, b.v.shape))
    return Var(a.v - b.v, (a, b), bw)


def mul(a, b):
    a = _wrap(a, b if isinstance(b, Var) else a); b = _wrap(b, a)
    def bw(g):
        a.acc(_unbroadcast(g * b.v, a.v.shape)); b.acc(_unbroadcast(g * a.v, b.v.shape))
    return Var(a.v * b.v, (a, b), bw)


def div(a, b):
    a = _wrap(a, b if isinstance(b, Var) else a); b = _wrap(b, a)
    def bw(g):
        a.acc(_unbroadcast(g / b.v, a.v.shape))
        b.acc(_unbroadcast(-g * a.v / (b.v * b.v), b.v.shape))
    return Var(a.v / b.v, (a, b), bw)


def sqrt(a):
    out = np.sqrt(a.v)
    return Var(out, (a,), lambda g: a.acc(g * (0.5 / out).astype(a.v.dtype)))


def relu(a):
    out = np.maximum(a.v, 0)
    return Var(out, (a,), lambda g: a.acc(g * (out > 0)))


def sigmoid(a):
    # numerically stable two-sided form; same value as torch.sigmoid to 1 ulp
    x = a.v
    e = np.exp(-np.abs(x))
    out = np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(x.dtype)
    return Var(out, (a,), lambda g: a.acc(g * out * (1 - out)))


def softmax_rows(a):
    """Softmax(dim=1) (`basic/activation.py:46-47`)."""
    z = a.v - a.v.max(axis=1, keepdims=True)
    e = np.exp(z)
    out = e / e.sum(axis=1, keepdims=True)
    def bw(g):
        a.acc(out * (g - (g * out).sum(axis=1, keepdims=True)))
    return Var(out, (a,), bw)


def detach(a):
    return Var(a.v, req=False)


# ------------------------------------------------------------ shape / reduce
def matmul(a, b):
    def bw(g):
        a.acc(g @ b.v.T); b.acc(a.v.T @ g)
    return Var(a.v @ b.v, (a, b), bw)


def sum_axis(a, axis, keepdims=False):
    def bw(g):
        if not keepdims:
            g = np.expand_dims(g, axis)
        a.acc(np.broadcast_to(g, a.v.shape))
    return Var(a.v.sum(axis=axis, keepdims=keepdims), (a,), bw)


def mean0(a):
    n = a.v.shape[0]
    return Var(a.v.mean(axis=0), (a,), lambda g: a.acc(np.broadcast_to(g / n, a.v.shape)))


def cat1(parts):
    """torch.cat(parts, dim=1)."""
    widths = [p.v.shape[1] for p in parts]
    offs = np.cumsum([0] + widths)
    def bw(g):
        for p, lo, hi in zip(parts, offs[:-1], offs[1:]):
            p.acc(g[:, lo:hi])
    return Var(np.concatenate([p.v for p in parts], axis=1), tuple(parts), bw)


def slice1(a, lo, hi):
    def bw(g):
        full = np.zeros_like(a.v); full[:, lo:hi] = g; a.acc(full)
    return Var(a.v[:, lo:hi], (a,), bw)


def reshape(a, shape):
    return Var(a.v.reshape(shape), (a,), lambda g: a.acc(g.reshape(a.v.shape)))


def embedding(weight, idx):
    """nn.Embedding forward; DENSE gradient (`basic/initializers.py:17`,
    sparse=False): grad[row] = sum of output grads of every lookup of row."""
    idx = np.asarray(idx).astype(np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= weight.v.shape[0]):
        raise IndexError("index out of range in self")
    def bw(g):
        full = np.zeros_like(weight.v)
        np.add.at(full, idx, g)
        weight.acc(full)
    return Var(weight.v[idx], (weight,), bw)


def einsum(spec, *ops):
    """np.einsum with the transposed-einsum derivative for every operand
    (used for HAMUR's `'mi,bij,jn->bmn'` / `'bf,bfj->bj'`, `hamur.py:346-358`)."""
    ins, out = spec.split("->")
    ins = ins.split(",")
    def bw(g):
        for i, op in enumerate(ops):
            if not op.req:
                continue
            others = [ops[j].v for j in range(len(ops)) if j != i]
            osubs = [ins[j] for j in range(len(ops)) if j != i]
            op.acc(np.einsum(",".join([out] + osubs) + "->" + ins[i], g, *others, optimize=True))
    # optimize=True: pairwise contractions (the same sums in another association; the naive three-operand loop is
    # O(B m k k n) and takes minutes at HAMUR's BASELINE widths)
    return Var(np.einsum(spec, *[o.v for o in ops], optimize=True), tuple(ops), bw)


# ------------------------------------------------------------------ fused nn
def linear(x, w, b=None):
    """nn.Linear: y = x @ W^T + b with W [out,in]."""
    def bw(g):
        x.acc(g @ w.v); w.acc(g.T @ x.v)
        if b is not None:
            b.acc(g.sum(axis=0))
    y = x.v @ w.v.T
    if b is not None:
        y = y + b.v
    return Var(y, (x, w) + ((b,) if b is not None else ()), bw)


def batchnorm_train(x, gamma, beta, eps):
    """BatchNorm1d in training mode: batch mean, BIASED batch variance for the
    normalisation (torch native_batch_norm).  Returns (y, mean, biased_var)."""
    n = x.v.shape[0]
    mu = x.v.mean(axis=0)
    xc = x.v - mu
    var = (xc * xc).mean(axis=0)
    rstd = (1.0 / np.sqrt(var + eps)).astype(x.v.dtype)
    xhat = xc * rstd
    def bw(g):
        gamma.acc((g * xhat).sum(axis=0)); beta.acc(g.sum(axis=0))
        gs = g.sum(axis=0); gx = (g * xhat).sum(axis=0)
        x.acc((gamma.v * rstd / n) * (n * g - gs - xhat * gx))
    return Var(xhat * gamma.v + beta.v, (x, gamma, beta), bw), mu, var


def batchnorm_eval(x, gamma, beta, rmean, rvar, eps):
    scale = gamma * const(1.0 / np.sqrt(rvar + eps), x.v.dtype)
    return (x - const(rmean, x.v.dtype)) * scale + beta


def layernorm(x, gamma, beta, eps):
    """torch.nn.LayerNorm over the last dim of x [B, N]: per-row mean, BIASED per-row variance, eps inside the sqrt,
    elementwise affine (`m3oe.py:59,124` nn.LayerNorm)."""
    n = x.v.shape[1]
    mu = x.v.mean(axis=1, keepdims=True)
    xc = x.v - mu
    var = (xc * xc).mean(axis=1, keepdims=True)
    rstd = (1.0 / np.sqrt(var + eps)).astype(x.v.dtype)
    xhat = xc * rstd
    def bw(g):
        gamma.acc((g * xhat).sum(axis=0)); beta.acc(g.sum(axis=0))
        dh = g * gamma.v
        x.acc(rstd * (dh - dh.mean(axis=1, keepdims=True) - xhat * (dh * xhat).mean(axis=1, keepdims=True)))
    return Var(xhat * gamma.v + beta.v, (x, gamma, beta), bw)


def select_domain(ys, dom):
    """final = 0; for d: final = where(dom == d, ys[d], final)
    (`mmoe.py:53-55`, `base_example.py:72-74`); ids outside [0, D) give 0.0.
    The compare is an exact integer equality on the raw id."""
    dom = np.asarray(dom)
    out = np.zeros_like(ys[0].v)
    masks = []
    for d, y in enumerate(ys):
        m = (dom == d)
        masks.append(m)
        out = np.where(m.reshape((-1,) + (1,) * (out.ndim - 1)), y.v, out)
    def bw(g):
        for m, y in zip(masks, ys):
            y.acc(g * m.reshape((-1,) + (1,) * (g.ndim - 1)))
    return Var(out, tuple(ys), bw)


def bce_mean(p, y):
    """torch.nn.BCELoss(reduction='mean') on probabilities
    (`ctr_trainer.py:56,70`): logs clamped at -100; the derivative is torch's
    (p - y) / max(p (1 - p), 1e-12) / B."""
    yv = np.asarray(y, dtype=p.v.dtype)
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(p.v), -100.0)
        l1p = np.maximum(np.log1p(-p.v), -100.0)
    loss = -(yv * lp + (1 - yv) * l1p)
    n = p.v.size
    def bw(g):
        p.acc(g * (p.v - yv) / np.maximum(p.v * (1 - p.v), 1e-12) / n)
    return Var(loss.mean(dtype=p.v.dtype), (p,), bw)
