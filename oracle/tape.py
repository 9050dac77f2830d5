"""TEST INFRASTRUCTURE ONLY -- a minimal reverse-mode tape over numpy float32, used by the BERT4Rec-ADT and STOSA-ADT
oracles (oracle/bert_oracle.py, oracle/stosa_oracle.py) to restate the reference's forward line by line and obtain
the gradients PyTorch's autograd would produce, without importing PyTorch.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

Every primitive below is the closed form of the ATen op the reference calls (cited per function); the golden
vectors recorded from the imported reference (tools/gen_golden_wide.py -> tests/golden/) pin forward values, loss
and every parameter gradient, which pins these vector-Jacobian products as well.
"""
import math

import numpy as np
from scipy.special import erf as _erf

from . import rng

F32 = np.float32


class Var:
    """A node: value `v` (float32 ndarray), gradient `g` (accumulated), parents and a vjp closure."""
    __slots__ = ("v", "g", "parents", "vjp", "name")

    def __init__(self, v, parents=(), vjp=None, name=None):
        self.v = np.asarray(v, dtype=F32)
        self.g = None
        self.parents = parents
        self.vjp = vjp
        self.name = name

    @property
    def shape(self):
        return self.v.shape

    def acc(self, g):
        g = np.asarray(g, dtype=F32)
        if g.shape != self.v.shape:   # un-broadcast
            while g.ndim > self.v.ndim:
                g = g.sum(axis=0)
            for ax, (a, b) in enumerate(zip(self.v.shape, g.shape)):
                if a == 1 and b != 1:
                    g = g.sum(axis=ax, keepdims=True)
        self.g = g.copy() if self.g is None else self.g + g


def leaf(v, name=None):
    return Var(v, name=name)


def const(v):
    return Var(v)


def backward(root, seed=None):
    """Reverse sweep from `root` (a scalar unless seed is given)."""
    order, seen = [], set()
    stack = [(root, False)]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if id(n) in seen:
            continue
        seen.add(id(n))
        stack.append((n, True))
        for p in n.parents:
            if id(p) not in seen:
                stack.append((p, False))
    root.g = np.ones_like(root.v) if seed is None else np.asarray(seed, dtype=F32)
    for n in reversed(order):
        if n.vjp is not None and n.g is not None:
            n.vjp(n.g)


# ---- elementwise -------------------------------------------------------------------------------------------------
def add(a, b):
    out = Var(a.v + b.v, (a, b))
    out.vjp = lambda g: (a.acc(g), b.acc(g))
    return out


def sub(a, b):
    out = Var(a.v - b.v, (a, b))
    out.vjp = lambda g: (a.acc(g), b.acc(-g))
    return out


def mul(a, b):
    out = Var(a.v * b.v, (a, b))
    out.vjp = lambda g: (a.acc(g * b.v), b.acc(g * a.v))
    return out


def scale(a, c):
    c = F32(c)
    out = Var(a.v * c, (a,))
    out.vjp = lambda g: a.acc(g * c)
    return out


def div_const(a, c):
    c = F32(c)
    out = Var(a.v / c, (a,))
    out.vjp = lambda g: a.acc(g / c)
    return out


def add_const(a, c):
    out = Var(a.v + np.asarray(c, dtype=F32), (a,))
    out.vjp = lambda g: a.acc(g)
    return out


def neg(a):
    return scale(a, -1.0)


def square(a):
    out = Var(a.v * a.v, (a,))
    out.vjp = lambda g: a.acc(g * 2.0 * a.v)
    return out


def sqrt(a):
    r = np.sqrt(a.v)
    out = Var(r, (a,))
    out.vjp = lambda g: a.acc(g * (F32(0.5) / r))
    return out


def clamp_min(a, lo):
    """torch.clamp(x, min=lo): gradient passes where x >= lo."""
    lo = F32(lo)
    out = Var(np.maximum(a.v, lo), (a,))
    out.vjp = lambda g: a.acc(g * (a.v >= lo))
    return out


def relu(a):
    out = Var(np.maximum(a.v, F32(0)), (a,))
    out.vjp = lambda g: a.acc(g * (a.v > 0))
    return out


def gelu(a):
    """nn.GELU() (erf form): bert4rec/model/modules.py:125, bert.py:52."""
    x = a.v
    cdf = (F32(0.5) * (F32(1) + _erf(x * F32(0.7071067811865476)))).astype(F32)
    out = Var(x * cdf, (a,))
    out.vjp = lambda g: a.acc(g * (cdf + x * F32(0.3989422804014327) * np.exp(F32(-0.5) * x * x)))
    return out


def elu(a, plus_one=False):
    """nn.ELU() (alpha 1): stosa/modules.py:212,477; `+ 1` for covariances (:236-238)."""
    x = a.v
    e = np.where(x > 0, x, np.exp(np.minimum(x, F32(0))) - F32(1)).astype(F32)
    out = Var(e + F32(1) if plus_one else e, (a,))
    out.vjp = lambda g: a.acc(g * np.where(x > 0, F32(1), np.exp(np.minimum(x, F32(0)))).astype(F32))
    return out


def sigmoid(a):
    s = (F32(1) / (F32(1) + np.exp(-a.v))).astype(F32)
    out = Var(s, (a,))
    out.vjp = lambda g: a.acc(g * s * (F32(1) - s))
    return out


def log(a):
    out = Var(np.log(a.v), (a,))
    out.vjp = lambda g: a.acc(g / a.v)
    return out


def mul_mask(a, m):
    """x * constant array (dropout keep * 1/(1-p), padding masks, istarget)."""
    m = np.asarray(m, dtype=F32)
    out = Var(a.v * m, (a,))
    out.vjp = lambda g: a.acc(g * m)
    return out


def masked_fill(a, mask, value):
    """x.masked_fill(mask, value): bert4rec/model/modules.py:90-92."""
    out = Var(np.where(mask, F32(value), a.v), (a,))
    out.vjp = lambda g: a.acc(np.where(mask, F32(0), g))
    return out


# ---- shape ---------------------------------------------------------------------------------------------------------
def reshape(a, shape):
    out = Var(a.v.reshape(shape), (a,))
    out.vjp = lambda g: a.acc(g.reshape(a.v.shape))
    return out


def transpose(a, axes):
    inv = np.argsort(axes)
    out = Var(np.transpose(a.v, axes), (a,))
    out.vjp = lambda g: a.acc(np.transpose(g, inv))
    return out


def index(a, idx):
    """a[idx] with basic/advanced indexing; scatter-add in the reverse."""
    out = Var(a.v[idx], (a,))

    def vjp(g):
        z = np.zeros_like(a.v)
        np.add.at(z, idx, g)
        a.acc(z)
    out.vjp = vjp
    return out


def embedding(table, ids, padding_idx=None):
    """nn.Embedding: rows of `table`; with padding_idx the gradient of that row is dropped (the forward still reads the
    stored row, which the reference's init overwrites with non-zero values: bert4rec/trainer.py:29-33)."""
    ids = np.asarray(ids)
    out = Var(table.v[ids], (table,))

    def vjp(g):
        z = np.zeros_like(table.v)
        gi = g.reshape(-1, g.shape[-1])
        fl = ids.reshape(-1)
        if padding_idx is not None:
            keep = fl != padding_idx
            np.add.at(z, fl[keep], gi[keep])
        else:
            np.add.at(z, fl, gi)
        table.acc(z)
    out.vjp = vjp
    return out


# ---- reductions ------------------------------------------------------------------------------------------------------
def sum_(a, axis=None, keepdims=False):
    out = Var(a.v.sum(axis=axis, keepdims=keepdims, dtype=F32), (a,))

    def vjp(g):
        gg = g if (keepdims or axis is None) else np.expand_dims(g, axis)
        a.acc(np.broadcast_to(gg, a.v.shape))
    out.vjp = vjp
    return out


def mean(a):
    n = a.v.size
    out = Var(a.v.mean(dtype=F32), (a,))
    out.vjp = lambda g: a.acc(np.broadcast_to(g / F32(n), a.v.shape))
    return out


# ---- linear algebra -----------------------------------------------------------------------------------------------------
def matmul(a, b):
    """torch.matmul on (..., m, k) x (..., k, n) (batch dims broadcast)."""
    out = Var(np.matmul(a.v, b.v), (a, b))
    out.vjp = lambda g: (a.acc(np.matmul(g, np.swapaxes(b.v, -1, -2))), b.acc(np.matmul(np.swapaxes(a.v, -1, -2), g)))
    return out


def linear(x, W, b=None):
    """nn.Linear: x W^T + b with W (out, in)."""
    y = np.matmul(x.v, W.v.T)
    if b is not None:
        y = y + b.v
    out = Var(y, (x, W) if b is None else (x, W, b))

    def vjp(g):
        x.acc(np.matmul(g, W.v))
        g2 = g.reshape(-1, g.shape[-1])
        W.acc(np.matmul(g2.T, x.v.reshape(-1, x.v.shape[-1])))
        if b is not None:
            b.acc(g2.sum(axis=0))
    out.vjp = vjp
    return out


def layernorm(x, w, b, eps):
    """LayerNorm over the last axis, eps inside the square root (torch.nn.LayerNorm and stosa/modules.py:86-99)."""
    mu = x.v.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x.v - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    rstd = (F32(1) / np.sqrt(var + F32(eps))).astype(F32)
    xh = xc * rstd
    out = Var(xh * w.v + b.v, (x, w, b))

    def vjp(g):
        g2 = g.reshape(-1, g.shape[-1])
        w.acc((g2 * xh.reshape(g2.shape)).sum(axis=0))
        b.acc(g2.sum(axis=0))
        dxh = g * w.v
        m1 = dxh.mean(axis=-1, keepdims=True, dtype=F32)
        m2 = (dxh * xh).mean(axis=-1, keepdims=True, dtype=F32)
        x.acc(rstd * (dxh - m1 - xh * m2))
    out.vjp = vjp
    return out


def softmax(a):
    m = a.v.max(axis=-1, keepdims=True)
    e = np.exp(a.v - m)
    p = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    out = Var(p, (a,))
    out.vjp = lambda g: a.acc(p * (g - (g * p).sum(axis=-1, keepdims=True, dtype=F32)))
    return out


def log_softmax(a):
    m = a.v.max(axis=-1, keepdims=True)
    z = a.v - m
    lse = np.log(np.exp(z).sum(axis=-1, keepdims=True, dtype=F32))
    out = Var(z - lse, (a,))
    out.vjp = lambda g: a.acc(g - np.exp(out.v) * g.sum(axis=-1, keepdims=True, dtype=F32))
    return out


def cross_entropy(logits, labels, ignore_index=0):
    """nn.CrossEntropyLoss(ignore_index=0), mean over the non-ignored rows: bert4rec/trainer.py:45,113-115."""
    labels = np.asarray(labels).reshape(-1)
    z = logits.v.reshape(labels.shape[0], -1)
    m = z.max(axis=-1, keepdims=True)
    lse = (m + np.log(np.exp(z - m).sum(axis=-1, keepdims=True, dtype=F32))).reshape(-1)
    keep = labels != ignore_index
    n = max(int(keep.sum()), 1)
    picked = z[np.arange(z.shape[0]), labels]
    out = Var(F32(((lse - picked) * keep).sum(dtype=F32) / F32(n)), (logits,))

    def vjp(g):
        p = np.exp(z - lse[:, None])
        p[np.arange(z.shape[0]), labels] -= 1
        p *= (keep[:, None] * (g / F32(n))).astype(F32)
        logits.acc(p.reshape(logits.v.shape))
    out.vjp = vjp
    return out


# ---- dropout with the shared hash RNG (oracle/rng.py; adt_amd/csrc/adt_common.cuh) ---------------------------------------------
def dropout(a, p, seed, site, idx, training=True):
    """Inverted dropout; `idx` gives every element's global index (same convention as the kernel that applies it)."""
    if not training or p <= 0.0:
        return a
    keep = rng.keep_mask(seed, site, idx, p)
    return mul_mask(a, keep.astype(F32) * F32(1.0 / (1.0 - rng.drop_prob(p))))


def idx_rows(T, N, row_offset=0):
    """Element indices of a (T, N) dense output: (row + row_offset) * N + col."""
    return (np.arange(T, dtype=np.int64)[:, None] + row_offset) * N + np.arange(N, dtype=np.int64)[None, :]


def idx_attn(B, H, L, b_offset=0):
    """Element indices of (B, H, L, L) attention probabilities: (((b + b_offset) * H + h) * L + q) * L + key."""
    bh = (np.arange(B, dtype=np.int64)[:, None] + b_offset) * H + np.arange(H, dtype=np.int64)[None, :]
    return ((bh[:, :, None] * L + np.arange(L, dtype=np.int64)[None, None, :])[:, :, :, None] * L + np.arange(L, dtype=np.int64)[None, None, None, :])


# ---- optimiser -------------------------------------------------------------------------------------------------------------------------
def clip_adam(P, G, state, lr, b1, b2, eps=1e-8, clip=None, l2=0.0):
    """torch.nn.utils.clip_grad_norm_(params, clip) then torch.optim.Adam(lr, betas, weight_decay=l2).step() over dicts
    of float32 arrays (in place).  Entries of G that are None are skipped exactly as torch skips grad=None."""
    names = [k for k in P if G.get(k) is not None]
    tn = math.sqrt(sum(float((G[k].astype(np.float64) ** 2).sum()) for k in names))
    coef = 1.0
    if clip is not None:
        coef = min(1.0, float(clip) / (tn + 1e-6))
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
    for k in names:
        g = (G[k] * F32(coef)).astype(F32)
        if l2:
            g = g + F32(l2) * P[k]
        m = state.setdefault("m", {}).get(k)
        v = state.setdefault("v", {}).get(k)
        m = np.zeros_like(P[k]) if m is None else m
        v = np.zeros_like(P[k]) if v is None else v
        m = (F32(b1) * m + F32(1 - b1) * g).astype(F32)
        v = (F32(b2) * v + F32(1 - b2) * g * g).astype(F32)
        state["m"][k], state["v"][k] = m, v
        P[k] = (P[k] - F32(lr / bc1) * m / (np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps))).astype(F32)
    return tn
