"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the SASRec-ADT hot path (forward, loss, backward,
clip + Adam, predict, ranking metrics).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (adt_amd/) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  tools/gen_golden.py imports the reference (/root/reference/sasrec, PyTorch CPU) in
the build container and records golden vectors under tests/golden/; tests/test_oracle_golden.py checks every
function here against them (forward tensors, loss, every parameter gradient, post-Adam weights).  The
arithmetic that lives in PyTorch itself (nn.MultiheadAttention, LayerNorm, Adam, clip_grad_norm_) is pinned
the same way, with torch 2.10.0 (the reference has no lockfile; README.md:13-21 asks for 1.11).

Each function cites the reference file:line it restates (paths relative to /root/reference).
Everything is batch-first (B, L, d); "rows" are tokens in (b, l) order.
"""
import math
import numpy as np

from . import rng

# ----------------------------------------------------------------------------------------------
# dropout sites (shared with adt_amd/csrc/adt_common.cuh)
SITE_EMB_SEQ = 1
SITE_EMB_DEC = 2


def enc_sites(i):
    b = 16 + 8 * i
    return {"attn": b, "ffn1": b + 1, "ffn2": b + 2}


def dec_sites(i):
    b = 128 + 8 * i
    return {"slf": b, "enc": b + 1, "ffn1": b + 2, "ffn2": b + 3}


LN_EPS = 1e-8  # sasrec/modules.py:638,640,660 ; sasrec/model.py:28


class Cfg:
    """Shape/config record (mirrors the fields SASRecADT reads from args, sasrec/model.py:8-30)."""

    def __init__(self, item_num, maxlen, hidden_units, num_heads, num_layers, dropout=0.0):
        self.item_num = item_num
        self.maxlen = maxlen
        self.hidden_units = hidden_units
        self.num_heads = num_heads
        self.num_layers = num_layers
        self.dropout = dropout


# ----------------------------------------------------------------------------------------------
# parameter inventory: names and shapes exactly as the reference's state_dict
def param_shapes(cfg):
    d, H, L, V = cfg.hidden_units, cfg.num_heads, cfg.maxlen, cfg.item_num
    hd = d // H
    s = [("item_emb.weight", (V + 1, d)), ("pos_emb.weight", (L, d))]
    for i in range(cfg.num_layers):
        p = "encoder.encoder_layers.%d." % i
        s += [(p + "attention_layernorm.weight", (d,)), (p + "attention_layernorm.bias", (d,)),
              (p + "attention_layer.in_proj_weight", (3 * d, d)), (p + "attention_layer.in_proj_bias", (3 * d,)),
              (p + "attention_layer.out_proj.weight", (d, d)), (p + "attention_layer.out_proj.bias", (d,)),
              (p + "forward_layernorm.weight", (d,)), (p + "forward_layernorm.bias", (d,)),
              (p + "forward_layer.conv1.weight", (d, d, 1)), (p + "forward_layer.conv1.bias", (d,)),
              (p + "forward_layer.conv2.weight", (d, d, 1)), (p + "forward_layer.conv2.bias", (d,)),
              (p + "sparse.weight", (H, hd)), (p + "sparse.bias", (H,))]
    for i in range(cfg.num_layers):
        p = "decoder.decoder_layers.%d." % i
        s += [(p + "layer_norm.weight", (d,)), (p + "layer_norm.bias", (d,))]
        for a in ("slf_attn", "enc_attn"):
            s += [(p + a + ".in_proj_weight", (3 * d, d)), (p + a + ".in_proj_bias", (3 * d,)),
                  (p + a + ".out_proj.weight", (d, d)), (p + a + ".out_proj.bias", (d,))]
        s += [(p + "pos_ffn.conv1.weight", (d, d, 1)), (p + "pos_ffn.conv1.bias", (d,)),
              (p + "pos_ffn.conv2.weight", (d, d, 1)), (p + "pos_ffn.conv2.bias", (d,)),
              (p + "pos_ffn_layernorm.weight", (d,)), (p + "pos_ffn_layernorm.bias", (d,))]
    s += [("last_layernorm.weight", (d,)), ("last_layernorm.bias", (d,))]
    return s


def is_unused(name, num_heads=2):
    """DecoderLayer.pos_ffn_layernorm is constructed but never called (sasrec/modules.py:664,666-677):
    its parameters get grad=None and are skipped by clip_grad_norm_/Adam.  With num_heads == 1 the
    independence loss is skipped (sasrec/main.py:160) and the head classifier gets grad=None too."""
    return "pos_ffn_layernorm" in name or (num_heads == 1 and ".sparse." in name)


def init_params(cfg, seed=0, dtype=np.float32):
    """numpy-RandomState weights in the spirit of sasrec/main.py:95-99 (xavier_normal_ on every >=2-D
    parameter, including the padding row; 1-D parameters keep torch defaults: LN weight 1, LN/linear bias
    small).  1-D biases get small random values here so that tests exercise them."""
    r = np.random.RandomState(seed)
    out = {}
    for name, shp in param_shapes(cfg):
        if len(shp) >= 2:
            fan_out, fan_in = shp[0], shp[1]
            rf = 1
            for x in shp[2:]:
                rf *= x
            std = math.sqrt(2.0 / ((fan_in + fan_out) * rf))
            out[name] = (r.randn(*shp) * std).astype(dtype)
        elif name.endswith("norm.weight"):
            out[name] = (1.0 + 0.1 * r.randn(*shp)).astype(dtype)
        else:
            out[name] = (0.05 * r.randn(*shp)).astype(dtype)
    return out


# ----------------------------------------------------------------------------------------------
# primitives
_OPERANDS = [None]


def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32: what `(__bf16)x` does in the kernels."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


class operands:
    """`with operands("bf16"):` -- every MATRIX product below rounds both operands to bfloat16 first and accumulates in fp32,
    the arithmetic of the MFMA path's bf16 mode.  Everything else (LayerNorm, softmax statistics, dropout,
    residuals, head classifier, logits, losses) stays fp32, as in the kernels.  Default (None): plain fp32, the reference's
    arithmetic.  Used by the GPU tests to hold the bf16 kernels to a tighter bound than fp32-vs-bf16 rounding noise allows."""

    def __init__(self, mode):
        assert mode in (None, "bf16")
        self.mode = mode

    def __enter__(self):
        self.prev = _OPERANDS[0]
        _OPERANDS[0] = self.mode

    def __exit__(self, *exc):
        _OPERANDS[0] = self.prev


def _mm(a, b):
    if _OPERANDS[0] == "bf16":
        return bf16_round(a) @ bf16_round(b)
    return a @ b


def _dropout(x, p, seed, site, idx):
    """y = x * keep / (1 - drop_prob(p)); keep from the shared hash RNG (oracle/rng.py: 8-bit thresholds)."""
    if p <= 0.0:
        return x, None
    keep = rng.keep_mask(seed, site, idx, p)
    scale = x.dtype.type(1.0 / (1.0 - rng.drop_prob(p)))
    return x * keep * scale, keep


def _row_idx(B, L, n, b_offset):
    """element index (b_global*L + l)*n + c for a (B, L, n) tensor."""
    b = (np.arange(B, dtype=np.int64) + b_offset)[:, None, None]
    l = np.arange(L, dtype=np.int64)[None, :, None]
    c = np.arange(n, dtype=np.int64)[None, None, :]
    return (b * L + l) * n + c


def layer_norm(x, w, b, eps=LN_EPS):
    """torch.nn.LayerNorm over the last dim (biased variance, eps inside the sqrt)."""
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(eps))
    xh = xc * rstd
    return xh * w + b, (xh, rstd)


def layer_norm_bwd(dy, cache, w):
    xh, rstd = cache
    dxh = dy * w
    dw = (dy * xh).reshape(-1, xh.shape[-1]).sum(0)
    db = dy.reshape(-1, xh.shape[-1]).sum(0)
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xh * (dxh * xh).mean(-1, keepdims=True))
    return dx, dw, db


def linear(x, w, b):
    return _mm(x, w.T) + b


def linear_bwd(dy, x, w):
    n = w.shape[0]
    dy2 = dy.reshape(-1, n)
    x2 = x.reshape(-1, w.shape[1])
    return _mm(dy, w), _mm(dy2.T, x2), dy2.sum(0)


def embed(ids, E, P, p, seed, site, b_offset=0):
    """sasrec/model.py:34-41 (and the identical decode():53-59):
    x = dropout(E[ids]*sqrt(d) + P[0..L-1]) * (ids != 0)."""
    B, L = ids.shape
    d = E.shape[1]
    x = E[ids] * E.dtype.type(d ** 0.5) + P[None, :L, :]
    x, keep = _dropout(x, p, seed, site, _row_idx(B, L, d, b_offset))
    m = (ids != 0)[..., None].astype(E.dtype)
    return x * m, (keep, m)


def attention(q, k, v, H, causal, p, seed, site, b_offset=0, key_keep=None):
    """sasrec/modules.py:21-64 (_scaled_dot_product_attention) with the head split/merge of
    multi_head_attention_forward (:457-468,:517): softmax(q/sqrt(hd) k^T + mask) -> dropout -> @ v.
    q, k, v, result: (B, L, d) with head h in columns [h*hd, (h+1)*hd)."""
    B, L, d = q.shape
    hd = d // H
    qh = q.reshape(B, L, H, hd).transpose(0, 2, 1, 3) / q.dtype.type(math.sqrt(hd))
    kh = k.reshape(B, L, H, hd).transpose(0, 2, 1, 3)
    vh = v.reshape(B, L, H, hd).transpose(0, 2, 1, 3)
    s = _mm(qh, kh.transpose(0, 1, 3, 2))  # (B,H,L,L)
    if causal:
        s = np.where(np.tril(np.ones((L, L), dtype=bool))[None, None], s, -np.inf).astype(q.dtype)
    if key_keep is not None:  # key-padding variant (bert4rec); not used by sasrec
        s = np.where(key_keep[:, None, None, :], s, q.dtype.type(-1e9))
    mx = s.max(-1, keepdims=True)
    e = np.exp(s - mx)
    den = e.sum(-1, keepdims=True)
    pr = e / den
    lse = (mx + np.log(den))[..., 0]
    if p > 0.0:
        bh = ((np.arange(B, dtype=np.int64) + b_offset)[:, None] * H + np.arange(H)[None, :])[:, :, None, None]
        idx = (bh * L + np.arange(L, dtype=np.int64)[None, None, :, None]) * L + np.arange(L, dtype=np.int64)[None, None, None, :]
        pd, keep = _dropout(pr, p, seed, site, idx)
    else:
        pd, keep = pr, None
    o = _mm(pd, vh).transpose(0, 2, 1, 3).reshape(B, L, d)
    return o, (qh, kh, vh, pr, pd, keep, p, lse)


def attention_bwd(do, cache, H):
    qh, kh, vh, pr, pd, keep, p, _ = cache
    B, _, L, hd = qh.shape
    d = H * hd
    doh = do.reshape(B, L, H, hd).transpose(0, 2, 1, 3)
    dvh = _mm(pd.transpose(0, 1, 3, 2), doh)
    dpd = _mm(doh, vh.transpose(0, 1, 3, 2))
    if keep is not None:
        dpr = dpd * keep * pr.dtype.type(1.0 / (1.0 - rng.drop_prob(p)))
    else:
        dpr = dpd
    ds = pr * (dpr - (dpr * pr).sum(-1, keepdims=True))
    dqh = _mm(ds, kh) / qh.dtype.type(math.sqrt(hd))
    dkh = _mm(ds.transpose(0, 1, 3, 2), qh)  # qh already carries the 1/sqrt(hd)
    unh = lambda t: t.transpose(0, 2, 1, 3).reshape(B, L, d)
    return unh(dqh), unh(dkh), unh(dvh)


def ffn(x, w1, b1, w2, b2, p, seed, s1, s2, b_offset=0):
    """PointWiseFeedForward without the residual (sasrec/modules.py:629):
    dropout2(conv2(relu(dropout1(conv1(x))))); Conv1d(k=1) == Linear d->d."""
    B, L, d = x.shape
    idx = _row_idx(B, L, d, b_offset)
    t = linear(x, w1, b1)
    t, k1 = _dropout(t, p, seed, s1, idx)
    u = np.maximum(t, 0)
    f = linear(u, w2, b2)
    f, k2 = _dropout(f, p, seed, s2, idx)
    return f, (x, u, k1, k2, p)


def ffn_bwd(df, cache, w1, w2):
    x, u, k1, k2, p = cache
    sc = x.dtype.type(1.0 / (1.0 - rng.drop_prob(p))) if p > 0 else x.dtype.type(1.0)
    if k2 is not None:
        df = df * k2 * sc
    du, dw2, db2 = linear_bwd(df, u, w2)
    dt = du * (u > 0)
    if k1 is not None:
        dt = dt * k1 * sc
    dx, dw1, db1 = linear_bwd(dt, x, w1)
    return dx, dw1, db1, dw2, db2


# ----------------------------------------------------------------------------------------------
# layers
def encoder_layer(x, m, P, pre, H, p, seed, sites, b_offset):
    """EncoderLayer.forward, sasrec/modules.py:644-655 (q from LN(x), k/v from raw x; residual adds LN(x))."""
    d = x.shape[-1]
    Win, bin_ = P[pre + "attention_layer.in_proj_weight"], P[pre + "attention_layer.in_proj_bias"]
    Q, lnc1 = layer_norm(x, P[pre + "attention_layernorm.weight"], P[pre + "attention_layernorm.bias"])
    q = linear(Q, Win[:d], bin_[:d])
    k = linear(x, Win[d:2 * d], bin_[d:2 * d])
    v = linear(x, Win[2 * d:], bin_[2 * d:])
    o, ac = attention(q, k, v, H, True, p, seed, sites["attn"], b_offset)
    a = linear(o, P[pre + "attention_layer.out_proj.weight"], P[pre + "attention_layer.out_proj.bias"])
    h = Q + a
    h2, lnc2 = layer_norm(h, P[pre + "forward_layernorm.weight"], P[pre + "forward_layernorm.bias"])
    f, fc = ffn(h2, P[pre + "forward_layer.conv1.weight"][..., 0], P[pre + "forward_layer.conv1.bias"],
                P[pre + "forward_layer.conv2.weight"][..., 0], P[pre + "forward_layer.conv2.bias"],
                p, seed, sites["ffn1"], sites["ffn2"], b_offset)
    y = (h2 + f) * m
    # independence head classifier: SparseInputLinear + log_softmax (sasrec/modules.py:648-649,679-703)
    B, L, _ = x.shape
    hd = d // H
    oh = o.reshape(B, L, H, hd)
    z = oh @ P[pre + "sparse.weight"].T + P[pre + "sparse.bias"]  # (B,L,H,H)
    zm = z.max(-1, keepdims=True)
    rec = z - zm - np.log(np.exp(z - zm).sum(-1, keepdims=True))
    return y, rec, dict(x=x, Q=Q, lnc1=lnc1, o=o, ac=ac, lnc2=lnc2, fc=fc, m=m, oh=oh, rec=rec)


def encoder_layer_bwd(dy, drec, c, P, pre, H, G):
    d = dy.shape[-1]
    Win = P[pre + "attention_layer.in_proj_weight"]
    g = dy * c["m"]
    dh2 = g
    dx_f, dw1, db1, dw2, db2 = ffn_bwd(g, c["fc"], P[pre + "forward_layer.conv1.weight"][..., 0],
                                       P[pre + "forward_layer.conv2.weight"][..., 0])
    G[pre + "forward_layer.conv1.weight"] = dw1[..., None]
    G[pre + "forward_layer.conv1.bias"] = db1
    G[pre + "forward_layer.conv2.weight"] = dw2[..., None]
    G[pre + "forward_layer.conv2.bias"] = db2
    dh2 = dh2 + dx_f
    dh, G[pre + "forward_layernorm.weight"], G[pre + "forward_layernorm.bias"] = layer_norm_bwd(
        dh2, c["lnc2"], P[pre + "forward_layernorm.weight"])
    dQ = dh
    do, G[pre + "attention_layer.out_proj.weight"], G[pre + "attention_layer.out_proj.bias"] = linear_bwd(
        dh, c["o"], P[pre + "attention_layer.out_proj.weight"])
    # classifier: rec = log_softmax(z); dz = drec - softmax(z) * sum(drec)
    if drec is not None:
        sm = np.exp(c["rec"])
        dz = drec - sm * drec.sum(-1, keepdims=True)
        Ws = P[pre + "sparse.weight"]
        hd = Ws.shape[1]
        G[pre + "sparse.weight"] = dz.reshape(-1, H).T @ c["oh"].reshape(-1, hd)
        G[pre + "sparse.bias"] = dz.reshape(-1, H).sum(0)
        do = do + (dz @ Ws).reshape(do.shape)
    else:
        # num_heads == 1: sasrec/main.py:160 skips the independence loss, rec_ind is never consumed and
        # torch leaves sparse.{weight,bias}.grad = None
        G[pre + "sparse.weight"] = None
        G[pre + "sparse.bias"] = None
    dq, dk, dv = attention_bwd(do, c["ac"], H)
    dQq, dwq, dbq = linear_bwd(dq, c["Q"], Win[:d])
    dxk, dwk, dbk = linear_bwd(dk, c["x"], Win[d:2 * d])
    dxv, dwv, dbv = linear_bwd(dv, c["x"], Win[2 * d:])
    G[pre + "attention_layer.in_proj_weight"] = np.concatenate([dwq, dwk, dwv], 0)
    G[pre + "attention_layer.in_proj_bias"] = np.concatenate([dbq, dbk, dbv], 0)
    dQ = dQ + dQq
    dx, G[pre + "attention_layernorm.weight"], G[pre + "attention_layernorm.bias"] = layer_norm_bwd(
        dQ, c["lnc1"], P[pre + "attention_layernorm.weight"])
    return dx + dxk + dxv


def _mha_std(xq, xkv, P, pre, H, p, seed, site, b_offset, self_attn):
    """torch.nn.MultiheadAttention (packed in_proj) as called at sasrec/modules.py:669-672."""
    d = xq.shape[-1]
    W, b = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    q = linear(xq, W[:d], b[:d])
    k = linear(xkv, W[d:2 * d], b[d:2 * d])
    v = linear(xkv, W[2 * d:], b[2 * d:])
    o, ac = attention(q, k, v, H, True, p, seed, site, b_offset)
    a = linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])
    return a, dict(xq=xq, xkv=xkv, o=o, ac=ac)


def _mha_std_bwd(da, c, P, pre, H, G):
    d = da.shape[-1]
    W = P[pre + "in_proj_weight"]
    do, G[pre + "out_proj.weight"], G[pre + "out_proj.bias"] = linear_bwd(da, c["o"], P[pre + "out_proj.weight"])
    dq, dk, dv = attention_bwd(do, c["ac"], H)
    dxq, dwq, dbq = linear_bwd(dq, c["xq"], W[:d])
    dxk, dwk, dbk = linear_bwd(dk, c["xkv"], W[d:2 * d])
    dxv, dwv, dbv = linear_bwd(dv, c["xkv"], W[2 * d:])
    G[pre + "in_proj_weight"] = np.concatenate([dwq, dwk, dwv], 0)
    G[pre + "in_proj_bias"] = np.concatenate([dbq, dbk, dbv], 0)
    return dxq, dxk + dxv


def decoder_layer(x, enc, m, P, pre, H, p, seed, sites, b_offset):
    """DecoderLayer.forward, sasrec/modules.py:666-677: no residual around self-attention, cross-attention
    is causal (same mask, sasrec/model.py:69-70), out = (LN(x) + a2 + FFN(a2)) * mask."""
    D, lnc = layer_norm(x, P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"])
    a1, c1 = _mha_std(D, D, P, pre + "slf_attn.", H, p, seed, sites["slf"], b_offset, True)
    a2, c2 = _mha_std(a1, enc, P, pre + "enc_attn.", H, p, seed, sites["enc"], b_offset, False)
    f, fc = ffn(a2, P[pre + "pos_ffn.conv1.weight"][..., 0], P[pre + "pos_ffn.conv1.bias"],
                P[pre + "pos_ffn.conv2.weight"][..., 0], P[pre + "pos_ffn.conv2.bias"],
                p, seed, sites["ffn1"], sites["ffn2"], b_offset)
    y = (D + a2 + f) * m
    return y, dict(lnc=lnc, c1=c1, c2=c2, fc=fc, m=m)


def decoder_layer_bwd(dy, c, P, pre, H, G):
    g = dy * c["m"]
    da2_f, dw1, db1, dw2, db2 = ffn_bwd(g, c["fc"], P[pre + "pos_ffn.conv1.weight"][..., 0],
                                        P[pre + "pos_ffn.conv2.weight"][..., 0])
    G[pre + "pos_ffn.conv1.weight"] = dw1[..., None]
    G[pre + "pos_ffn.conv1.bias"] = db1
    G[pre + "pos_ffn.conv2.weight"] = dw2[..., None]
    G[pre + "pos_ffn.conv2.bias"] = db2
    da2 = g + da2_f
    da1, denc = _mha_std_bwd(da2, c["c2"], P, pre + "enc_attn.", H, G)
    dDq, dDkv = _mha_std_bwd(da1, c["c1"], P, pre + "slf_attn.", H, G)
    dD = g + dDq + dDkv
    dx, G[pre + "layer_norm.weight"], G[pre + "layer_norm.bias"] = layer_norm_bwd(dD, c["lnc"], P[pre + "layer_norm.weight"])
    return dx, denc


# ----------------------------------------------------------------------------------------------
# model
def forward(P, cfg, seq, dec, pos, neg, training=False, seed=0, b_offset=0):
    """SASRecADT.forward, sasrec/model.py:67-81.  Returns (pos_logits, neg_logits, enc_in[list],
    dec_out[list, reversed as sasrec/modules.py:756], rec[list], cache).  rec[i] is in TOKEN order
    (B, L, H, H); `rec_reference_order` converts to the row-permuted tensor the reference returns."""
    p = cfg.dropout if training else 0.0
    H, nl = cfg.num_heads, cfg.num_layers
    E, Pw = P["item_emb.weight"], P["pos_emb.weight"]
    x, ec = embed(seq, E, Pw, p, seed, SITE_EMB_SEQ, b_offset)
    m = ec[1]
    enc_in, recs, encc = [], [], []
    for i in range(nl):
        enc_in.append(x)
        x, rec, c = encoder_layer(x, m, P, "encoder.encoder_layers.%d." % i, H, p, seed, enc_sites(i), b_offset)
        recs.append(rec)
        encc.append(c)
    f, lncl = layer_norm(x, P["last_layernorm.weight"], P["last_layernorm.bias"])
    y, dc = embed(dec, E, Pw, p, seed, SITE_EMB_DEC, b_offset)
    md = dc[1]
    dec_out, decc = [], []
    for i in range(nl):
        y, c = decoder_layer(y, f, md, P, "decoder.decoder_layers.%d." % i, H, p, seed, dec_sites(i), b_offset)
        dec_out.append(y)
        decc.append(c)
    dec_out_rev = dec_out[::-1]
    pe, ne = E[pos], E[neg]
    pos_logits = (f * pe).sum(-1)
    neg_logits = (f * ne).sum(-1)
    cache = dict(seq=seq, dec=dec, pos=pos, neg=neg, ec=ec, dc=dc, encc=encc, decc=decc, lncl=lncl, f=f,
                 pe=pe, ne=ne, p=p)
    return pos_logits, neg_logits, enc_in, dec_out_rev, recs, cache


def rec_reference_order(rec):
    """sasrec/modules.py:518: `attn_output.view(bsz, tgt_len, H, hd)` reinterprets (L, B, E) memory, so the
    reference's row r = l*B + b holds token (b, l).  Token-order (B,L,H,H) -> reference-order (B,L,H,H)."""
    B, L = rec.shape[:2]
    return np.ascontiguousarray(rec.transpose(1, 0, 2, 3)).reshape(B, L, *rec.shape[2:])


def rec_token_order(rec_ref):
    B, L = rec_ref.shape[:2]
    return np.ascontiguousarray(rec_ref.reshape(L, B, *rec_ref.shape[2:]).transpose(1, 0, 2, 3))


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def loss_and_seeds(P, cfg, out, pos, lambdas1, lambdas2, weight_decay, norms=None):
    """Loss assembly of sasrec/main.py:146-170 and the gradients it sends back into the five model outputs.
    norms = (n_bce, n_mse, n_nll) lets a data-parallel shard use the GLOBAL normalisers."""
    pos_logits, neg_logits, enc_in, dec_out, recs = out[:5]
    dt = pos_logits.dtype
    B, L = pos.shape
    H, d = cfg.num_heads, cfg.hidden_units
    mk = (pos != 0)
    n_bce = float(mk.sum()) if norms is None else float(norms[0])
    n_mse = float(B * L * d) if norms is None else float(norms[1])
    n_nll = float(B * L * H) if norms is None else float(norms[2])
    parts = {}
    parts["bce_pos"] = float((softplus(-pos_logits) * mk).sum() / n_bce)  # BCEWithLogits, target 1 (:151)
    parts["bce_neg"] = float((softplus(neg_logits) * mk).sum() / n_bce)   # target 0 (:152)
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    d_pos = ((sig(pos_logits) - 1.0) * mk / n_bce).astype(dt)
    d_neg = (sig(neg_logits) * mk / n_bce).astype(dt)
    d_enc, d_dec = [], []
    parts["mse"] = []
    for i in range(len(enc_in)):  # :155-158 (mean over ALL B*L*d elements, padding included)
        diff = enc_in[i] - dec_out[i]
        parts["mse"].append(float((diff.astype(np.float64) ** 2).sum() / n_mse))
        g = (dt.type(2.0 * lambdas1[i] / n_mse) * diff).astype(dt)
        d_enc.append(g)
        d_dec.append(-g)
    d_rec = []
    parts["nll"] = []
    if H > 1:  # :160-169 ; lambdas2[i] uses the STALE loop variable i == last layer index
        lam2 = lambdas2[len(enc_in) - 1]
        eye = np.eye(H, dtype=dt)
        for l in range(len(recs)):
            parts["nll"].append(float(-(recs[l].astype(np.float64) * eye).sum() / n_nll))
            d_rec.append(np.broadcast_to(dt.type(-lam2 / n_nll) * eye, recs[l].shape).copy())
    else:
        lam2 = 0.0
        d_rec = [None] * len(recs)
    wnorm = float(np.sqrt((P["item_emb.weight"].astype(np.float64) ** 2).sum()))  # :170 (un-squared Frobenius)
    parts["wd"] = weight_decay * wnorm
    loss = parts["bce_pos"] + parts["bce_neg"] + sum(l1 * m for l1, m in zip(lambdas1, parts["mse"])) \
        + lam2 * sum(parts["nll"]) + parts["wd"]
    return loss, parts, (d_pos, d_neg, d_enc, d_dec, d_rec)


def backward(P, cfg, cache, seeds, weight_decay=0.0, add_wd=True):
    """Manual reverse pass for forward(); `seeds` = gradients w.r.t. the five outputs.  Returns a dict of
    gradients keyed like P (unused parameters -> None, as torch leaves them)."""
    d_pos, d_neg, d_enc, d_dec, d_rec = seeds
    H, nl = cfg.num_heads, cfg.num_layers
    E = P["item_emb.weight"]
    G = {}
    dE = np.zeros_like(E)
    f = cache["f"]
    df = d_pos[..., None] * cache["pe"] + d_neg[..., None] * cache["ne"]
    np.add.at(dE, cache["pos"], d_pos[..., None] * f)
    np.add.at(dE, cache["neg"], d_neg[..., None] * f)
    # decoder, last layer first.  dec_out (reversed) index j pairs with decoder layer nl-1-j.
    dy = None
    for i in reversed(range(nl)):
        up = d_dec[nl - 1 - i]
        dy = up if dy is None else dy + up
        dy, denc = decoder_layer_bwd(dy, cache["decc"][i], P, "decoder.decoder_layers.%d." % i, H, G)
        df = df + denc
    d_decemb = dy
    dx, G["last_layernorm.weight"], G["last_layernorm.bias"] = layer_norm_bwd(df, cache["lncl"], P["last_layernorm.weight"])
    for i in reversed(range(nl)):
        dx = encoder_layer_bwd(dx, d_rec[i], cache["encc"][i], P, "encoder.encoder_layers.%d." % i, H, G)
        dx = dx + d_enc[i]
    dP = np.zeros_like(P["pos_emb.weight"])
    for ids, g, c in ((cache["seq"], dx, cache["ec"]), (cache["dec"], d_decemb, cache["dc"])):
        keep, m = c
        g = g * m
        if keep is not None:
            g = g * keep * g.dtype.type(1.0 / (1.0 - rng.drop_prob(cache["p"])))
        dP[: g.shape[1]] += g.sum(0)
        np.add.at(dE, ids, g * E.dtype.type(E.shape[1] ** 0.5))
    if add_wd and weight_decay != 0.0:
        nrm = np.sqrt((E.astype(np.float64) ** 2).sum())
        dE = dE + (weight_decay / nrm * E).astype(E.dtype)
    G["item_emb.weight"] = dE
    G["pos_emb.weight"] = dP
    for name, _ in param_shapes(cfg):
        if is_unused(name, H):
            G[name] = None
    return G


def grad_norm(G):
    """torch.nn.utils.clip_grad_norm_ total norm (2-norm over all non-None grads), sasrec/main.py:172."""
    s = 0.0
    for g in G.values():
        if g is not None:
            s += float((g.astype(np.float64) ** 2).sum())
    return math.sqrt(s)


def clip_adam(P, G, state, lr=1e-3, betas=(0.9, 0.98), eps=1e-8, clip=5.0):
    """clip_grad_norm_(clip) then torch.optim.Adam step (sasrec/main.py:122,172-173).  In place on P/state."""
    tn = grad_norm(G)
    coef = min(1.0, clip / (tn + 1e-6))
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1 = 1.0 - betas[0] ** t
    bc2 = 1.0 - betas[1] ** t
    for k, g in G.items():
        if g is None:
            continue
        g = g * P[k].dtype.type(coef)
        m = state.setdefault("m." + k, np.zeros_like(P[k]))
        v = state.setdefault("v." + k, np.zeros_like(P[k]))
        m *= betas[0]
        m += (1 - betas[0]) * g
        v *= betas[1]
        v += (1 - betas[1]) * g * g
        denom = np.sqrt(v) / math.sqrt(bc2) + eps
        P[k] -= (lr / bc1) * m / denom
    return tn, coef


def train_step(P, cfg, state, batch, lambdas1, lambdas2, weight_decay, lr=1e-3, clip=5.0, seed=0,
               training=True, norms=None, b_offset=0):
    """One pass of the loop body sasrec/main.py:143-173."""
    seq, dec, pos, neg = batch
    out = forward(P, cfg, seq, dec, pos, neg, training=training, seed=seed, b_offset=b_offset)
    loss, parts, seeds = loss_and_seeds(P, cfg, out, pos, lambdas1, lambdas2, weight_decay, norms)
    G = backward(P, cfg, out[5], seeds, weight_decay)
    tn, coef = clip_adam(P, G, state, lr=lr, clip=clip)
    return loss, tn, G


def predict(P, cfg, seq, item_idx=None):
    """SASRecADT.predict, sasrec/model.py:83-97: encoder only, last position, dot with candidate rows
    (item_idx (B, C)) or the whole table (item_idx None == full=True)."""
    H, nl = cfg.num_heads, cfg.num_layers
    E = P["item_emb.weight"]
    x, ec = embed(seq, E, P["pos_emb.weight"], 0.0, 0, SITE_EMB_SEQ)
    m = ec[1]
    for i in range(nl):
        x, _, _ = encoder_layer(x, m, P, "encoder.encoder_layers.%d." % i, H, 0.0, 0, enc_sites(i), 0)
    f, _ = layer_norm(x, P["last_layernorm.weight"], P["last_layernorm.bias"])
    ff = f[:, -1, :]
    if item_idx is None:
        return ff @ E.T
    return np.einsum("bcd,bd->bc", E[item_idx], ff)


def rank_of_first(scores):
    """rank = argsort(argsort(-scores))[:, 0] (sasrec/utils.py:410): number of candidates ranked strictly
    ahead of column 0 under a stable sort of -scores (ties broken by column order)."""
    s0 = scores[:, :1]
    return (scores[:, 1:] > s0).sum(1).astype(np.int64)


def metrics_from_ranks(ranks, n_candidates, ks=(5, 10)):
    """evaluate_loader, sasrec/utils.py:395-428: HR@k, NDCG@k, AUC with candidates_size = 1 + n_candidates."""
    ranks = np.asarray(ranks, dtype=np.int64)
    n = float(len(ranks))
    ndcg, hr = {}, {}
    for k in ks:
        hit = ranks < k
        hr[k] = float(hit.sum()) / n
        ndcg[k] = float((1.0 / np.log2(ranks[hit] + 2.0)).sum()) / n
    r1 = ranks + 1
    S = 1 + n_candidates
    auc = float(np.mean((S - r1) / (S - 1)))
    return (ndcg, hr), auc
