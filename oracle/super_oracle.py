"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the SASRec-ADT supernet: SuperSASRecModel with its
4-candidate layer mixing, the warm-up loss of the evolutionary search, gradients (through oracle/tape.py), clip + Adam.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(adt_amd/) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  tools/gen_golden_super.py imports the reference (/root/reference/sasrec supersasrec.py,
super_modules.py, base_super_modules.py) and records tests/golden/super_*.npz; tests/test_oracle_wide.py checks this file
against them.  Paths below are relative to /root/reference.
"""
import math

import numpy as np

from . import tape as tp

F32 = np.float32
LN_EPS = 1e-8
SITE_EMB_SEQ, SITE_EMB_DEC = 1, 2
CAND_SITE = 4096     # dropout sites of the k-th mixed candidate are offset by CAND_SITE * (k + 1)


def enc_sites(i, k):
    b = 16 + 8 * i + CAND_SITE * (k + 1)
    return {"attn": b, "ffn1": b + 1, "ffn2": b + 2}


def dec_sites(i, k):
    b = 128 + 8 * i + CAND_SITE * (k + 1)
    return {"slf": b, "enc": b + 1, "ffn1": b + 2, "ffn2": b + 3}


class Cfg:
    def __init__(self, item_num, maxlen, hidden_units, num_heads, num_layers, rec_choice, ind_choice, dropout=0.0):
        self.item_num, self.maxlen, self.hidden_units, self.num_heads, self.num_layers = item_num, maxlen, hidden_units, num_heads, num_layers
        self.rec_choice, self.ind_choice = np.asarray(rec_choice, np.float64), np.asarray(ind_choice, np.float64)
        self.dropout = dropout

    @property
    def block(self):
        return len(self.rec_choice) * len(self.ind_choice)     # super_modules.py:20


_ENC = [("attention_layernorm.weight", "d"), ("attention_layernorm.bias", "d"), ("attention_layer.in_proj_weight", "3dd"), ("attention_layer.in_proj_bias", "3d"),
        ("attention_layer.out_proj.weight", "dd"), ("attention_layer.out_proj.bias", "d"), ("forward_layernorm.weight", "d"), ("forward_layernorm.bias", "d"),
        ("forward_layer.conv1.weight", "dd1"), ("forward_layer.conv1.bias", "d"), ("forward_layer.conv2.weight", "dd1"), ("forward_layer.conv2.bias", "d"),
        ("sparse.weight", "Hh"), ("sparse.bias", "H")]
_DEC = [("layer_norm.weight", "d"), ("layer_norm.bias", "d"), ("slf_attn.in_proj_weight", "3dd"), ("slf_attn.in_proj_bias", "3d"),
        ("slf_attn.out_proj.weight", "dd"), ("slf_attn.out_proj.bias", "d"), ("enc_attn.in_proj_weight", "3dd"), ("enc_attn.in_proj_bias", "3d"),
        ("enc_attn.out_proj.weight", "dd"), ("enc_attn.out_proj.bias", "d"), ("pos_ffn.conv1.weight", "dd1"), ("pos_ffn.conv1.bias", "d"),
        ("pos_ffn.conv2.weight", "dd1"), ("pos_ffn.conv2.bias", "d"), ("pos_ffn_layernorm.weight", "d"), ("pos_ffn_layernorm.bias", "d")]


def _shape(code, d, H):
    return {"d": (d,), "3d": (3 * d,), "dd": (d, d), "3dd": (3 * d, d), "dd1": (d, d, 1), "Hh": (H, d // H), "H": (H,)}[code]


def param_shapes(cfg):
    d, H = cfg.hidden_units, cfg.num_heads
    s = [("item_emb.weight", (cfg.item_num + 1, d)), ("pos_emb.weight", (cfg.maxlen, d))]
    for i in range(cfg.num_layers):
        for c in range(cfg.block):
            s += [("encoder.encoder_layers.%d.%d.%s" % (i, c, n), _shape(code, d, H)) for n, code in _ENC]
    for i in range(cfg.num_layers):
        for c in range(cfg.block):
            s += [("decoder.decoder_layers.%d.%d.%s" % (i, c, n), _shape(code, d, H)) for n, code in _DEC]
    return s


def init_params(cfg, seed=0):
    """xavier_normal_-like weights for >= 2-D tensors, small random 1-D tensors (so that every gradient is exercised),
    LayerNorm weights near 1 -- numpy RNG, regenerated identically by the golden generator and the tests."""
    r = np.random.RandomState(seed)
    P = {}
    for name, shape in param_shapes(cfg):
        if len(shape) >= 2:
            fan = shape[0] + shape[1] * (shape[2] if len(shape) > 2 else 1)
            P[name] = (r.standard_normal(shape) * math.sqrt(2.0 / fan)).astype(F32)
        elif "norm.weight" in name:
            P[name] = (1.0 + 0.05 * r.standard_normal(shape)).astype(F32)
        else:
            P[name] = (0.05 * r.standard_normal(shape)).astype(F32)
    return P


# ---- candidate selection: BaseSuperModule._get_position / _get_shared (sasrec/base_super_modules.py:15-40) -------------
def get_position(weight, choice):
    i1 = int(np.where(choice > weight)[0][0])
    i0 = i1 - 1
    p0 = (weight - choice[i0]) / (choice[i1] - choice[i0])
    return i0, i1, p0, 1 - p0


def get_shared(cfg, cand):
    """[(4 layer indices, 4 weights)] per depth; note the reference strides BOTH index pairs by rec_size (:33-36)."""
    out = []
    rs = len(cfg.rec_choice)
    for i in range(len(cand) // 2):
        i0, i1, p0, p1 = get_position(cand[2 * i], cfg.rec_choice)
        i2, i3, p2, p3 = get_position(cand[2 * i + 1], cfg.ind_choice)
        out.append(((i0 * rs + i2, i1 * rs + i2, i0 * rs + i3, i1 * rs + i3), (p1 * p3, p0 * p3, p1 * p2, p0 * p2)))
    return out


def get_weight(choices, prob):
    """SearcherEvolution._get_weight (sasrec/evolution.py:123-137): piecewise-linear interpolation of the choice table."""
    split = 1 / (len(choices) - 1)
    idx = 0
    while prob > split:
        idx += 1
        prob -= split
    rd = prob / split
    return choices[idx] * (1 - rd) + choices[idx + 1] * rd


def cand_to_block(cfg, cand):
    """SearcherEvolution._set_choice (evolution.py:139-153): probabilities -> (block_cand for set_choice, rec_weights, ind_weights)."""
    block, rec_w, ind_w = [], [], []
    for i in range(0, len(cand), 2):
        rw, iw = get_weight(cfg.rec_choice, cand[i]), get_weight(cfg.ind_choice, cand[i + 1])
        rec_w.append(rw)
        ind_w.append(iw)
        block += [rw, iw]
    return np.array(block), rec_w, ind_w


# ---- layers -------------------------------------------------------------------------------------------------------------
def _rows_idx(B, L, d, b_offset):
    return tp.idx_rows(B * L, d, b_offset * L).reshape(B, L, d)


def _embed(V, cfg, ids, training, seed, site, b_offset):
    """SuperSASRecModel.log2feats / decode embedding part (sasrec/supersasrec.py:45-53, 64-71)."""
    B, L = ids.shape
    d = cfg.hidden_units
    x = tp.scale(tp.embedding(V["item_emb.weight"], ids, padding_idx=0), math.sqrt(d))
    x = tp.add(x, tp.embedding(V["pos_emb.weight"], np.tile(np.arange(L), (B, 1))))
    x = tp.dropout(x, cfg.dropout, seed, site, _rows_idx(B, L, d, b_offset), training)
    return tp.mul_mask(x, (ids != 0).astype(F32)[:, :, None])


def _mha(V, cfg, p, q_in, kv_in, training, seed, site, b_offset, packed_q_from_kv):
    """MultiheadAttentionADT / torch.nn.MultiheadAttention core with the causal float mask (sasrec/modules.py:270-527)."""
    B, L, d = q_in.shape
    H = cfg.num_heads
    hd = d // H
    W, b = V[p + ".in_proj_weight"], V[p + ".in_proj_bias"]
    Wq, Wk, Wv = tp.index(W, slice(0, d)), tp.index(W, slice(d, 2 * d)), tp.index(W, slice(2 * d, 3 * d))
    bq, bk, bv = tp.index(b, slice(0, d)), tp.index(b, slice(d, 2 * d)), tp.index(b, slice(2 * d, 3 * d))

    def split(x):
        return tp.transpose(tp.reshape(x, (B, L, H, hd)), (0, 2, 1, 3))
    q = tp.scale(split(tp.linear(q_in, Wq, bq)), 1.0 / math.sqrt(hd))
    k, v = split(tp.linear(kv_in, Wk, bk)), split(tp.linear(kv_in, Wv, bv))
    s = tp.matmul(q, tp.transpose(k, (0, 1, 3, 2)))
    s = tp.masked_fill(s, np.broadcast_to(np.triu(np.ones((L, L), bool), 1)[None, None], s.shape), -np.inf)
    w = tp.dropout(tp.softmax(s), cfg.dropout, seed, site, tp.idx_attn(B, H, L, b_offset), training)
    o = tp.transpose(tp.matmul(w, v), (0, 2, 1, 3))         # (B, L, H, hd)
    out = tp.linear(tp.reshape(o, (B, L, d)), V[p + ".out_proj.weight"], V[p + ".out_proj.bias"])
    return out, o


def _conv(V, name):
    W = V[name + ".weight"]
    return tp.reshape(W, W.v.shape[:2]), V[name + ".bias"]


def _ffn(V, cfg, p, x, training, seed, s1, s2, b_offset):
    """PointWiseFeedForward (sasrec/modules.py:618-633): x + drop(conv2(relu(drop(conv1(x)))))."""
    B, L, d = x.shape
    W1, b1 = _conv(V, p + ".conv1")
    W2, b2 = _conv(V, p + ".conv2")
    h = tp.relu(tp.dropout(tp.linear(x, W1, b1), cfg.dropout, seed, s1, _rows_idx(B, L, d, b_offset), training))
    h = tp.dropout(tp.linear(h, W2, b2), cfg.dropout, seed, s2, _rows_idx(B, L, d, b_offset), training)
    return tp.add(h, x)


def enc_layer(V, cfg, p, x, mask, training, seed, st, b_offset):
    """EncoderLayer.forward (sasrec/modules.py:644-655) -> (seqs, log-probabilities of the head classifier)."""
    Q = tp.layernorm(x, V[p + ".attention_layernorm.weight"], V[p + ".attention_layernorm.bias"], LN_EPS)
    a, o = _mha(V, cfg, p + ".attention_layer", Q, x, training, seed, st["attn"], b_offset, False)
    rec = tp.log_softmax(tp.linear(o, V[p + ".sparse.weight"], V[p + ".sparse.bias"]))
    h = tp.layernorm(tp.add(Q, a), V[p + ".forward_layernorm.weight"], V[p + ".forward_layernorm.bias"], LN_EPS)
    y = _ffn(V, cfg, p + ".forward_layer", h, training, seed, st["ffn1"], st["ffn2"], b_offset)
    return tp.mul_mask(y, mask), rec


def dec_layer(V, cfg, p, x, enc, mask, training, seed, st, b_offset):
    """DecoderLayer.forward (sasrec/modules.py:666-677)."""
    D = tp.layernorm(x, V[p + ".layer_norm.weight"], V[p + ".layer_norm.bias"], LN_EPS)
    a1, _ = _mha(V, cfg, p + ".slf_attn", D, D, training, seed, st["slf"], b_offset, True)
    a2, _ = _mha(V, cfg, p + ".enc_attn", a1, enc, training, seed, st["enc"], b_offset, False)
    y = tp.add(D, _ffn(V, cfg, p + ".pos_ffn", a2, training, seed, st["ffn1"], st["ffn2"], b_offset))
    return tp.mul_mask(y, mask)


def encode_vars(V, cfg, shared, seq, training=False, seed=0, b_offset=0):
    """log2feats + SuperEncoder.forward (supersasrec.py:45-61, super_modules.py:35-50) -> (feats, enc_inputs, rec log-probs)."""
    x = _embed(V, cfg, seq, training, seed, SITE_EMB_SEQ, b_offset)
    smask = (seq != 0).astype(F32)[:, :, None]
    enc_in, recs = [], []
    for i, (idxs, ws) in enumerate(shared):
        enc_in.append(x)
        outs, inds = [], []
        for k, (idx, w) in enumerate(zip(idxs, ws)):
            y, rec = enc_layer(V, cfg, "encoder.encoder_layers.%d.%d" % (i, idx), x, smask, training, seed, enc_sites(i, k), b_offset)
            outs.append(tp.scale(y, w))
            inds.append(tp.scale(rec, w))
        x = outs[0]
        r = inds[0]
        for k in range(1, 4):
            x, r = tp.add(x, outs[k]), tp.add(r, inds[k])
        recs.append(tp.log_softmax(r))       # log_softmax of the mixed log-probabilities (super_modules.py:49)
    return x, enc_in, recs                    # no last_layernorm in the supernet


def forward_vars(V, cfg, block_cand, seq, dec, pos, neg, training=False, seed=0, b_offset=0):
    """SuperSASRecModel.forward (sasrec/supersasrec.py:81-94) with SuperEncoder/SuperDecoder.forward (super_modules.py:35-50,
    :74-85) for the candidate set by set_choice(block_cand)."""
    shared = get_shared(cfg, block_cand)
    feats, enc_in, recs = encode_vars(V, cfg, shared, seq, training, seed, b_offset)
    y = _embed(V, cfg, dec, training, seed, SITE_EMB_DEC, b_offset)
    dmask = (dec != 0).astype(F32)[:, :, None]
    dec_out = []
    for i, (idxs, ws) in enumerate(shared):
        outs = [tp.scale(dec_layer(V, cfg, "decoder.decoder_layers.%d.%d" % (i, idx), y, feats, dmask, training, seed, dec_sites(i, k), b_offset), w)
                for k, (idx, w) in enumerate(zip(idxs, ws))]
        y = outs[0]
        for k in range(1, 4):
            y = tp.add(y, outs[k])
        dec_out.append(y)
    dec_out.reverse()
    pe = tp.embedding(V["item_emb.weight"], pos, padding_idx=0)
    ne = tp.embedding(V["item_emb.weight"], neg, padding_idx=0)
    return tp.sum_(tp.mul(feats, pe), axis=-1), tp.sum_(tp.mul(feats, ne), axis=-1), enc_in, dec_out, recs


def as_vars(P):
    return {k: tp.leaf(v, k) for k, v in P.items()}


def forward(P, cfg, block_cand, seq, dec, pos, neg):
    pl, nl, ei, do, rc = forward_vars(as_vars(P), cfg, block_cand, seq, dec, pos, neg)
    return pl.v, nl.v, [t.v for t in ei], [t.v for t in do], [t.v for t in rc]


def loss_and_grads(P, cfg, cand, seq, dec, pos, neg, training=True, seed=0):
    """The loop body of SearcherEvolution._train_warmup (sasrec/evolution.py:286-316) for candidate `cand` (probabilities);
    the independence weight is ind_weights[i] with the stale index i of the reconstruction loop (:313)."""
    block, rec_w, ind_w = cand_to_block(cfg, cand)
    V = as_vars(P)
    pl, nl, enc_in, dec_out, recs = forward_vars(V, cfg, block, seq, dec, pos, neg, training, seed)
    ist = (pos != 0).astype(F32)
    n = float(ist.sum())
    bce_p = tp.div_const(tp.sum_(tp.mul_mask(tp.neg(tp.log(tp.sigmoid(pl))), ist)), n)              # BCEWithLogits(target 1)
    bce_n = tp.div_const(tp.sum_(tp.mul_mask(tp.neg(tp.log(tp.sigmoid(tp.neg(nl)))), ist)), n)      # BCEWithLogits(target 0)
    loss = tp.add(bce_p, bce_n)
    i = 0
    for i in range(len(enc_in)):
        loss = tp.add(loss, tp.scale(tp.mean(tp.square(tp.sub(enc_in[i], dec_out[i]))), rec_w[i]))
    H = cfg.num_heads
    if H > 1:
        for l in range(len(recs)):
            diag = tp.index(recs[l], (slice(None), slice(None), np.arange(H), np.arange(H)))
            loss = tp.add(loss, tp.scale(tp.neg(tp.mean(diag)), ind_w[i]))
    tp.backward(loss)
    G = {k: V[k].g for k in P}      # None for the candidates that were not mixed in and for pos_ffn_layernorm
    return float(loss.v), G


def train_step(P, cfg, state, cand, seq, dec, pos, neg, lr=1e-3, weight_decay=0.0, clip=5.0, training=True, seed=0):
    """loss.backward(); clip_grad_norm_(clip); Adam(betas (0.9, 0.999), weight_decay) (evolution.py:109, 314-316)."""
    loss, G = loss_and_grads(P, cfg, cand, seq, dec, pos, neg, training, seed)
    tn = tp.clip_adam(P, G, state, lr, 0.9, 0.999, 1e-8, clip, weight_decay)
    return loss, tn


def predict(P, cfg, block_cand, seq, item_indices):
    """SuperSASRecModel.predict (supersasrec.py:96-111): candidate scores from the last position of the encoder output."""
    V = as_vars(P)
    feats, _, _ = encode_vars(V, cfg, get_shared(cfg, block_cand), seq)
    final = feats.v[:, -1, :]
    return np.einsum("bcd,bd->bc", P["item_emb.weight"][np.asarray(item_indices)], final)
