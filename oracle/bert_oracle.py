"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the BERT4Rec-ADT hot path: forward, loss,
gradients (through oracle/tape.py), clip + Adam, predict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(adt_amd/) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  tools/gen_golden_wide.py imports the reference (/root/reference/bert4rec/model, PyTorch CPU) in
the build container and records tests/golden/bert_*.npz (eval-mode forward tensors, the dropout-0 training loss, every
parameter gradient, the clip norm, weights after 1 and 3 Adam steps, predict scores); tests/test_oracle_wide.py checks
this file against them.  Paths below are relative to /root/reference.
"""
import math

import numpy as np

from . import tape as tp

F32 = np.float32
LN_EPS = 1e-5          # bert4rec/model/modules.py:33-36,108-111 ; bert.py:56
MASK_FILL = -1e9       # bert4rec/model/modules.py:90-92
SITE_EMB_SEQ, SITE_EMB_DEC = 1, 2


def enc_sites(i):
    b = 16 + 8 * i
    return {"attn": b, "after_multi": b + 1, "final": b + 2}


def dec_sites(i):
    b = 128 + 8 * i
    return {"attn": b, "after_multi": b + 1, "src_attn": b + 2, "after_src": b + 3, "final": b + 4}


class Cfg:
    """Fields BertModel reads from args (bert4rec/model/bert.py:9-58, options.py:38-50)."""

    def __init__(self, item_num, maxlen, hidden_units, num_heads, num_layers, inner_units, dropout=0.0, attention_dropout=0.0,
                 type_vocab_size=2):
        self.item_num, self.maxlen, self.hidden_units, self.num_heads = item_num, maxlen, hidden_units, num_heads
        self.num_layers, self.inner_units, self.dropout, self.attention_dropout = num_layers, inner_units, dropout, attention_dropout
        self.type_vocab_size = type_vocab_size

    @property
    def vocab(self):
        return self.item_num + 100    # bert.py:20


_MHA = ("query_transfer", "key_transfer", "value_transfer", "out_transfer")


def param_shapes(cfg):
    """state_dict names and shapes of the reference's BertModel."""
    d, H, L, V, I = cfg.hidden_units, cfg.num_heads, cfg.maxlen, cfg.vocab, cfg.inner_units
    hd = d // H
    s = [("mask_bias", (V,)), ("item_emb.word_emb.weight", (V, d)), ("item_emb.pos_emb.weight", (L, d)),
         ("item_emb.sent_emb.weight", (cfg.type_vocab_size, d)), ("item_emb.layer_norm.weight", (d,)), ("item_emb.layer_norm.bias", (d,))]

    def mha(p):
        return [(p + "." + t + "." + w, (d, d) if w == "weight" else (d,)) for t in _MHA for w in ("weight", "bias")]

    def ln(p):
        return [(p + ".layer_norm.weight", (d,)), (p + ".layer_norm.bias", (d,))]

    def ffn(p):
        return [(p + ".fc1.weight", (I, d)), (p + ".fc1.bias", (I,)), (p + ".fc2.weight", (d, I)), (p + ".fc2.bias", (d,))]

    for i in range(cfg.num_layers):
        p = "encoder.encoder_layers.%d" % i
        s += mha(p + ".multi_head_attention") + ln(p + ".drop_residual_normalize_layer_after_multi") + ffn(p + ".ffn")
        s += ln(p + ".drop_residual_normalize_layer_final") + [(p + ".head_classifier.weight", (H, hd)), (p + ".head_classifier.bias", (H,))]
    for i in range(cfg.num_layers):
        p = "decoder.decoder_layers.%d" % i
        s += mha(p + ".dec_multi_head_attention") + ln(p + ".drop_residual_normalize_layer_after_multi")
        s += mha(p + ".src_dec_attention") + ln(p + ".drop_residual_normalize_layer_after_src_dec") + ffn(p + ".ffn")
        s += ln(p + ".drop_residual_normalize_layer_final")
    s += [("mask_trans_feat.weight", (d, d)), ("mask_trans_feat.bias", (d,)), ("mask_layer_norm.weight", (d,)), ("mask_layer_norm.bias", (d,))]
    return s


def init_params(cfg, seed=0, std=0.02):
    """BertTrainer's initialisation (bert4rec/trainer.py:29-37): N(0.01, std) on Linear/Embedding weights, LayerNorm 1/0,
    Linear biases 0; mask_bias zeros (bert.py:53-55).  numpy RNG, so both sides regenerate it without torch."""
    r = np.random.RandomState(seed)
    P = {}
    for name, shape in param_shapes(cfg):
        if name == "mask_bias" or name.endswith(".bias"):
            P[name] = np.zeros(shape, F32)
        elif "layer_norm" in name:
            P[name] = np.ones(shape, F32)
        else:
            P[name] = (0.01 + std * r.standard_normal(shape)).astype(F32)
    return P


# ----------------------------------------------------------------------------------------------------------------------
def _embed(V, cfg, ids, training, seed, site, b_offset):
    """BertEmbedding.forward (bert4rec/model/modules.py:41-48): LN(word[id] + pos[l] + sent[0]) -> dropout."""
    B, L = ids.shape
    d = cfg.hidden_units
    x = tp.embedding(V["item_emb.word_emb.weight"], ids, padding_idx=0)
    pos = tp.embedding(V["item_emb.pos_emb.weight"], np.tile(np.arange(L), (B, 1)), padding_idx=0)
    sent = tp.embedding(V["item_emb.sent_emb.weight"], np.zeros((B, L), np.int64), padding_idx=0)
    x = tp.add(tp.add(x, pos), sent)
    x = tp.layernorm(x, V["item_emb.layer_norm.weight"], V["item_emb.layer_norm.bias"], LN_EPS)
    return tp.dropout(x, cfg.dropout, seed, site, tp.idx_rows(B * L, d, b_offset * L).reshape(B, L, d), training)


def _mha(V, cfg, p, q_in, kv_in, key_valid, training, seed, site, b_offset):
    """MultiHeadAttention.forward (bert4rec/model/modules.py:76-101); key_valid (B, L) bool."""
    B, L, d = q_in.shape
    H = cfg.num_heads
    hd = d // H

    def split(x):
        return tp.transpose(tp.reshape(x, (B, L, H, hd)), (0, 2, 1, 3))

    q = split(tp.linear(q_in, V[p + ".query_transfer.weight"], V[p + ".query_transfer.bias"]))
    k = split(tp.linear(kv_in, V[p + ".key_transfer.weight"], V[p + ".key_transfer.bias"]))
    v = split(tp.linear(kv_in, V[p + ".value_transfer.weight"], V[p + ".value_transfer.bias"]))
    s = tp.div_const(tp.matmul(q, tp.transpose(k, (0, 1, 3, 2))), math.sqrt(hd))
    s = tp.masked_fill(s, np.broadcast_to(~key_valid[:, None, None, :], s.shape), MASK_FILL)
    w = tp.dropout(tp.softmax(s), cfg.attention_dropout, seed, site, tp.idx_attn(B, H, L, b_offset), training)
    o = tp.transpose(tp.matmul(w, v), (0, 2, 1, 3))                    # (B, L, H, hd)
    merged = tp.reshape(o, (B, L, d))
    return tp.linear(merged, V[p + ".out_transfer.weight"], V[p + ".out_transfer.bias"]), o


def _drn(V, cfg, p, out, prev, training, seed, site, b_offset):
    """DropResidualNormalizeLayer.forward (bert4rec/model/modules.py:113-117): LN(dropout(out) + prev)."""
    B, L, d = out.shape
    x = tp.dropout(out, cfg.attention_dropout, seed, site, tp.idx_rows(B * L, d, b_offset * L).reshape(B, L, d), training)
    return tp.layernorm(tp.add(x, prev), V[p + ".layer_norm.weight"], V[p + ".layer_norm.bias"], LN_EPS)


def _ffn(V, p, x):
    """FFN.forward (bert4rec/model/modules.py:135-139), act = GELU."""
    return tp.linear(tp.gelu(tp.linear(x, V[p + ".fc1.weight"], V[p + ".fc1.bias"])), V[p + ".fc2.weight"], V[p + ".fc2.bias"])


def _downstream(V, x):
    """BertModel.downstream (bert4rec/model/bert.py:80-90)."""
    h = tp.gelu(tp.linear(x, V["mask_trans_feat.weight"], V["mask_trans_feat.bias"]))
    h = tp.layernorm(h, V["mask_layer_norm.weight"], V["mask_layer_norm.bias"], LN_EPS)
    return tp.add(tp.linear(h, V["item_emb.word_emb.weight"]), V["mask_bias"])


def _encode(V, cfg, src, training, seed, b_offset):
    """BertModel.log2feats + Encoder.forward (bert.py:60-67, modules.py:208-216)."""
    valid = src > 0
    x = _embed(V, cfg, src, training, seed, SITE_EMB_SEQ, b_offset)
    enc_inputs, ind_outputs = [], []
    for i in range(cfg.num_layers):
        p = "encoder.encoder_layers.%d" % i
        st = enc_sites(i)
        enc_inputs.append(x)
        a, o = _mha(V, cfg, p + ".multi_head_attention", x, x, valid, training, seed, st["attn"], b_offset)
        h = _drn(V, cfg, p + ".drop_residual_normalize_layer_after_multi", a, x, training, seed, st["after_multi"], b_offset)
        x = _drn(V, cfg, p + ".drop_residual_normalize_layer_final", _ffn(V, p + ".ffn", h), h, training, seed, st["final"], b_offset)
        ind_outputs.append(tp.log_softmax(tp.linear(o, V[p + ".head_classifier.weight"], V[p + ".head_classifier.bias"])))
    return x, enc_inputs, ind_outputs, valid


def forward_vars(V, cfg, src, dec, training=False, seed=0, b_offset=0):
    """BertModel.forward (bert.py:92-108) on tape variables -> (logits, enc_inputs, dec_outputs reversed, ind_outputs)."""
    enc, enc_inputs, ind_outputs, src_valid = _encode(V, cfg, src, training, seed, b_offset)
    dvalid = dec > 0
    x = _embed(V, cfg, dec, training, seed, SITE_EMB_DEC, b_offset)
    dec_outputs = []
    for i in range(cfg.num_layers):      # DecoderLayer.forward (modules.py:297-325)
        p = "decoder.decoder_layers.%d" % i
        st = dec_sites(i)
        a, _ = _mha(V, cfg, p + ".dec_multi_head_attention", x, x, dvalid, training, seed, st["attn"], b_offset)
        g = _drn(V, cfg, p + ".drop_residual_normalize_layer_after_multi", a, x, training, seed, st["after_multi"], b_offset)
        a2, _ = _mha(V, cfg, p + ".src_dec_attention", g, enc, src_valid, training, seed, st["src_attn"], b_offset)
        g2 = _drn(V, cfg, p + ".drop_residual_normalize_layer_after_src_dec", a2, g, training, seed, st["after_src"], b_offset)
        x = _drn(V, cfg, p + ".drop_residual_normalize_layer_final", _ffn(V, p + ".ffn", g2), g2, training, seed, st["final"], b_offset)
        dec_outputs.append(x)
    dec_outputs.reverse()
    return _downstream(V, enc), enc_inputs, dec_outputs, ind_outputs


def as_vars(P):
    return {k: tp.leaf(v, k) for k, v in P.items()}


def forward(P, cfg, src, dec, training=False, seed=0, b_offset=0):
    out = forward_vars(as_vars(P), cfg, src, dec, training, seed, b_offset)
    return out[0].v, [t.v for t in out[1]], [t.v for t in out[2]], [t.v for t in out[3]]


def loss_and_grads(P, cfg, src, dec, labels, lambda1, lambda2, training=True, seed=0, b_offset=0, n_valid=None):
    """Loss assembly of BertTrainer.train (bert4rec/trainer.py:112-134) and its gradients.  n_valid overrides the CE
    normaliser (count of labels != 0) with the GLOBAL count when `src` is a data-parallel shard."""
    V = as_vars(P)
    logits, enc_in, dec_out, rec = forward_vars(V, cfg, src, dec, training, seed, b_offset)
    lab = np.asarray(labels).reshape(-1)
    ce = tp.cross_entropy(logits, lab, ignore_index=0)
    if n_valid is not None:
        ce = tp.scale(ce, float(max(int((lab != 0).sum()), 1)) / float(n_valid))
    loss = ce
    parts = {"ce": float(ce.v), "mse": [], "nll": []}
    if len(enc_in) == len(dec_out):
        for i in range(len(enc_in)):
            if lambda1[i] != 0:
                m = tp.mean(tp.square(tp.sub(enc_in[i], dec_out[i])))
                parts["mse"].append(float(m.v))
                loss = tp.add(loss, tp.scale(m, lambda1[i]))
    H = cfg.num_heads
    if H > 1:
        for l in range(len(rec)):
            if lambda2[l] != 0:
                diag = tp.index(rec[l], (slice(None), slice(None), np.arange(H), np.arange(H)))   # rec[b, l, h, h]
                n = tp.neg(tp.mean(diag))
                parts["nll"].append(float(n.v))
                loss = tp.add(loss, tp.scale(n, lambda2[l]))
    tp.backward(loss)
    G = {k: (V[k].g if V[k].g is not None else np.zeros_like(P[k])) for k in P}
    return float(loss.v), parts, G


def train_step(P, cfg, state, src, dec, labels, lambda1, lambda2, lr=1e-3, weight_decay=0.0, clip=5.0, training=True, seed=0):
    """One iteration of BertTrainer.train: loss.backward, clip_grad_norm_(clip), Adam(betas (0.9, 0.999), weight_decay)."""
    loss, parts, G = loss_and_grads(P, cfg, src, dec, labels, lambda1, lambda2, training, seed)
    tn = tp.clip_adam(P, G, state, lr, 0.9, 0.999, 1e-8, clip, weight_decay)
    return loss, tn


def predict(P, cfg, seqs, candidates):
    """BertModel.predict (bert.py:110-116): scores of the candidates at the last position."""
    V = as_vars(P)
    enc, _, _, _ = _encode(V, cfg, seqs, False, 0, 0)
    logits = _downstream(V, enc).v[:, -1, :]
    return np.take_along_axis(logits, np.asarray(candidates), axis=1)
