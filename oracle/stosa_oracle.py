"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the STOSA-ADT hot path: DisenDistSAModel.finetune,
the BPR / positive-vs-negative loss on Wasserstein distances, the reconstruction and independence terms, gradients
(through oracle/tape.py), Adam, and the full-sort distance scores.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(adt_amd/) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  tools/gen_golden_stosa.py imports the reference (/root/reference/stosa, PyTorch CPU) in the build
container and records tests/golden/stosa_*.npz (eval-mode finetune outputs, the dropout-0 training loss, every parameter
gradient incl. which are None, weights after 1 and 3 Adam steps, full-sort distances); tests/test_oracle_wide.py checks
this file against them.  Paths below are relative to /root/reference.
"""
import math

import numpy as np

from . import tape as tp

F32 = np.float32
LN_EPS = 1e-12                      # stosa/modules.py:87
MASK_ADD = F32(-2 ** 32 + 1)        # stosa/models.py:226,230 (float32: -4294967296)
SITE_EMB = {"seq_mean": 1, "seq_cov": 2, "dec_mean": 3, "dec_cov": 4}


def enc_sites(i):
    b = 16 + 8 * i
    return {"attn": b, "out_mean": b + 1, "out_cov": b + 2, "ffn_mean": b + 3, "ffn_cov": b + 4}


def dec_sites(i):
    b = 128 + 8 * i
    return {"attn": b, "out_mean": b + 1, "out_cov": b + 2, "ffn_mean": b + 3, "ffn_cov": b + 4}


class Cfg:
    """Fields DisenDistSAModel reads from args (stosa/models.py:166-180, main.py:26-38)."""

    def __init__(self, item_size, maxlen, hidden_units, num_heads, num_layers, dropout=0.0, attention_dropout=0.0, num_users=4,
                 pvn_weight=0.005):
        self.item_size, self.maxlen, self.hidden_units, self.num_heads, self.num_layers = item_size, maxlen, hidden_units, num_heads, num_layers
        self.dropout, self.attention_dropout, self.num_users, self.pvn_weight = dropout, attention_dropout, num_users, pvn_weight


_ATT = ("mean_query", "cov_query", "mean_key", "cov_key", "mean_value", "cov_value")


def param_shapes(cfg):
    """state_dict names and shapes of the reference's DisenDistSAModel."""
    d, H, L = cfg.hidden_units, cfg.num_heads, cfg.maxlen
    hd = d // H
    s = [("item_mean_embeddings.weight", (cfg.item_size, d)), ("item_cov_embeddings.weight", (cfg.item_size, d)),
         ("position_mean_embeddings.weight", (L, d)), ("position_cov_embeddings.weight", (L, d)), ("user_margins.weight", (cfg.num_users, 1))]

    def att(p):
        o = []
        for n in _ATT + ("mean_dense", "cov_dense"):
            o += [(p + "." + n + ".weight", (d, d)), (p + "." + n + ".bias", (d,))]
        return o + [(p + ".LayerNorm.weight", (d,)), (p + ".LayerNorm.bias", (d,))]

    def inter(p):
        return [(p + ".dense_1.weight", (4 * d, d)), (p + ".dense_1.bias", (4 * d,)), (p + ".dense_2.weight", (d, 4 * d)), (p + ".dense_2.bias", (d,)),
                (p + ".LayerNorm.weight", (d,)), (p + ".LayerNorm.bias", (d,))]

    for i in range(cfg.num_layers):
        p = "item_encoder.layer.%d" % i
        s += att(p + ".attention") + inter(p + ".mean_intermediate") + inter(p + ".cov_intermediate")
        s += [(p + ".mean_independence_layer.weight", (H, hd)), (p + ".mean_independence_layer.bias", (H,)),
              (p + ".cov_independence_layer.weight", (H, hd)), (p + ".cov_independence_layer.bias", (H,))]
    for i in range(cfg.num_layers):
        p = "item_decoder.layer.%d" % i
        s += att(p + ".dec_attention") + att(p + ".enc_attention") + inter(p + ".mean_intermediate") + inter(p + ".cov_intermediate")
    s += [("LayerNorm.weight", (d,)), ("LayerNorm.bias", (d,)), ("decLayerNorm.weight", (d,)), ("decLayerNorm.bias", (d,))]
    return s


def is_unused(name):
    """Parameters outside the loss graph (torch leaves grad=None and Adam skips them): the user margins and decLayerNorm
    are never used by finetune (stosa/models.py:212-260), and DistDecLayer discards its dec_attention output
    (stosa/modules.py:537-538)."""
    return name.startswith("user_margins") or name.startswith("decLayerNorm") or ".dec_attention." in name


def init_params(cfg, seed=0, std=0.02):
    """DisenDistSAModel.init_weights (stosa/models.py:262-272): N(0.01, std) on Linear/Embedding weights, LayerNorm 1/0,
    Linear biases 0 (perturbed by the golden generator so that every gradient is exercised)."""
    r = np.random.RandomState(seed)
    P = {}
    for name, shape in param_shapes(cfg):
        if name.endswith(".bias"):
            P[name] = np.zeros(shape, F32)
        elif "LayerNorm" in name:
            P[name] = np.ones(shape, F32)
        else:
            P[name] = (0.01 + std * r.standard_normal(shape)).astype(F32)
    return P


# ----------------------------------------------------------------------------------------------------------------------
def wasserstein_distance(m1, c1, m2, c2):
    """stosa/modules.py:22-28 (elementwise over the last axis)."""
    ret = tp.sum_(tp.square(tp.sub(m1, m2)), axis=-1)
    s1, s2 = tp.sqrt(tp.clamp_min(c1, 1e-24)), tp.sqrt(tp.clamp_min(c2, 1e-24))
    return tp.add(ret, tp.sum_(tp.square(tp.sub(s1, s2)), axis=-1))


def wasserstein_distance_matmul(m1, c1, m2, c2):
    """stosa/modules.py:30-43."""
    def T(x):
        return tp.transpose(x, tuple(range(x.v.ndim - 2)) + (x.v.ndim - 1, x.v.ndim - 2))
    m1_2 = tp.sum_(tp.square(m1), axis=-1, keepdims=True)
    m2_2 = tp.sum_(tp.square(m2), axis=-1, keepdims=True)
    ret = tp.add(tp.add(tp.scale(tp.matmul(m1, T(m2)), -2.0), m1_2), T(m2_2))
    c1_2 = tp.sum_(c1, axis=-1, keepdims=True)
    c2_2 = tp.sum_(c2, axis=-1, keepdims=True)
    s1, s2 = tp.sqrt(tp.clamp_min(c1, 1e-24)), tp.sqrt(tp.clamp_min(c2, 1e-24))
    cov_ret = tp.add(tp.add(tp.scale(tp.matmul(s1, T(s2)), -2.0), c1_2), T(c2_2))
    return tp.add(ret, cov_ret)


def _rows_idx(B, L, d, b_offset):
    return tp.idx_rows(B * L, d, b_offset * L).reshape(B, L, d)


def _embed(V, cfg, ids, which, training, seed, b_offset):
    """add_position_mean_embedding / add_position_cov_embedding (stosa/models.py:183-210)."""
    B, L = ids.shape
    d = cfg.hidden_units
    x = tp.add(tp.embedding(V["item_%s_embeddings.weight" % which], ids, padding_idx=0),
               tp.embedding(V["position_%s_embeddings.weight" % which], np.tile(np.arange(L), (B, 1))))
    x = tp.layernorm(x, V["LayerNorm.weight"], V["LayerNorm.bias"], LN_EPS)
    return x


def _attention(V, cfg, p, mean_q, cov_q, mean_kv, cov_kv, mask_add, training, seed, st, b_offset):
    """DistAttention.forward / DistEDAttention.forward (stosa/modules.py:222-275, 311-361)."""
    B, L, d = mean_q.shape
    H = cfg.num_heads
    hd = d // H

    def split(x):
        return tp.transpose(tp.reshape(x, (B, L, H, hd)), (0, 2, 1, 3))

    def lin(x, n):
        return tp.linear(x, V[p + "." + n + ".weight"], V[p + "." + n + ".bias"])
    mq, mk, mv = split(lin(mean_q, "mean_query")), split(lin(mean_kv, "mean_key")), split(lin(mean_kv, "mean_value"))
    cq = split(tp.elu(lin(cov_q, "cov_query"), True))
    ck = split(tp.elu(lin(cov_kv, "cov_key"), True))
    cv = split(tp.elu(lin(cov_kv, "cov_value"), True))
    s = tp.div_const(tp.neg(wasserstein_distance_matmul(mq, cq, mk, ck)), math.sqrt(hd))
    s = tp.add_const(s, mask_add)
    probs = tp.dropout(tp.softmax(s), cfg.attention_dropout, seed, st["attn"], tp.idx_attn(B, H, L, b_offset), training)
    ctx_m = tp.transpose(tp.matmul(probs, mv), (0, 2, 1, 3))                     # (B, L, H, hd)
    ctx_c = tp.transpose(tp.matmul(tp.square(probs), cv), (0, 2, 1, 3))
    lw, lb = V[p + ".LayerNorm.weight"], V[p + ".LayerNorm.bias"]
    hm = tp.dropout(lin(tp.reshape(ctx_m, (B, L, d)), "mean_dense"), cfg.dropout, seed, st["out_mean"], _rows_idx(B, L, d, b_offset), training)
    hm = tp.layernorm(tp.add(hm, mean_q), lw, lb, LN_EPS)
    hc = tp.dropout(lin(tp.reshape(ctx_c, (B, L, d)), "cov_dense"), cfg.dropout, seed, st["out_cov"], _rows_idx(B, L, d, b_offset), training)
    hc = tp.layernorm(tp.add(hc, cov_q), lw, lb, LN_EPS)
    return hm, hc, ctx_m, ctx_c


def _intermediate(V, cfg, p, x, training, seed, site, b_offset):
    """DistIntermediate.forward (stosa/modules.py:485-494)."""
    B, L, d = x.shape
    h = tp.elu(tp.linear(x, V[p + ".dense_1.weight"], V[p + ".dense_1.bias"]))
    h = tp.linear(h, V[p + ".dense_2.weight"], V[p + ".dense_2.bias"])
    h = tp.dropout(h, cfg.dropout, seed, site, _rows_idx(B, L, d, b_offset), training)
    return tp.layernorm(tp.add(h, x), V[p + ".LayerNorm.weight"], V[p + ".LayerNorm.bias"], LN_EPS)


def _mask(ids):
    """extended attention mask (stosa/models.py:214-231): 0 where key j is a real item and j <= i, else -2^32."""
    B, L = ids.shape
    ok = (ids > 0)[:, None, None, :] & np.tril(np.ones((L, L), bool))[None, None]
    return ((1.0 - ok.astype(F32)) * MASK_ADD).astype(F32)


def finetune_vars(V, cfg, input_ids, dec_ids, training=False, seed=0, b_offset=0):
    """DisenDistSAModel.finetune (stosa/models.py:212-260) -> (mean_out, cov_out, enc_inputs, enc_recs, dec_outputs)."""
    B, L = input_ids.shape
    d = cfg.hidden_units
    mask = _mask(input_ids)

    def emb(ids, which, site):
        x = _embed(V, cfg, ids, which, training, seed, b_offset)
        x = tp.dropout(x, cfg.dropout, seed, site, _rows_idx(B, L, d, b_offset), training)
        return tp.elu(x, which == "cov")
    m, c = emb(input_ids, "mean", SITE_EMB["seq_mean"]), emb(input_ids, "cov", SITE_EMB["seq_cov"])
    dm, dc = emb(dec_ids, "mean", SITE_EMB["dec_mean"]), emb(dec_ids, "cov", SITE_EMB["dec_cov"])
    enc_inputs, enc_recs = [], []
    for i in range(cfg.num_layers):      # DistLayer.forward (modules.py:518-525), DistSAEncoder.forward (:551-565)
        p = "item_encoder.layer.%d" % i
        st = enc_sites(i)
        enc_inputs.append([m, c])
        hm, hc, rm, rc = _attention(V, cfg, p + ".attention", m, c, m, c, mask, training, seed, st, b_offset)
        m = _intermediate(V, cfg, p + ".mean_intermediate", hm, training, seed, st["ffn_mean"], b_offset)
        c = tp.elu(_intermediate(V, cfg, p + ".cov_intermediate", hc, training, seed, st["ffn_cov"], b_offset), True)
        rm = tp.log_softmax(tp.linear(rm, V[p + ".mean_independence_layer.weight"], V[p + ".mean_independence_layer.bias"]))
        rc = tp.log_softmax(tp.linear(rc, V[p + ".cov_independence_layer.weight"], V[p + ".cov_independence_layer.bias"]))
        enc_recs.append([rm, rc])
    dec_outputs = []
    for i in range(cfg.num_layers):      # DistDecLayer.forward (modules.py:535-541): dec_attention is computed and discarded
        p = "item_decoder.layer.%d" % i
        st = dec_sites(i)
        hm, hc, _, _ = _attention(V, cfg, p + ".enc_attention", dm, dc, m, c, mask, training, seed, st, b_offset)
        dm = _intermediate(V, cfg, p + ".mean_intermediate", hm, training, seed, st["ffn_mean"], b_offset)
        dc = tp.elu(_intermediate(V, cfg, p + ".cov_intermediate", hc, training, seed, st["ffn_cov"], b_offset), True)
        dec_outputs.append([dm, dc])
    return m, c, enc_inputs, enc_recs, dec_outputs


def as_vars(P):
    return {k: tp.leaf(v, k) for k, v in P.items()}


def finetune(P, cfg, input_ids, dec_ids, training=False, seed=0):
    m, c, ei, er, do = finetune_vars(as_vars(P), cfg, input_ids, dec_ids, training, seed)
    return m.v, c.v, [[a.v, b.v] for a, b in ei], [[a.v, b.v] for a, b in er], [[a.v, b.v] for a, b in do]


def bpr_terms(V, cfg, seq_mean, seq_cov, pos_ids, neg_ids, n_target=None):
    """DistSAModelTrainer.bpr_optimization (stosa/trainer.py:358-391) -> (loss, auc, pvn_loss) tape variables."""
    d = cfg.hidden_units
    pos_m = tp.reshape(tp.embedding(V["item_mean_embeddings.weight"], pos_ids, padding_idx=0), (-1, d))
    pos_c = tp.reshape(tp.elu(tp.embedding(V["item_cov_embeddings.weight"], pos_ids, padding_idx=0), True), (-1, d))
    neg_m = tp.reshape(tp.embedding(V["item_mean_embeddings.weight"], neg_ids, padding_idx=0), (-1, d))
    neg_c = tp.reshape(tp.elu(tp.embedding(V["item_cov_embeddings.weight"], neg_ids, padding_idx=0), True), (-1, d))
    sm, sc = tp.reshape(seq_mean, (-1, d)), tp.reshape(seq_cov, (-1, d))
    pos_l = wasserstein_distance(sm, sc, pos_m, pos_c)
    neg_l = wasserstein_distance(sm, sc, neg_m, neg_c)
    pvn = wasserstein_distance(pos_m, pos_c, neg_m, neg_c)
    ist = (np.asarray(pos_ids).reshape(-1) > 0).astype(F32)
    n = float(ist.sum()) if n_target is None else float(n_target)
    x = tp.add_const(tp.sub(neg_l, pos_l), 1e-24)
    loss = tp.div_const(tp.sum_(tp.mul_mask(tp.neg(tp.log(tp.sigmoid(x))), ist)), n)
    pvn_loss = tp.scale(tp.div_const(tp.sum_(tp.mul_mask(tp.clamp_min(tp.sub(pos_l, pvn), 0.0), ist)), n), cfg.pvn_weight)
    auc = float((((np.sign(neg_l.v - pos_l.v) + 1) / 2) * ist).sum() / n)
    return loss, auc, pvn_loss


def loss_and_grads(P, cfg, input_ids, dec_ids, pos_ids, neg_ids, lambda1, lambda2, training=True, seed=0, b_offset=0, n_target=None,
                   norms_scale=1):
    """Loss assembly of DistSAModelTrainer.iteration (stosa/trainer.py:534-556) and its gradients (None for unused params).
    n_target / norms_scale give the GLOBAL normalisers when the inputs are a data-parallel shard."""
    V = as_vars(P)
    m, c, enc_in, enc_rec, dec_out = finetune_vars(V, cfg, input_ids, dec_ids, training, seed, b_offset)
    bpr, auc, pvn = bpr_terms(V, cfg, m, c, pos_ids, neg_ids, n_target)
    loss = bpr
    dec_rev = dec_out[::-1]
    H = cfg.num_heads
    parts = {"bpr": float(bpr.v), "pvn": float(pvn.v), "auc": auc}
    for l in range(cfg.num_layers):
        for t in (0, 1):
            loss = tp.add(loss, tp.scale(tp.div_const(tp.mean(tp.square(tp.sub(enc_in[l][t], dec_rev[l][t]))), norms_scale), lambda1[l]))
    for l in range(cfg.num_layers):
        for t in (0, 1):
            diag = tp.index(enc_rec[l][t], (slice(None), slice(None), np.arange(H), np.arange(H)))
            loss = tp.add(loss, tp.scale(tp.div_const(tp.neg(tp.mean(diag)), norms_scale), lambda2[l]))
    loss = tp.add(loss, pvn)
    tp.backward(loss)
    G = {k: (None if is_unused(k) else (V[k].g if V[k].g is not None else np.zeros_like(P[k]))) for k in P}
    return float(loss.v), parts, G


def train_step(P, cfg, state, input_ids, dec_ids, pos_ids, neg_ids, lambda1, lambda2, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0,
               training=True, seed=0):
    """One iteration: loss.backward(); Adam(lr, betas, weight_decay).step() -- no gradient clipping (trainer.py:557-559)."""
    loss, parts, G = loss_and_grads(P, cfg, input_ids, dec_ids, pos_ids, neg_ids, lambda1, lambda2, training, seed)
    tp.clip_adam(P, G, state, lr, betas[0], betas[1], 1e-8, None, weight_decay)
    return loss, parts


def predict_full(P, cfg, input_ids, dec_ids):
    """Full-sort scores (stosa/trainer.py:583-595 + dist_predict_full :464-479): distance of the last state to every item."""
    V = as_vars(P)
    m, c, _, _, _ = finetune_vars(V, cfg, input_ids, dec_ids, False, 0)
    sm, sc = tp.const(m.v[:, -1, :]), tp.const(c.v[:, -1, :])
    return wasserstein_distance_matmul(sm, sc, V["item_mean_embeddings.weight"], tp.elu(V["item_cov_embeddings.weight"], True)).v
