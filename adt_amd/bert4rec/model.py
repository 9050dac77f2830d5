"""BertModel on the MI355X hot path -- drop-in for the reference's bert4rec/model/bert.py:BertModel (BERT4Rec-ADT:
bidirectional masked-item encoder + reconstruction decoder + per-head independence classifiers).

Same constructor (`BertModel(usernum, itemnum, args)`), `forward(src_ids, dec_ids, seq_pos_ids, seq_sent_ids, deq_pos_ids,
deq_sent_ids)` 4-tuple, `predict(user_ids, seqs, seq_pos_ids, seq_sent_ids, candidates)` and state_dict names/shapes
(SURVEY.md 8b), but every parameter is a view into one flat fp32 buffer and all arithmetic runs in libadt_hip.so:
dense layers on MFMA (adt_gemm.cuh), masked bidirectional attention (adt_attn_gen.cuh), LayerNorm / dropout / embedding
row kernels, and an all-item logits + cross-entropy evaluated on the masked rows only (the reference materialises
(B, L, V+100) logits; rows whose label is 0 contribute neither loss nor gradient, bert4rec/trainer.py:45).

`train_step()` runs forward, loss assembly (bert4rec/trainer.py:112-134), backward, clip_grad_norm_ and Adam with its
coupled weight decay on the device; it is HIP-graph capturable (masked-row counts live in device memory).
"""
import os

import numpy as np
import torch

from .. import _lib, custom_ops, ops
from ..wide import Act, FlatModule, Tape, give

LN_EPS = 1e-5
MASK_FILL = -1e9
SITE_EMB_SEQ, SITE_EMB_DEC = 1, 2
_MHA = ("query_transfer", "key_transfer", "value_transfer", "out_transfer")


def enc_sites(i):
    b = 16 + 8 * i
    return {"attn": b, "after_multi": b + 1, "final": b + 2}


def dec_sites(i):
    b = 128 + 8 * i
    return {"attn": b, "after_multi": b + 1, "src_attn": b + 2, "after_src": b + 3, "final": b + 4}


# path components in the order the reference's constructors register them (bert4rec/model/bert.py:20-55, modules.py)
REF_ORDER = ["mask_bias", "item_emb", "encoder", "decoder", "mask_trans_feat", "mask_layer_norm", "encoder_layers", "decoder_layers",
             "word_emb", "pos_emb", "sent_emb", "multi_head_attention", "dec_multi_head_attention",
             "drop_residual_normalize_layer_after_multi", "src_dec_attention", "drop_residual_normalize_layer_after_src_dec", "ffn",
             "drop_residual_normalize_layer_final", "head_classifier", "layer_norm", "query_transfer", "key_transfer", "value_transfer",
             "out_transfer", "fc1", "fc2", "weight", "bias"]


def param_table(item_num, maxlen, d, H, nl, inner, type_vocab, vocab=None, block=0):
    """[(state_dict name, shape)] in flat order.  Within an attention block the three projection weights, then the three
    biases, are consecutive so that q/k/v (or k/v) run as one GEMM.  block > 0: the supernet's layout -- `block` candidate layers
    per depth, named encoder.encoder_layers.<depth>.<candidate>.* (bert4rec/model/modules.py:227-239, :366-378)."""
    V = item_num + 100 if vocab is None else vocab
    hd = d // H
    t = [("mask_bias", (V,)), ("item_emb.word_emb.weight", (V, d)), ("item_emb.pos_emb.weight", (maxlen, d)),
         ("item_emb.sent_emb.weight", (type_vocab, d)), ("item_emb.layer_norm.weight", (d,)), ("item_emb.layer_norm.bias", (d,))]

    def mha(p):
        return [(p + "." + n + ".weight", (d, d)) for n in _MHA[:3]] + [(p + "." + n + ".bias", (d,)) for n in _MHA[:3]] + \
               [(p + ".out_transfer.weight", (d, d)), (p + ".out_transfer.bias", (d,))]

    def ln(p):
        return [(p + ".layer_norm.weight", (d,)), (p + ".layer_norm.bias", (d,))]

    def ffn(p):
        return [(p + ".fc1.weight", (inner, d)), (p + ".fc1.bias", (inner,)), (p + ".fc2.weight", (d, inner)), (p + ".fc2.bias", (d,))]

    prefixes = lambda kind, i: ["%s.%s_layers.%d" % (kind, kind, i)] if block == 0 else ["%s.%s_layers.%d.%d" % (kind, kind, i, c) for c in range(block)]
    for i in range(nl):
        for p in prefixes("encoder", i):
            t += mha(p + ".multi_head_attention") + ln(p + ".drop_residual_normalize_layer_after_multi") + ffn(p + ".ffn")
            t += ln(p + ".drop_residual_normalize_layer_final") + [(p + ".head_classifier.weight", (H, hd)), (p + ".head_classifier.bias", (H,))]
    for i in range(nl):
        for p in prefixes("decoder", i):
            t += mha(p + ".dec_multi_head_attention") + ln(p + ".drop_residual_normalize_layer_after_multi")
            t += mha(p + ".src_dec_attention") + ln(p + ".drop_residual_normalize_layer_after_src_dec") + ffn(p + ".ffn")
            t += ln(p + ".drop_residual_normalize_layer_final")
    t += [("mask_trans_feat.weight", (d, d)), ("mask_trans_feat.bias", (d,)), ("mask_layer_norm.weight", (d,)), ("mask_layer_norm.bias", (d,))]
    return t


class BertModel(FlatModule):
    def __init__(self, usernum, itemnum, args, vocab=None, inner_units=None, block=0):
        super().__init__()
        self.usernum, self.itemnum = usernum, itemnum
        self.maxlen, self.num_heads, self.num_layers = args.maxlen, args.num_heads, args.num_layers
        self.hidden_units, self.inner_units = args.hidden_units, (args.inner_units if inner_units is None else inner_units)
        self.dropout, self.attention_dropout = _lib.dropout_rate(args.dropout, "dropout"), _lib.dropout_rate(args.attention_dropout, "attention_dropout")
        self.vocab = itemnum + 100 if vocab is None else vocab          # bert4rec/model/bert.py:20 (the supernet: itemnum + 2)
        self.ldv = (self.vocab + 3) // 4 * 4
        self.use_lce = os.environ.get("ADT_LCE", "1") != "0"            # fused all-item logits + CE where the shape allows (bf16, d 128 / 256)
        self.args = args
        self.prec = {"f32": ops.PREC_F32, "fp32": ops.PREC_F32, "bf16": ops.PREC_BF16}[getattr(args, "precision", "bf16")]
        if self.hidden_units % 64 or (self.hidden_units // self.num_heads) not in (16, 32, 64):
            raise _lib.AdtError("BertModel (adt_amd): hidden_units must be a multiple of 64 with head size 16/32/64, got d=%d H=%d"
                                % (self.hidden_units, self.num_heads))
        self._build_flat(param_table(itemnum, args.maxlen, args.hidden_units, args.num_heads, args.num_layers, self.inner_units,
                                     getattr(args, "type_vocab_size", 2), self.vocab, block), args.device, REF_ORDER)
        # bert4rec/trainer.py:29-37 re-initialises every Linear/Embedding weight N(0.01, initializer_range), LayerNorm 1/0,
        # Linear bias 0; do the same here so that a freshly constructed model is usable
        g = torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (1 << 31))
        std = float(getattr(args, "initializer_range", 0.02))
        for name, shape in self.table:
            v = self.P(name)
            if "layer_norm.weight" in name:
                v.fill_(1.0)
            elif name.endswith(".weight"):
                v.copy_(0.01 + std * torch.randn(shape, generator=g))

    # ------------------------------------------------------------------------------------------------------------------
    def _embed(self, tp, ids, site):
        """BertEmbedding.forward (bert4rec/model/modules.py:41-48)."""
        P, G = self.P, self.G
        L = self.maxlen
        x0 = Act(ops.embed_sum_fwd(ids, P("item_emb.word_emb.weight"), P("item_emb.pos_emb.weight"), L, P("item_emb.sent_emb.weight")[0]))

        def bw():   # runs after the LayerNorm / dropout closures registered below (reverse order)
            if x0.g is None:
                return
            T, d = x0.g.shape
            # nn.Embedding(padding_idx=0) for all three tables (modules.py:19-32): word row 0 and position 0 receive no
            # gradient from here (adt_item_scatter skips id 0; position row 0 is cleared at the end of the backward); the
            # sentence table is only ever indexed at 0, so its gradient is identically zero
            _lib.check(self.lib.adt_item_scatter(ops._p(ids), ops._p(x0.g), d, None, T, d, 1.0, 0.0, None, 0, 0,
                                                 ops._p(G("item_emb.word_emb.weight")), 1, 0, ops._stream()), "item_scatter")
            _lib.check(self.lib.adt_posemb_bwd(ops._p(self._ones(T)), ops._p(x0.g), T, L, d, 0.0, None, 0, 0,
                                               ops._p(G("item_emb.pos_emb.weight")), ops._stream()), "posemb_bwd")
        tp.bw.append(bw)
        z = tp.layernorm(x0, P("item_emb.layer_norm.weight"), P("item_emb.layer_norm.bias"), G("item_emb.layer_norm.weight"),
                         G("item_emb.layer_norm.bias"), LN_EPS)
        return tp.dropact(z, self.dropout, site)

    def _ones(self, T):
        if getattr(self, "_ones_buf", None) is None or self._ones_buf.numel() < T:
            self._ones_buf = torch.ones(T, device=self.dev, dtype=torch.int32)
        return self._ones_buf

    @staticmethod
    def _dqkv(holders, d):
        """Gradient buffers of the projection outputs, laid out like the packed forward tensors."""
        for h in holders:
            h.g = torch.empty_like(h.t)
        if len(holders) == 1:
            g = holders[0].g
            return g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
        return holders[0].g, holders[1].g[:, :d], holders[1].g[:, d:]

    def _ffn_drn(self, tp, p, x, ln_prefix, site):
        """FFN (modules.py:135-139, GELU) followed by its DropResidualNormalizeLayer."""
        P, G = self.P, self.G
        h = tp.dense(x, P(p + ".fc1.weight"), P(p + ".fc1.bias"), G(p + ".fc1.weight"), G(p + ".fc1.bias"), act=ops.ACT_GELU)
        z = tp.dense(h, P(p + ".fc2.weight"), P(p + ".fc2.bias"), G(p + ".fc2.weight"), G(p + ".fc2.bias"), p=self.attention_dropout, site=site, R=x)
        return tp.layernorm(z, P(ln_prefix + ".layer_norm.weight"), P(ln_prefix + ".layer_norm.bias"), G(ln_prefix + ".layer_norm.weight"),
                            G(ln_prefix + ".layer_norm.bias"), LN_EPS)

    def _attn_drn(self, tp, p, ln_prefix, xq, xkv, key_ids, B, site_attn, site_drop, want_o=False):
        """Attention sublayer + DropResidualNormalizeLayer: LN(dropout(out_transfer(attn)) + xq)."""
        P, G = self.P, self.G
        d, H, L = self.hidden_units, self.num_heads, self.maxlen
        pa = tp.p_eff(self.attention_dropout)
        # projections
        qw, vb = p + ".query_transfer.weight", p + ".value_transfer.bias"
        if xq is xkv:
            qkv = tp.dense(xq, self.span(qw, p + ".value_transfer.weight", (3 * d, d)), self.span(p + ".query_transfer.bias", vb, (3 * d,)),
                           self.span(qw, p + ".value_transfer.weight", (3 * d, d), grad=True), self.span(p + ".query_transfer.bias", vb, (3 * d,), grad=True))
            q, k, v = qkv.t[:, :d], qkv.t[:, d:2 * d], qkv.t[:, 2 * d:]
            holders = (qkv,)
        else:
            qa = tp.dense(xq, P(qw), P(p + ".query_transfer.bias"), G(qw), G(p + ".query_transfer.bias"))
            kw = p + ".key_transfer.weight"
            kva = tp.dense(xkv, self.span(kw, p + ".value_transfer.weight", (2 * d, d)), self.span(p + ".key_transfer.bias", vb, (2 * d,)),
                           self.span(kw, p + ".value_transfer.weight", (2 * d, d), grad=True), self.span(p + ".key_transfer.bias", vb, (2 * d,), grad=True))
            q, k, v = qa.t, kva.t[:, :d], kva.t[:, d:]
            holders = (qa, kva)
        O, LSE = ops.attn_masked_fwd(self.prec, q, k, v, B, H, L, False, key_ids, MASK_FILL, pa, self._seed, site_attn, tp.b_offset)
        o = Act(O)

        def bw():
            if o.g is None:
                return
            ops.attn_masked_bwd(self.prec, q, k, v, O, LSE, o.g, B, H, L, False, key_ids, MASK_FILL, pa, self._seed, site_attn, tp.b_offset,
                                out=self._dqkv(holders, d))
        tp.bw.append(bw)
        ow = p + ".out_transfer.weight"
        z = tp.dense(o, P(ow), P(p + ".out_transfer.bias"), G(ow), G(p + ".out_transfer.bias"), p=self.attention_dropout, site=site_drop, R=xq)
        y = tp.layernorm(z, P(ln_prefix + ".layer_norm.weight"), P(ln_prefix + ".layer_norm.bias"), G(ln_prefix + ".layer_norm.weight"),
                         G(ln_prefix + ".layer_norm.bias"), LN_EPS)
        return y, o

    def _encode(self, tp, src, B):
        """log2feats + Encoder.forward (bert.py:60-67, modules.py:208-216)."""
        P, G = self.P, self.G
        x = self._embed(tp, src, SITE_EMB_SEQ)
        enc_inputs, recs = [], []
        for i in range(self.num_layers):
            enc_inputs.append(x)
            x, rec = self._enc_layer(tp, "encoder.encoder_layers.%d" % i, x, src, B, enc_sites(i))
            recs.append(rec)
        return x, enc_inputs, recs

    def _enc_layer(self, tp, p, x, src, B, st):
        """EncoderLayer.forward (modules.py:165-182) -> (output, head-classifier log-probabilities)."""
        P, G = self.P, self.G
        h, o = self._attn_drn(tp, p + ".multi_head_attention", p + ".drop_residual_normalize_layer_after_multi", x, x, src, B, st["attn"],
                              st["after_multi"])
        rec = tp.headcls(o, P(p + ".head_classifier.weight"), P(p + ".head_classifier.bias"), G(p + ".head_classifier.weight"),
                         G(p + ".head_classifier.bias"))
        return self._ffn_drn(tp, p + ".ffn", h, p + ".drop_residual_normalize_layer_final", st["final"]), rec

    def _dec_layer(self, tp, p, x, dec, src, enc, B, st):
        """DecoderLayer.forward (modules.py:297-325)."""
        g, _ = self._attn_drn(tp, p + ".dec_multi_head_attention", p + ".drop_residual_normalize_layer_after_multi", x, x, dec, B, st["attn"],
                              st["after_multi"])
        g2, _ = self._attn_drn(tp, p + ".src_dec_attention", p + ".drop_residual_normalize_layer_after_src_dec", g, enc, src, B, st["src_attn"],
                               st["after_src"])
        return self._ffn_drn(tp, p + ".ffn", g2, p + ".drop_residual_normalize_layer_final", st["final"])

    def _decode(self, tp, dec, src, enc, B):
        """decode + Decoder.forward (bert.py:69-78, modules.py:297-325, 352-358); outputs in layer order (not yet reversed)."""
        x = self._embed(tp, dec, SITE_EMB_DEC)
        tp.mark_decoder_start()
        outs = []
        for i in range(self.num_layers):
            x = self._dec_layer(tp, "decoder.decoder_layers.%d" % i, x, dec, src, enc, B, dec_sites(i))
            outs.append(x)
        return outs

    def _head(self, tp, x):
        """downstream up to the LayerNorm (bert.py:80-85): LN(GELU(mask_trans_feat(x)))."""
        P, G = self.P, self.G
        h = tp.dense(x, P("mask_trans_feat.weight"), P("mask_trans_feat.bias"), G("mask_trans_feat.weight"), G("mask_trans_feat.bias"), act=ops.ACT_GELU)
        return tp.layernorm(h, P("mask_layer_norm.weight"), P("mask_layer_norm.bias"), G("mask_layer_norm.weight"), G("mask_layer_norm.bias"), LN_EPS)

    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, src_ids, dec_ids, seq_pos_ids=None, seq_sent_ids=None, deq_pos_ids=None, deq_sent_ids=None):
        """bert.py:92-108.  Position ids are always 0..L-1 and sentence ids 0 in the reference's callers (trainer.py:104-107);
        they are accepted for signature compatibility.  Returns (logits (B, L, V+100), enc_inputs, dec_outputs reversed,
        ind_outputs).  Under autograd the tensors are wired into it (adt_amd::model_forward), so the reference's loop body
        (bert4rec/trainer.py:100-138: CrossEntropyLoss over all items, loss.backward(), clip, Adam) runs on them unchanged -- with the
        reference's (B, L, V+100) logits materialised; FusedBertTrainer.step (masked rows only) is the fast way to train."""
        ids = [self.ids(src_ids), self.ids(dec_ids)]
        if custom_ops.wants_grad(self):
            outs = custom_ops.forward_with_grad(self, ids)
        else:
            with torch.no_grad():
                outs, _ = self._op_forward(ids, self.training)
        nl = self.num_layers
        return outs[0], list(outs[1:1 + nl]), list(outs[1 + nl:1 + 2 * nl]), list(outs[1 + 2 * nl:1 + 3 * nl])

    def _op_forward(self, ids, training):
        src, dec = ids
        B, L = src.shape
        d, H = self.hidden_units, self.num_heads
        if training:
            self.next_seed()
        tp = Tape(self, self.prec, training)
        enc, enc_inputs, recs = self._encode(tp, src.view(-1), B)
        dec_outs = self._decode(tp, dec.view(-1), src.view(-1), enc, B)
        h = self._head(tp, enc)
        logits = tp.dense(h, self.P("item_emb.word_emb.weight"), self.P("mask_bias"), self.G("item_emb.word_emb.weight"), self.G("mask_bias"))
        dec_outs.reverse()
        outs = [logits.t.view(B, L, self.vocab)] + [a.t.view(B, L, d) for a in enc_inputs] + [a.t.view(B, L, d) for a in dec_outs] + \
               [r.t.view(B, L, H, H) for r in recs]
        return outs, {"tp": tp, "acts": [logits] + list(enc_inputs) + list(dec_outs) + list(recs)}

    def _op_backward(self, st, grads):
        self.flat_grad.zero_()
        for a, g in zip(st["acts"], grads):
            give(a, custom_ops.take_grad(g, tuple(a.t.shape)))
        st["tp"].backward()
        self.G("item_emb.pos_emb.weight")[0].zero_()   # padding_idx = 0 of the position table (modules.py:24-28)
        return custom_ops.param_grads(self)

    @torch.no_grad()
    def predict(self, user_ids, seqs, seq_pos_ids=None, seq_sent_ids=None, candidates=None, want_rank=False):
        """bert.py:110-116: candidate scores at the last position (the appended [MASK] token)."""
        src = self.ids(seqs)
        cand = self.ids(candidates)
        B, L = src.shape
        tp = Tape(self, self.prec, False)
        enc, _, _ = self._encode(tp, src.view(-1), B)
        h = self._head(tp, Act(ops.gather_rows(enc.t, torch.arange(L - 1, B * L, L, device=self.dev, dtype=torch.int32))))
        logits, rank = ops.score_rank_bias(h.t, self.hidden_units, self.P("item_emb.word_emb.weight"), self.P("mask_bias"), cand, B, cand.shape[1],
                                           want_rank)
        return (logits, rank) if want_rank else logits

    # ------------------------------------------------------------------------------------------------------------------
    def stage(self, src, dec, labels, n_valid_global=None):
        """Host -> device staging of one batch (the reference moves LongTensors per call, trainer.py:100-110): int32 ids,
        the indices and labels of the masked rows (label != 0) and their count, and the loss normalisers
        {1 / n_valid (global), B*L*d, B*L*H}."""
        src, dec, labels = (np.ascontiguousarray(np.asarray(a), dtype=np.int32) for a in (src, dec, labels))
        B, L = src.shape
        flat = labels.reshape(-1)
        rows = np.nonzero(flat)[0].astype(np.int32)
        M = int(rows.size)
        cap = B * L
        rows_p = np.zeros(cap, np.int32)
        rows_p[:M] = rows
        lab_p = np.zeros(cap, np.int32)
        lab_p[:M] = flat[rows]
        nv = float(max(M if n_valid_global is None else n_valid_global, 1))      # a batch with no masked position: CE term 0, never 1/0
        return {"B": B, "src": self.ids(src), "dec": self.ids(dec), "rows": self.ids(rows_p), "labels": self.ids(lab_p),
                "M": torch.tensor([M], device=self.dev, dtype=torch.int32), "M_host": M,
                "inv_count": torch.tensor([1.0 / nv], device=self.dev, dtype=torch.float32)}

    def loss_forward_backward(self, st, lambda1, lambda2, norms, loss_slots, b_offset=0, mcap=None):
        """Forward, loss assembly (bert4rec/trainer.py:112-134) and backward into flat_grad (accumulated).  norms: device
        {_, n_mse, n_nll}; loss_slots: (1 + 2*num_layers) x 64 floats {ce, mse_i.., nll_l..}.  mcap bounds the rows the
        masked-row GEMMs are launched for (default: all B*L; the live count is st["M"] in device memory)."""
        B, L, d, H, nl = st["B"], self.maxlen, self.hidden_units, self.num_heads, self.num_layers
        T = B * L
        tp = Tape(self, self.prec, self.training, row_offset=b_offset * L, b_offset=b_offset)
        src, dec = st["src"].view(-1), st["dec"].view(-1)
        enc, enc_inputs, recs = self._encode(tp, src, B)
        dec_outs = self._decode(tp, dec, src, enc, B)
        h = self._head(tp, enc)
        # all-item logits + CE on the masked rows only
        mcap = T if mcap is None else min(T, mcap)
        Mdev = st["M"]
        E, gE = self.P("item_emb.word_emb.weight"), self.G("item_emb.word_emb.weight")
        h.g = torch.zeros_like(h.t)
        if self.use_lce and ops.lce_supported(self.prec, d):
            # fused: online log-sum-exp forward, tile-recomputing backward; no (rows, V) logits in memory (adt_amd/csrc/adt_lce.cuh)
            ops.lce_fwd_bwd(h.t, st["rows"], st["labels"], mcap, Mdev, E, self.P("mask_bias"), st["inv_count"], loss_slots[0], h.g, gE,
                            self.G("mask_bias"))
        else:
            hm = ops.gather_rows(h.t, st["rows"], mcap, Mdev)
            logits, _ = ops.dense_fwd(self.prec, hm, E, self.P("mask_bias"), t_dev=Mdev, ldy=self.ldv)
            ops.ce_rows(logits, st["labels"], self.vocab, st["inv_count"], loss_slots[0], mcap, Mdev)
            dhm = torch.empty_like(hm)
            ops.dense_bwd(self.prec, logits, hm, E, gE, self.G("mask_bias"), dhm, False, t_dev=Mdev)
            ops.scatter_rows(dhm, st["rows"], h.g, False, mcap, Mdev)
        # reconstruction (MSE) and independence (NLL) seeds
        for i in range(nl):
            if lambda1[i] != 0:
                a, bq = enc_inputs[i], dec_outs[nl - 1 - i]        # decoder outputs are reversed (modules.py:357)
                if a.g is None:
                    a.g = torch.zeros_like(a.t)
                bq.g = torch.empty_like(bq.t)
                ops.mse_seed(a.t, bq.t, lambda1[i], norms, a.g, True, bq.g, loss_slots[1 + i])
        if H > 1:
            for l in range(nl):
                if lambda2[l] != 0:
                    recs[l].g = torch.empty_like(recs[l].t)
                    ops.nll_seed(recs[l].t, H, lambda2[l], norms, recs[l].g, loss_slots[1 + nl + l])
        tp.backward()
        self.G("item_emb.pos_emb.weight")[0].zero_()   # padding_idx = 0 of the position table (modules.py:24-28)
