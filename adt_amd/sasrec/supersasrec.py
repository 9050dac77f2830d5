"""SuperSASRecModel on the MI355X hot path -- drop-in for the reference's sasrec/supersasrec.py (+ super_modules.py,
base_super_modules.py): the weight-sharing supernet the evolutionary lambda search trains and evaluates.

Per depth there are rec_size * ind_size candidate encoder layers and as many decoder layers; `set_choice(block_cand)`
selects four of them and their bilinear weights, and a forward pass runs those four on the same input and mixes them
(classifier log-probabilities are mixed, then log_softmax'd again: super_modules.py:42-49).  Every candidate layer runs on
the same stage kernels as the plain model (dense layers on MFMA, causal attention with LDS-resident K/V, LayerNorm,
head classifier); the mixing and its reverse are one axpy kernel per candidate.

`SuperTrainer.step()` is the loop body of SearcherEvolution._train_warmup (sasrec/evolution.py:286-316): BCE +
rec_weights[i] * MSE + ind_weights[stale i] * NLL, backward, clip_grad_norm_, Adam with coupled weight decay -- with
torch's semantics for parameters whose grad is None (candidates that were not mixed in keep their moments, their own
step count and are not decayed).
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib, custom_ops, ops
from ..supersearch import candidate_features, cand_to_block, get_position, get_shared, get_weight  # noqa: F401
from ..wide import Act, FlatModule, Tape, give
from .model import REF_ORDER

LN_EPS = 1e-8
SITE_EMB_SEQ, SITE_EMB_DEC = 1, 2
CAND_SITE = 4096

_ENC = ["attention_layernorm.weight", "attention_layernorm.bias", "attention_layer.in_proj_weight", "attention_layer.in_proj_bias",
        "attention_layer.out_proj.weight", "attention_layer.out_proj.bias", "forward_layernorm.weight", "forward_layernorm.bias",
        "forward_layer.conv1.weight", "forward_layer.conv1.bias", "forward_layer.conv2.weight", "forward_layer.conv2.bias", "sparse.weight", "sparse.bias"]
_DEC = ["layer_norm.weight", "layer_norm.bias", "slf_attn.in_proj_weight", "slf_attn.in_proj_bias", "slf_attn.out_proj.weight", "slf_attn.out_proj.bias",
        "enc_attn.in_proj_weight", "enc_attn.in_proj_bias", "enc_attn.out_proj.weight", "enc_attn.out_proj.bias", "pos_ffn.conv1.weight",
        "pos_ffn.conv1.bias", "pos_ffn.conv2.weight", "pos_ffn.conv2.bias", "pos_ffn_layernorm.weight", "pos_ffn_layernorm.bias"]


def _shapes(d, H):
    hd = d // H
    enc = [(d,), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (d,), (d,), (d, d, 1), (d,), (d, d, 1), (d,), (H, hd), (H,)]
    dec = [(d,), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (d, d, 1), (d,), (d, d, 1), (d,), (d,), (d,)]
    return enc, dec


def enc_sites(i, k):
    b = 16 + 8 * i + CAND_SITE * (k + 1)
    return {"attn": b, "ffn1": b + 1, "ffn2": b + 2}


def dec_sites(i, k):
    b = 128 + 8 * i + CAND_SITE * (k + 1)
    return {"slf": b, "enc": b + 1, "ffn1": b + 2, "ffn2": b + 3}


class SuperSASRecModel(FlatModule):
    def __init__(self, usernum, itemnum, rec_choice, ind_choice, args):
        super().__init__()
        self.usernum, self.itemnum = usernum, itemnum
        self.num_heads, self.maxlen, self.num_layers = args.num_heads, args.maxlen, args.num_layers
        self.hidden_units, self.dropout = args.hidden_units, _lib.dropout_rate(args.dropout, "dropout")
        self.rec_choice, self.ind_choice = np.asarray(rec_choice, np.float64), np.asarray(ind_choice, np.float64)
        self.block = len(self.rec_choice) * len(self.ind_choice)
        self.prec = {"f32": ops.PREC_F32, "fp32": ops.PREC_F32, "bf16": ops.PREC_BF16}[getattr(args, "precision", "bf16")]
        d, H = self.hidden_units, self.num_heads
        if d != 64 or (d // H) not in (16, 32, 64):
            raise _lib.AdtError("SuperSASRecModel (adt_amd): built for hidden_units=64 with head size 16/32/64, got d=%d H=%d" % (d, H))
        es, ds = _shapes(d, H)
        table = [("item_emb.weight", (itemnum + 1, d)), ("pos_emb.weight", (args.maxlen, d))]
        for i in range(self.num_layers):
            for c in range(self.block):
                table += [("encoder.encoder_layers.%d.%d.%s" % (i, c, n), s) for n, s in zip(_ENC, es)]
        for i in range(self.num_layers):
            for c in range(self.block):
                table += [("decoder.decoder_layers.%d.%d.%s" % (i, c, n), s) for n, s in zip(_DEC, ds)]
        self._build_flat(table, args.device, REF_ORDER)
        g = torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (1 << 31))
        for name, shape in self.table:     # evolution.py:103-107: xavier_normal_ on >= 2-D tensors; 1-D keep torch defaults
            v = self.P(name)
            if len(shape) >= 2:
                fan = shape[0] + shape[1] * (shape[2] if len(shape) > 2 else 1)
                v.copy_(torch.randn(shape, generator=g) * math.sqrt(2.0 / fan))
            elif "norm.weight" in name:
                v.fill_(1.0)
        self.shared = [((0, 0, 0, 0), (0.0, 0.0, 0.0, 0.0)) for _ in range(self.num_layers)]

    def set_choice(self, cand):
        """supersasrec.py:113-115 / base_super_modules.py:42-57."""
        self.shared = get_shared(self.rec_choice, self.ind_choice, np.asarray(cand, np.float64))

    def layer_range(self, kind, depth, cand):
        """Flat [lo, hi) of the trainable tensors of one candidate layer (pos_ffn_layernorm, never used, is excluded;
        so is the head classifier when there is a single head)."""
        names = _ENC if kind == "encoder" else _DEC
        last = names[-3] if kind == "decoder" else (names[-1] if self.num_heads > 1 else names[-3])
        p = "%s.%s_layers.%d.%d." % (kind, kind, depth, cand)
        lo = self._views[p + names[0]][0]
        o, n, _ = self._views[p + last]
        return lo, o + (n + 3) // 4 * 4

    # ---- grouped candidate execution on the per-sequence fused layer kernels -----------------------------------------------------
    # (adt_seq_enc_layer_fwd / bwd, adt_seq_dec_layer_fwd / bwd: one launch per candidate layer forward, two / four per backward,
    # instead of ~10 / ~25 stage launches; the bilinear mixing weight is the output epilogue y_mix (+)= w_k * layer_k(x) and the
    # gradient prologue dy_k = w_k * dy_mix.)  bf16 operands, L % 4 == 0, L <= 224; ADT_SUPER_FUSED=0 keeps the stage kernels.
    _ENC_FIELDS = dict(zip(_lib.EncLayerPtrs.NAMES, _ENC))
    _DEC_FIELDS = dict(zip(_lib.DecLayerPtrs.NAMES, _DEC[:14]))

    def fused_layers(self):
        if getattr(self, "_fused", None) is None:
            import os
            on = os.environ.get("ADT_SUPER_FUSED", "1") != "0" and self.prec == ops.PREC_BF16
            self._fused = bool(on and self.lib.adt_seq_layer_supported(self.prec, self.maxlen, self.hidden_units, self.hidden_units // self.num_heads))
            if self._fused:
                self._wimg = torch.empty(3 * self.flat.numel(), device=self.dev, dtype=torch.float32)      # 6 bf16 per parameter float
                self._blocks = {}
        return self._fused

    def _ptrs(self, kind, prefix, grad):
        cls, fields = (_lib.EncLayerPtrs, self._ENC_FIELDS) if kind == "enc" else (_lib.DecLayerPtrs, self._DEC_FIELDS)
        key = (prefix, grad)
        cache = self.__dict__.setdefault("_ptr_cache", {})
        if key not in cache:
            src = self.G if grad else self.P
            cache[key] = cls(**{f: src(prefix + "." + n).data_ptr() for f, n in fields.items()})
        return cache[key]

    def pack_selected(self, layers=None):
        """bf16 LDS images of every 64 x 64 weight of the selected candidate layers (once per step: the weights change); `layers`: an
        explicit set of (depth, candidate) pairs instead of the current block choice."""
        offs = []
        d = self.hidden_units
        if layers is None:
            layers = sorted({(depth, idx) for depth, (idxs, _) in enumerate(self.shared) for idx in idxs})
        for depth, idx in layers:
            if True:
                e, dd = "encoder.encoder_layers.%d.%d." % (depth, idx), "decoder.decoder_layers.%d.%d." % (depth, idx)
                base = self._views[e + "attention_layer.in_proj_weight"][0]
                offs += [base, base + d * d, base + 2 * d * d] + [self._views[e + n][0] for n in ("attention_layer.out_proj.weight",
                                                                                                  "forward_layer.conv1.weight", "forward_layer.conv2.weight")]
                for n in ("slf_attn.in_proj_weight", "enc_attn.in_proj_weight"):
                    base = self._views[dd + n][0]
                    offs += [base, base + d * d, base + 2 * d * d]
                offs += [self._views[dd + n][0] for n in ("slf_attn.out_proj.weight", "enc_attn.out_proj.weight", "pos_ffn.conv1.weight",
                                                          "pos_ffn.conv2.weight")]
        for s0 in range(0, len(offs), 256):
            chunk = offs[s0:s0 + 256]
            arr = (ctypes.c_int * len(chunk))(*chunk)
            _lib.check(self.lib.adt_pack_wimg(ops._p(self.flat), ops._p(self._wimg), arr, len(chunk), ops._stream()), "pack_wimg")

    def _scratch(self, T):
        n = 4 * ((T * 64 + 63) // 64 * 64)
        if getattr(self, "_scr", None) is None or self._scr.numel() < n:
            self._scr = torch.empty(n, device=self.dev, dtype=torch.float32)
        return self._scr

    def _enc_layers_fused(self, tp, depth, x, ids, B):
        """The four selected encoder layers of one depth: -> (mixed output Act, [head-classifier Acts], weights)."""
        L, H = self.maxlen, self.num_heads
        idxs, ws = self.shared[depth]
        T = B * L
        p = tp.p_eff(self.dropout)
        y = Act(torch.empty(T, self.hidden_units, device=self.dev, dtype=torch.float32))
        nsave = self.lib.adt_seq_layer_save_floats(B, L, H, 0)
        recs, saves = [], []
        for k, (idx, wk) in enumerate(zip(idxs, ws)):
            pre = "encoder.encoder_layers.%d.%d" % (depth, idx)
            st = enc_sites(depth, k)
            save = torch.empty(nsave, device=self.dev, dtype=torch.float32)
            rec = Act(torch.empty(T, H * H, device=self.dev, dtype=torch.float32)) if H > 1 else None
            _lib.check(self.lib.adt_seq_enc_layer_fwd(B, L, H, ops._p(ids), ops._p(x.t), ctypes.byref(self._ptrs("enc", pre, False)), ops._p(self.flat),
                                                      ops._p(self._wimg), float(p), ops._p(self._seed), st["attn"], st["ffn1"], st["ffn2"], tp.b_offset,
                                                      int(tp.training), ops._p(save), ops._p(y.t), float(wk), int(k > 0),
                                                      ops._p(rec.t) if rec is not None else None, ops._stream()), "seq_enc_layer_fwd")
            recs.append(rec)
            saves.append(save)

        def bw():
            if y.g is None and all(r is None or r.g is None for r in recs):
                return
            gy = y.g if y.g is not None else torch.zeros_like(y.t)
            for k, (idx, wk) in enumerate(zip(idxs, ws)):
                pre = "encoder.encoder_layers.%d.%d" % (depth, idx)
                st = enc_sites(depth, k)
                acc = x.g is not None
                if not acc:
                    x.g = torch.empty_like(x.t)
                rec = recs[k]
                drec = rec.g if (rec is not None and rec.g is not None) else None
                _lib.check(self.lib.adt_seq_enc_layer_bwd(B, L, H, ops._p(ids), ops._p(x.t), ctypes.byref(self._ptrs("enc", pre, False)),
                                                          ctypes.byref(self._ptrs("enc", pre, True)), ops._p(self.flat), ops._p(self._wimg), float(p),
                                                          ops._p(self._seed), st["attn"], st["ffn1"], st["ffn2"], tp.b_offset, ops._p(saves[k]),
                                                          ops._p(gy), float(wk), ops._p(rec.t) if rec is not None else None,
                                                          ops._p(drec) if drec is not None else None, ops._p(x.g), int(acc), ops._p(self._scratch(T)),
                                                          ops._stream()), "seq_enc_layer_bwd")
        tp.bw.append(bw)
        return y, recs, ws

    def _dec_layers_fused(self, tp, depth, x, feats, ids, B):
        L, H = self.maxlen, self.num_heads
        idxs, ws = self.shared[depth]
        T = B * L
        p = tp.p_eff(self.dropout)
        y = Act(torch.empty(T, self.hidden_units, device=self.dev, dtype=torch.float32))
        nsave = self.lib.adt_seq_layer_save_floats(B, L, H, 1)
        saves = []
        for k, (idx, wk) in enumerate(zip(idxs, ws)):
            pre = "decoder.decoder_layers.%d.%d" % (depth, idx)
            st = dec_sites(depth, k)
            save = torch.empty(nsave, device=self.dev, dtype=torch.float32)
            _lib.check(self.lib.adt_seq_dec_layer_fwd(B, L, H, ops._p(ids), ops._p(x.t), ops._p(feats.t), ctypes.byref(self._ptrs("dec", pre, False)),
                                                      ops._p(self.flat), ops._p(self._wimg), float(p), ops._p(self._seed), st["slf"], st["enc"], st["ffn1"],
                                                      st["ffn2"], tp.b_offset, ops._p(save), ops._p(y.t), float(wk), int(k > 0), ops._stream()),
                       "seq_dec_layer_fwd")
            saves.append(save)

        def bw():
            if y.g is None:
                return
            if feats.g is None:
                feats.g = torch.zeros_like(feats.t)
            for k, (idx, wk) in enumerate(zip(idxs, ws)):
                pre = "decoder.decoder_layers.%d.%d" % (depth, idx)
                st = dec_sites(depth, k)
                acc = x.g is not None
                if not acc:
                    x.g = torch.empty_like(x.t)
                _lib.check(self.lib.adt_seq_dec_layer_bwd(B, L, H, ops._p(ids), ops._p(x.t), ops._p(feats.t), ctypes.byref(self._ptrs("dec", pre, False)),
                                                          ctypes.byref(self._ptrs("dec", pre, True)), ops._p(self.flat), ops._p(self._wimg), float(p),
                                                          ops._p(self._seed), st["slf"], st["enc"], st["ffn1"], st["ffn2"], tp.b_offset, ops._p(saves[k]),
                                                          ops._p(y.g), float(wk), ops._p(x.g), int(acc), ops._p(feats.g), ops._p(self._scratch(T)),
                                                          ops._stream()), "seq_dec_layer_bwd")
        tp.bw.append(bw)
        return y

    # ------------------------------------------------------------------------------------------------------------------
    def _embed(self, tp, ids, site):
        P, G = self.P, self.G
        L = self.maxlen
        p = tp.p_eff(self.dropout)
        x = Act(ops.embed_fwd(ids, P("item_emb.weight"), P("pos_emb.weight"), L, p, self._seed, site, tp.row_offset))

        def bw():
            if x.g is not None:
                ops.embed_bwd(ids, x.g, L, p, self._seed, site, G("item_emb.weight"), G("pos_emb.weight"), tp.row_offset)
        tp.bw.append(bw)
        return x

    def _attn(self, tp, q, kv, B, site, qkv=None):
        """Causal attention on packed projections; q: (T, d) Act, kv: (T, 2d) Act (or qkv: one (T, 3d) Act)."""
        d, H, L = self.hidden_units, self.num_heads, self.maxlen
        p = tp.p_eff(self.dropout)
        if qkv is not None:
            Q, K, V = qkv.t[:, :d], qkv.t[:, d:2 * d], qkv.t[:, 2 * d:]
        else:
            Q, K, V = q.t, kv.t[:, :d], kv.t[:, d:]
        O, LSE = ops.attn_fwd(self.prec, Q, K, V, B, H, L, True, p, self._seed, site, tp.b_offset)
        o = Act(O)

        def bw():
            if o.g is None:
                return
            if qkv is not None:
                qkv.g = torch.empty_like(qkv.t)
                out = (qkv.g[:, :d], qkv.g[:, d:2 * d], qkv.g[:, 2 * d:])
            else:
                q.g, kv.g = torch.empty_like(q.t), torch.empty_like(kv.t)
                out = (q.g, kv.g[:, :d], kv.g[:, d:])
            ops.attn_bwd(self.prec, Q, K, V, O, LSE, o.g, B, H, L, True, p, self._seed, site, tp.b_offset, out=out)
        tp.bw.append(bw)
        return o

    def _conv(self, name, grad=False):
        d = self.hidden_units
        return (self.G(name) if grad else self.P(name)).view(d, d)

    def _enc_layer(self, tp, p, x, ids, B, st):
        """EncoderLayer.forward (sasrec/modules.py:644-655) -> (seqs, head-classifier log-probabilities)."""
        P, G = self.P, self.G
        d = self.hidden_units
        Q = tp.layernorm(x, P(p + ".attention_layernorm.weight"), P(p + ".attention_layernorm.bias"), G(p + ".attention_layernorm.weight"),
                         G(p + ".attention_layernorm.bias"), LN_EPS)
        W, b, gW, gb = P(p + ".attention_layer.in_proj_weight"), P(p + ".attention_layer.in_proj_bias"), G(p + ".attention_layer.in_proj_weight"), \
            G(p + ".attention_layer.in_proj_bias")
        q = tp.dense(Q, W[:d], b[:d], gW[:d], gb[:d])           # q from LN(x) ...
        kv = tp.dense(x, W[d:], b[d:], gW[d:], gb[d:])          # ... k, v from the un-normalised x (modules.py:646-647)
        o = self._attn(tp, q, kv, B, st["attn"])
        rec = tp.headcls(o, P(p + ".sparse.weight"), P(p + ".sparse.bias"), G(p + ".sparse.weight"), G(p + ".sparse.bias"))
        ow = p + ".attention_layer.out_proj"
        h = tp.dense(o, P(ow + ".weight"), P(ow + ".bias"), G(ow + ".weight"), G(ow + ".bias"), R=Q)
        h2 = tp.layernorm(h, P(p + ".forward_layernorm.weight"), P(p + ".forward_layernorm.bias"), G(p + ".forward_layernorm.weight"),
                          G(p + ".forward_layernorm.bias"), LN_EPS)
        c1, c2 = p + ".forward_layer.conv1", p + ".forward_layer.conv2"
        f1 = tp.dense(h2, self._conv(c1 + ".weight"), P(c1 + ".bias"), self._conv(c1 + ".weight", True), G(c1 + ".bias"), act=ops.ACT_RELU,
                      p=self.dropout, site=st["ffn1"])
        y = tp.dense(f1, self._conv(c2 + ".weight"), P(c2 + ".bias"), self._conv(c2 + ".weight", True), G(c2 + ".bias"), p=self.dropout,
                     site=st["ffn2"], R=h2, mask_ids=ids)
        return y, rec

    def _dec_layer(self, tp, p, x, enc, ids, B, st):
        """DecoderLayer.forward (sasrec/modules.py:666-677)."""
        P, G = self.P, self.G
        d = self.hidden_units
        D = tp.layernorm(x, P(p + ".layer_norm.weight"), P(p + ".layer_norm.bias"), G(p + ".layer_norm.weight"), G(p + ".layer_norm.bias"), LN_EPS)
        s, e = p + ".slf_attn", p + ".enc_attn"
        qkv = tp.dense(D, P(s + ".in_proj_weight"), P(s + ".in_proj_bias"), G(s + ".in_proj_weight"), G(s + ".in_proj_bias"))
        o1 = self._attn(tp, None, None, B, st["slf"], qkv=qkv)
        a1 = tp.dense(o1, P(s + ".out_proj.weight"), P(s + ".out_proj.bias"), G(s + ".out_proj.weight"), G(s + ".out_proj.bias"))
        W, b, gW, gb = P(e + ".in_proj_weight"), P(e + ".in_proj_bias"), G(e + ".in_proj_weight"), G(e + ".in_proj_bias")
        q2 = tp.dense(a1, W[:d], b[:d], gW[:d], gb[:d])
        kv2 = tp.dense(enc, W[d:], b[d:], gW[d:], gb[d:])
        o2 = self._attn(tp, q2, kv2, B, st["enc"])
        a2 = tp.dense(o2, P(e + ".out_proj.weight"), P(e + ".out_proj.bias"), G(e + ".out_proj.weight"), G(e + ".out_proj.bias"))
        c1, c2 = p + ".pos_ffn.conv1", p + ".pos_ffn.conv2"
        f1 = tp.dense(a2, self._conv(c1 + ".weight"), P(c1 + ".bias"), self._conv(c1 + ".weight", True), G(c1 + ".bias"), act=ops.ACT_RELU,
                      p=self.dropout, site=st["ffn1"])
        return tp.dense(f1, self._conv(c2 + ".weight"), P(c2 + ".bias"), self._conv(c2 + ".weight", True), G(c2 + ".bias"), p=self.dropout,
                        site=st["ffn2"], R=a2, R2=D, mask_ids=ids)

    def _encode(self, tp, seq, B):
        x = self._embed(tp, seq, SITE_EMB_SEQ)
        enc_in, recs = [], []
        fused = self.fused_layers()
        if fused:
            self.pack_selected()
        for i, (idxs, ws) in enumerate(self.shared):
            enc_in.append(x)
            if fused:
                x, rk, wk = self._enc_layers_fused(tp, i, x, seq, B)
                if self.num_heads > 1:
                    recs.append(tp.log_softmax(tp.mix([(r, float(w)) for r, w in zip(rk, wk)]), self.num_heads))
                else:
                    recs.append(Act(torch.zeros(x.t.shape[0], 1, device=self.dev, dtype=torch.float32)))     # log_softmax over one class
                continue
            outs, inds = [], []
            for k, (idx, w) in enumerate(zip(idxs, ws)):
                y, rec = self._enc_layer(tp, "encoder.encoder_layers.%d.%d" % (i, idx), x, seq, B, enc_sites(i, k))
                outs.append((y, float(w)))
                inds.append((rec, float(w)))
            x = tp.mix(outs)
            recs.append(tp.log_softmax(tp.mix(inds), self.num_heads))
        return x, enc_in, recs

    def _decode(self, tp, dec, feats, B):
        y = self._embed(tp, dec, SITE_EMB_DEC)
        outs = []
        fused = self.fused_layers()
        for i, (idxs, ws) in enumerate(self.shared):
            if fused:
                y = self._dec_layers_fused(tp, i, y, feats, dec, B)
                outs.append(y)
                continue
            parts = [(self._dec_layer(tp, "decoder.decoder_layers.%d.%d" % (i, idx), y, feats, dec, B, dec_sites(i, k)), float(w))
                     for k, (idx, w) in enumerate(zip(idxs, ws))]
            y = tp.mix(parts)
            outs.append(y)
        return outs

    def forward(self, user_ids, log_seqs, dec_seqs, pos_seqs, neg_seqs):
        """supersasrec.py:81-94 -> (pos_logits, neg_logits, encoder_layer_input, decoder_layer_output [reversed], rec_layer_ind).
        Under autograd the tensors are wired into it (adt_amd::model_forward), so the reference's warm-up loop body
        (sasrec/evolution.py:296-316) runs on them unchanged; SuperTrainer.step is the fused way."""
        ids = [self.ids(a) for a in (log_seqs, dec_seqs, pos_seqs, neg_seqs)]
        if custom_ops.wants_grad(self):
            outs = custom_ops.forward_with_grad(self, ids)
        else:
            with torch.no_grad():
                outs, _ = self._op_forward(ids, self.training)
        nl = self.num_layers
        return outs[0], outs[1], list(outs[2:2 + nl]), list(outs[2 + nl:2 + 2 * nl]), list(outs[2 + 2 * nl:2 + 3 * nl])

    def _op_forward(self, ids, training):
        seq, dec, pos, neg = ids
        B, L = seq.shape
        d, H = self.hidden_units, self.num_heads
        if training:
            self.next_seed()
        tp = Tape(self, self.prec, training)
        feats, enc_in, recs = self._encode(tp, seq.view(-1), B)
        dec_outs = self._decode(tp, dec.view(-1), feats, B)
        pl, nl = ops.logits_fwd(feats.t, self.P("item_emb.weight"), pos.view(-1), neg.view(-1))
        dec_outs.reverse()
        outs = [pl.view(B, L), nl.view(B, L)] + [a.t.view(B, L, d) for a in enc_in] + [a.t.view(B, L, d) for a in dec_outs] + \
               [r.t.view(B, L, H, H) for r in recs]
        return outs, {"tp": tp, "feats": feats, "acts": list(enc_in) + list(dec_outs) + list(recs), "pos": pos, "neg": neg}

    def _op_backward(self, st, grads):
        tp, feats, pos, neg = st["tp"], st["feats"], st["pos"].view(-1), st["neg"].view(-1)
        self.flat_grad.zero_()
        dpos, dneg = (custom_ops.take_grad(g, (pos.numel(),)) for g in grads[:2])
        give(feats, ops.logits_bwd(feats.t, self.P("item_emb.weight"), pos, neg, dpos, dneg, self.G("item_emb.weight")))
        for a, g in zip(st["acts"], grads[2:]):
            give(a, custom_ops.take_grad(g, tuple(a.t.shape)))
        tp.backward()
        return custom_ops.param_grads(self)

    @torch.no_grad()
    def predict(self, user_ids, log_seqs, item_indices, full=False, want_rank=False):
        """supersasrec.py:96-111."""
        seq = self.ids(log_seqs)
        B, L = seq.shape
        was = self.training
        self.eval()
        tp = Tape(self, self.prec, False)
        feats, _, _ = self._encode(tp, seq.view(-1), B)
        self.train(was)
        d = self.hidden_units
        cand = None if full else self.ids(item_indices)
        C = self.itemnum + 1 if full else cand.shape[1]
        logits, rank = ops.score_rank(feats.t[L - 1:], L * d, self.P("item_emb.weight"), cand, B, C, want_rank)
        return (logits, rank) if want_rank else logits

    @torch.no_grad()
    def predict_rank_candidates(self, log_seqs, item_indices, shared_list, stats=None):
        """Ranks of the positive (column 0) under EVERY block choice of `shared_list` (get_shared output per candidate) in one pass:
        (P, B).  Depth-0 layers are shared between candidates, deeper layers run once per distinct layer on the stacked inputs of
        the candidates that select them (supersearch.candidate_features)."""
        seq = self.ids(log_seqs)
        B, L = seq.shape
        d = self.hidden_units
        was = self.training
        self.eval()
        tp = Tape(self, self.prec, False)
        flat = seq.view(-1)
        x0 = self._embed(tp, flat, SITE_EMB_SEQ)

        fused = self.fused_layers()
        if fused:
            self.pack_selected(sorted({(depth, idx) for sh in shared_list for depth, (idxs, _) in enumerate(sh) for idx in idxs}))
        H = self.num_heads

        def run_layer(depth, idx, x, n):
            ids = flat if n == 1 else flat.repeat(n)
            pre = "encoder.encoder_layers.%d.%d" % (depth, idx)
            if not fused:
                y, _ = self._enc_layer(tp, pre, Act(x[0]), ids, B * n, enc_sites(depth, 0))
                return (y.t,)
            y = torch.empty_like(x[0])          # the same per-sequence kernel a single-candidate predict() runs: identical numbers
            save = torch.empty(self.lib.adt_seq_layer_save_floats(B * n, L, H, 0), device=self.dev, dtype=torch.float32)
            _lib.check(self.lib.adt_seq_enc_layer_fwd(B * n, L, H, ops._p(ids), ops._p(x[0]), ctypes.byref(self._ptrs("enc", pre, False)), ops._p(self.flat),
                                                      ops._p(self._wimg), 0.0, ops._p(self._seed), 0, 0, 0, 0, 0, ops._p(save), ops._p(y), 0.0, 0, None,
                                                      ops._stream()), "seq_enc_layer_fwd")
            return (y,)
        feats = candidate_features(run_layer, (x0.t,), shared_list, self.num_layers, stats=stats)
        self.train(was)
        P = len(shared_list)
        F = feats[0][0] if P == 1 else torch.cat([f[0] for f in feats], 0)
        full = item_indices is None
        cand = None if full else self.ids(item_indices)
        C = self.itemnum + 1 if full else cand.shape[1]
        if cand is not None and P > 1:
            cand = cand.repeat(P, 1)
        _, rank = ops.score_rank(F[L - 1:], L * d, self.P("item_emb.weight"), cand, P * B, C, True)
        return rank.view(P, B)

    def predict_rank(self, log_seqs, item_indices, want_rank=True):
        """(scores, rank of column 0) -- the interface adt_amd.sasrec.utils.evaluate_loader drives."""
        return self.predict(None, log_seqs, item_indices, full=item_indices is None, want_rank=True)

    def loss_forward_backward(self, ids, rec_w, ind_w, norms, loss_slots, b_offset=0):
        """Forward, the warm-up loss (sasrec/evolution.py:296-313) and backward into flat_grad.  ids: device int32 (seq, dec,
        pos, neg) (B, L); norms: device {n_bce, n_mse, n_nll}; loss_slots: (2 + 2*num_layers) x 64 {bce_pos, bce_neg, mse.., nll..}."""
        seq, dec, pos, neg = ids
        B, L = seq.shape
        nl, H = self.num_layers, self.num_heads
        tp = Tape(self, self.prec, self.training, row_offset=b_offset * L, b_offset=b_offset)
        feats, enc_in, recs = self._encode(tp, seq.view(-1), B)
        dec_outs = self._decode(tp, dec.view(-1), feats, B)
        E, gE = self.P("item_emb.weight"), self.G("item_emb.weight")
        pl, nlg = ops.logits_fwd(feats.t, E, pos.view(-1), neg.view(-1))
        dpos, dneg = ops.bce_seed(pl, nlg, pos.view(-1), norms, loss_slots[0:2].view(-1))
        give(feats, ops.logits_bwd(feats.t, E, pos.view(-1), neg.view(-1), dpos, dneg, gE))
        i = 0
        for i in range(nl):
            a, bq = enc_in[i], dec_outs[nl - 1 - i]
            if a.g is None:
                a.g = torch.zeros_like(a.t)
            g_b = torch.empty_like(bq.t)
            ops.mse_seed(a.t, bq.t, rec_w[i], norms, a.g, True, g_b, loss_slots[2 + i])
            give(bq, g_b)
        if H > 1:
            for l in range(nl):
                recs[l].g = torch.empty_like(recs[l].t)
                ops.nll_seed(recs[l].t, H, ind_w[i], norms, recs[l].g, loss_slots[2 + nl + l])    # stale index i (evolution.py:313)
        tp.backward()


class SuperTrainer:
    """One warm-up optimisation step of the supernet with torch.optim.Adam's per-parameter bookkeeping: only the embeddings
    and the candidate layers that were mixed in are clipped (global norm), decayed and stepped, each with its own step count."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=5.0, seed=2022):
        self.model = model
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip
        dev = model.dev
        self.m, self.v = torch.zeros_like(model.flat), torch.zeros_like(model.flat)
        self.gn2 = torch.zeros(64, device=dev, dtype=torch.float32)
        self.loss_slots = torch.zeros(2 + 2 * model.num_layers, 64, device=dev, dtype=torch.float32)
        self.steps = {}       # (lo, hi) -> Adam step count of that range
        self._emb_range = (0, model._views["encoder.encoder_layers.0.0." + _ENC[0]][0])
        self.rec_weights = [0.0] * model.num_layers
        self.ind_weights = [0.0] * model.num_layers
        model.set_seed(seed * 1000003 + 12345)

    get_weight = staticmethod(get_weight)

    def set_choice(self, cand):
        """SearcherEvolution._set_choice (evolution.py:139-153): probabilities -> loss weights + the model's block choice."""
        m = self.model
        block, rw, iw = cand_to_block(m.rec_choice, m.ind_choice, cand)
        self.rec_weights[:], self.ind_weights[:] = rw, iw
        m.set_choice(block)

    def step(self, seq, dec, pos, neg):
        m = self.model
        m.train()
        ids = tuple(m.ids(a) for a in (seq, dec, pos, neg))
        B, L = ids[0].shape
        T = B * L
        n_bce = float(np.count_nonzero(np.asarray(pos)))
        norms = torch.tensor([n_bce, float(T * m.hidden_units), float(T * m.num_heads)], device=m.dev, dtype=torch.float32)
        m._seed.add_(-1640531535)
        self.loss_slots.zero_()
        m.flat_grad.zero_()
        m.loss_forward_backward(ids, self.rec_weights, self.ind_weights, norms, self.loss_slots)
        ops.grad_sumsq(m.flat_grad, self.gn2)
        ranges = [self._emb_range]
        for depth, (idxs, _) in enumerate(m.shared):
            for idx in sorted(set(idxs)):
                ranges += [m.layer_range("encoder", depth, idx), m.layer_range("decoder", depth, idx)]
        for lo, hi in ranges:
            t = self.steps.get((lo, hi), 0) + 1
            self.steps[(lo, hi)] = t
            ops.adam_range(m.flat[lo:hi], m.flat_grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.wd, self.clip, self.lr, self.betas[0],
                           self.betas[1], self.eps, t, self.gn2)

    def loss(self):
        m = self.model
        nl = m.num_layers
        s = self.loss_slots.sum(1)
        w = [1.0, 1.0] + list(self.rec_weights) + [self.ind_weights[nl - 1] if m.num_heads > 1 else 0.0] * nl
        return (s * torch.tensor(w, device=m.dev, dtype=torch.float32)).sum()

    def grad_norm(self):
        return self.gn2.sum().sqrt()
