"""DisenDistSAModel on the MI355X hot path -- drop-in for the reference's stosa/models.py:DisenDistSAModel (STOSA-ADT:
stochastic mean/covariance item embeddings, Wasserstein-distance self-attention, reconstruction decoder, per-head
independence classifiers on both the mean and the covariance contexts).

Same constructor (`DisenDistSAModel(args)`), `finetune(input_ids, dec_ids, user_ids)` 7-tuple and state_dict names /
shapes, `item_mean_embeddings` / `item_cov_embeddings` reachable as attributes (the reference's trainer
reads them directly, stosa/trainer.py:361-364,466-467).  All arithmetic runs in libadt_hip.so: dense layers
(adt_gemm.cuh), the Wasserstein attention and the BPR / full-sort distance kernels (adt_stosa.cuh, exact fp32 wave-level
reductions), LayerNorm / dropout / ELU row kernels.
"""
import numpy as np
import torch

from .. import _lib, custom_ops, ops
from ..wide import Act, FlatModule, Tape, give

LN_EPS = 1e-12
SITE_EMB = {"seq_mean": 1, "seq_cov": 2, "dec_mean": 3, "dec_cov": 4}


def enc_sites(i):
    b = 16 + 8 * i
    return {"attn": b, "out_mean": b + 1, "out_cov": b + 2, "ffn_mean": b + 3, "ffn_cov": b + 4}


def dec_sites(i):
    b = 128 + 8 * i
    return {"attn": b, "out_mean": b + 1, "out_cov": b + 2, "ffn_mean": b + 3, "ffn_cov": b + 4}


# path components in the order the reference's constructors register them (stosa/models.py:169-178, modules.py)
REF_ORDER = ["item_mean_embeddings", "item_cov_embeddings", "position_mean_embeddings", "position_cov_embeddings", "user_margins",
             "item_encoder", "item_decoder", "layer", "attention", "dec_attention", "enc_attention", "mean_intermediate", "cov_intermediate",
             "mean_independence_layer", "cov_independence_layer", "mean_query", "cov_query", "mean_key", "cov_key", "mean_value",
             "cov_value", "mean_dense", "cov_dense", "dense_1", "dense_2", "LayerNorm", "decLayerNorm", "weight", "bias"]


def param_table(item_size, maxlen, d, H, nl, num_users, block=0, dec_layernorm=True):
    """([(state_dict name, shape)] in flat order, index of the first parameter outside the loss graph).  block > 0: the supernet's
    layout, `block` candidate layers per depth named item_encoder.layer.<depth>.<candidate>.* (stosa/super_modules.py:66-72);
    dec_layernorm: the plain model's unused decLayerNorm (stosa/models.py:176; the supernet has none).  Within an
    attention block the mean q/k/v weights (then biases), and the covariance ones, are consecutive so that each triple
    runs as one GEMM.  Parameters the reference never trains (grad None: user_margins, decLayerNorm, every decoder
    layer's dec_attention, stosa/modules.py:537-538) sit at the end, outside the optimizer's prefix."""
    hd = d // H

    def att(p):
        o = []
        for kind in ("mean", "cov"):
            o += [(p + ".%s_%s.weight" % (kind, n), (d, d)) for n in ("query", "key", "value")]
            o += [(p + ".%s_%s.bias" % (kind, n), (d,)) for n in ("query", "key", "value")]
        for n in ("mean_dense", "cov_dense"):
            o += [(p + "." + n + ".weight", (d, d)), (p + "." + n + ".bias", (d,))]
        return o + [(p + ".LayerNorm.weight", (d,)), (p + ".LayerNorm.bias", (d,))]

    def inter(p):
        return [(p + ".dense_1.weight", (4 * d, d)), (p + ".dense_1.bias", (4 * d,)), (p + ".dense_2.weight", (d, 4 * d)), (p + ".dense_2.bias", (d,)),
                (p + ".LayerNorm.weight", (d,)), (p + ".LayerNorm.bias", (d,))]

    t = [("item_mean_embeddings.weight", (item_size, d)), ("item_cov_embeddings.weight", (item_size, d)),
         ("position_mean_embeddings.weight", (maxlen, d)), ("position_cov_embeddings.weight", (maxlen, d)),
         ("LayerNorm.weight", (d,)), ("LayerNorm.bias", (d,))]
    prefixes = lambda kind, i: ["%s.layer.%d" % (kind, i)] if block == 0 else ["%s.layer.%d.%d" % (kind, i, c) for c in range(block)]
    for i in range(nl):
        for p in prefixes("item_encoder", i):
            t += att(p + ".attention") + inter(p + ".mean_intermediate") + inter(p + ".cov_intermediate")
            t += [(p + ".mean_independence_layer.weight", (H, hd)), (p + ".mean_independence_layer.bias", (H,)),
                  (p + ".cov_independence_layer.weight", (H, hd)), (p + ".cov_independence_layer.bias", (H,))]
    for i in range(nl):
        for p in prefixes("item_decoder", i):
            t += att(p + ".enc_attention") + inter(p + ".mean_intermediate") + inter(p + ".cov_intermediate")
    n_trained = len(t)
    t += [("user_margins.weight", (num_users, 1))]
    if dec_layernorm:
        t += [("decLayerNorm.weight", (d,)), ("decLayerNorm.bias", (d,))]
    for i in range(nl):
        for p in prefixes("item_decoder", i):
            t += att(p + ".dec_attention")
    return t, n_trained


class DisenDistSAModel(FlatModule):
    def __init__(self, args, block=0, dec_layernorm=True):
        super().__init__()
        self.args = args
        self.item_size, self.maxlen, self.hidden_units = args.item_size, args.maxlen, args.hidden_units
        self.num_heads, self.num_layers = args.num_heads, args.num_layers
        self.dropout, self.attention_dropout = _lib.dropout_rate(args.dropout, "dropout"), _lib.dropout_rate(args.attention_dropout, "attention_dropout")
        self.prec = {"f32": ops.PREC_F32, "fp32": ops.PREC_F32, "bf16": ops.PREC_BF16}[getattr(args, "precision", "bf16")]
        d, H = self.hidden_units, self.num_heads
        if d % 64 or (d // H) not in (16, 32, 64):
            raise _lib.AdtError("DisenDistSAModel (adt_amd): hidden_units must be a multiple of 64 with head size 16/32/64, got d=%d H=%d" % (d, H))
        if getattr(args, "distance_metric", "wasserstein") != "wasserstein":
            raise _lib.AdtError("DisenDistSAModel (adt_amd): only distance_metric='wasserstein' is built")
        table, n_trained = param_table(args.item_size, args.maxlen, d, H, args.num_layers, args.num_users, block, dec_layernorm)
        self._build_flat(table, getattr(args, "device", "cuda:0"), REF_ORDER)
        self.n_trained_floats = self._views[table[n_trained][0]][0]     # optimizer prefix (flat floats)
        # init_weights (stosa/models.py:262-272): N(0.01, initializer_range) on Linear/Embedding weights, LayerNorm 1/0, biases 0
        g = torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (1 << 31))
        std = float(getattr(args, "initializer_range", 0.02))
        for name, shape in self.table:
            v = self.P(name)
            if "LayerNorm.weight" in name:
                v.fill_(1.0)
            elif name.endswith(".weight"):
                v.copy_(0.01 + std * torch.randn(shape, generator=g))

    # ------------------------------------------------------------------------------------------------------------------
    def _embed(self, tp, ids, which, site):
        """add_position_mean_embedding / add_position_cov_embedding (stosa/models.py:183-210)."""
        P, G = self.P, self.G
        L = self.maxlen
        en, pn = "item_%s_embeddings.weight" % which, "position_%s_embeddings.weight" % which
        x0 = Act(ops.embed_sum_fwd(ids, P(en), P(pn), L))

        def bw():   # runs after the LayerNorm / dropout closures registered below
            if x0.g is None:
                return
            T, d = x0.g.shape
            _lib.check(self.lib.adt_item_scatter(ops._p(ids), ops._p(x0.g), d, None, T, d, 1.0, 0.0, None, 0, 0, ops._p(G(en)), 1, 0, ops._stream()),
                       "item_scatter")   # padding_idx = 0: rows of id 0 are skipped
            _lib.check(self.lib.adt_posemb_bwd(ops._p(self._ones(T)), ops._p(x0.g), T, L, d, 0.0, None, 0, 0, ops._p(G(pn)), ops._stream()), "posemb_bwd")
        tp.bw.append(bw)
        z = tp.layernorm(x0, P("LayerNorm.weight"), P("LayerNorm.bias"), G("LayerNorm.weight"), G("LayerNorm.bias"), LN_EPS)
        return tp.dropact(z, self.dropout, site, ops.ACT_ELU1 if which == "cov" else ops.ACT_ELU)

    def _ones(self, T):
        if getattr(self, "_ones_buf", None) is None or self._ones_buf.numel() < T:
            self._ones_buf = torch.ones(T, device=self.dev, dtype=torch.int32)
        return self._ones_buf

    def _attention(self, tp, p, mq, cq, mkv, ckv, key_ids, B, st):
        """DistAttention.forward / DistEDAttention.forward (stosa/modules.py:222-275, 311-361) -> (mean_hidden, cov_hidden,
        mean context, cov context)."""
        P, G, sp = self.P, self.G, self.span
        d, H, L = self.hidden_units, self.num_heads, self.maxlen
        pa = tp.p_eff(self.attention_dropout)

        def proj(x, kind, first, last, n, act):
            w0, w1 = p + ".%s_%s.weight" % (kind, first), p + ".%s_%s.weight" % (kind, last)
            b0, b1 = p + ".%s_%s.bias" % (kind, first), p + ".%s_%s.bias" % (kind, last)
            return tp.dense(x, sp(w0, w1, (n * d, d)), sp(b0, b1, (n * d,)), sp(w0, w1, (n * d, d), grad=True), sp(b0, b1, (n * d,), grad=True), act=act)
        if mq is mkv:
            pm, pc = proj(mq, "mean", "query", "value", 3, ops.ACT_NONE), proj(cq, "cov", "query", "value", 3, ops.ACT_ELU1)
            Qm, Km, Vm = pm.t[:, :d], pm.t[:, d:2 * d], pm.t[:, 2 * d:]
            Qc, Kc, Vc = pc.t[:, :d], pc.t[:, d:2 * d], pc.t[:, 2 * d:]
            holders = ((pm,), (pc,))
        else:
            qm, qc = proj(mq, "mean", "query", "query", 1, ops.ACT_NONE), proj(cq, "cov", "query", "query", 1, ops.ACT_ELU1)
            km, kc = proj(mkv, "mean", "key", "value", 2, ops.ACT_NONE), proj(ckv, "cov", "key", "value", 2, ops.ACT_ELU1)
            Qm, Km, Vm = qm.t, km.t[:, :d], km.t[:, d:]
            Qc, Kc, Vc = qc.t, kc.t[:, :d], kc.t[:, d:]
            holders = ((qm, km), (qc, kc))
        Om, Oc, LSE = ops.wattn_fwd(Qm, Qc, Km, Kc, Vm, Vc, key_ids, B, H, L, pa, self._seed, st["attn"], tp.b_offset, prec=self.prec)
        om, oc = Act(Om), Act(Oc)

        def bw():
            if om.g is None and oc.g is None:
                return
            T = Om.shape[0]
            gom = om.g if om.g is not None else torch.zeros_like(Om)
            goc = oc.g if oc.g is not None else torch.zeros_like(Oc)
            gm = torch.empty(T, 3 * d, device=self.dev, dtype=torch.float32)
            gc = torch.empty(T, 3 * d, device=self.dev, dtype=torch.float32)
            ops.wattn_bwd(Qm, Qc, Km, Kc, Vm, Vc, key_ids, Om, Oc, LSE, gom, goc, B, H, L, pa, self._seed, st["attn"], tp.b_offset,
                          out=(gm[:, :d], gc[:, :d], gm[:, d:2 * d], gc[:, d:2 * d], gm[:, 2 * d:], gc[:, 2 * d:]), prec=self.prec)
            for hold, g in ((holders[0], gm), (holders[1], gc)):
                if len(hold) == 1:
                    hold[0].g = g
                else:
                    hold[0].g, hold[1].g = g[:, :d], g[:, d:]
        tp.bw.append(bw)
        lw, lb, glw, glb = P(p + ".LayerNorm.weight"), P(p + ".LayerNorm.bias"), G(p + ".LayerNorm.weight"), G(p + ".LayerNorm.bias")
        zm = tp.dense(om, P(p + ".mean_dense.weight"), P(p + ".mean_dense.bias"), G(p + ".mean_dense.weight"), G(p + ".mean_dense.bias"),
                      p=self.dropout, site=st["out_mean"], R=mq)
        hm = tp.layernorm(zm, lw, lb, glw, glb, LN_EPS)
        zc = tp.dense(oc, P(p + ".cov_dense.weight"), P(p + ".cov_dense.bias"), G(p + ".cov_dense.weight"), G(p + ".cov_dense.bias"),
                      p=self.dropout, site=st["out_cov"], R=cq)
        hc = tp.layernorm(zc, lw, lb, glw, glb, LN_EPS)
        return hm, hc, om, oc

    def _intermediate(self, tp, p, x, site, elu1):
        """DistIntermediate.forward (stosa/modules.py:485-494); elu1: the covariance branch's ELU(.)+1 (:522,540)."""
        P, G = self.P, self.G
        h = tp.dense(x, P(p + ".dense_1.weight"), P(p + ".dense_1.bias"), G(p + ".dense_1.weight"), G(p + ".dense_1.bias"), act=ops.ACT_ELU)
        z = tp.dense(h, P(p + ".dense_2.weight"), P(p + ".dense_2.bias"), G(p + ".dense_2.weight"), G(p + ".dense_2.bias"), p=self.dropout, site=site, R=x)
        y = tp.layernorm(z, P(p + ".LayerNorm.weight"), P(p + ".LayerNorm.bias"), G(p + ".LayerNorm.weight"), G(p + ".LayerNorm.bias"), LN_EPS)
        return tp.dropact(y, 0.0, 0, ops.ACT_ELU1) if elu1 else y

    def _enc_layer(self, tp, p, m, c, inp, B, st):
        """DistLayer.forward (stosa/modules.py:518-525) -> (mean, cov, mean / cov head-classifier log-probabilities)."""
        P, G = self.P, self.G
        hm, hc, om, oc = self._attention(tp, p + ".attention", m, c, m, c, inp, B, st)
        rm = tp.headcls(om, P(p + ".mean_independence_layer.weight"), P(p + ".mean_independence_layer.bias"),
                        G(p + ".mean_independence_layer.weight"), G(p + ".mean_independence_layer.bias"))
        rc = tp.headcls(oc, P(p + ".cov_independence_layer.weight"), P(p + ".cov_independence_layer.bias"),
                        G(p + ".cov_independence_layer.weight"), G(p + ".cov_independence_layer.bias"))
        m2 = self._intermediate(tp, p + ".mean_intermediate", hm, st["ffn_mean"], False)
        c2 = self._intermediate(tp, p + ".cov_intermediate", hc, st["ffn_cov"], True)
        return m2, c2, rm, rc

    def _dec_layer(self, tp, p, dm, dc, m, c, inp, B, st):
        """DistDecLayer.forward (modules.py:535-541): dec_attention's output is discarded by the reference, so it is not computed;
        enc_attention takes queries from the decoder input and keys / values (and the mask) from the encoder side."""
        hm, hc, _, _ = self._attention(tp, p + ".enc_attention", dm, dc, m, c, inp, B, st)
        return (self._intermediate(tp, p + ".mean_intermediate", hm, st["ffn_mean"], False),
                self._intermediate(tp, p + ".cov_intermediate", hc, st["ffn_cov"], True))

    def _finetune(self, tp, inp, dec, B):
        """Token-major body of finetune (stosa/models.py:212-260)."""
        P, G = self.P, self.G
        m, c = self._embed(tp, inp, "mean", SITE_EMB["seq_mean"]), self._embed(tp, inp, "cov", SITE_EMB["seq_cov"])
        dm, dc = self._embed(tp, dec, "mean", SITE_EMB["dec_mean"]), self._embed(tp, dec, "cov", SITE_EMB["dec_cov"])
        enc_inputs, enc_recs, dec_outs = [], [], []
        for i in range(self.num_layers):
            enc_inputs.append((m, c))
            m, c, rm, rc = self._enc_layer(tp, "item_encoder.layer.%d" % i, m, c, inp, B, enc_sites(i))
            enc_recs.append((rm, rc))
        tp.mark_decoder_start()
        for i in range(self.num_layers):
            dm, dc = self._dec_layer(tp, "item_decoder.layer.%d" % i, dm, dc, m, c, inp, B, dec_sites(i))
            dec_outs.append((dm, dc))
        return m, c, enc_inputs, enc_recs, dec_outs

    def finetune(self, input_ids, dec_ids, user_ids=None):
        """stosa/models.py:212-260 -> (mean_out, cov_out, att_scores=None, margins, enc_inputs, enc_recs, dec_outputs).  The
        (B, H, L, L) attention probabilities are never materialised (no caller reads them: trainer.py:534,585).  Under autograd the
        tensors are wired into it (adt_amd::model_forward) and `item_mean_embeddings` / `item_cov_embeddings` / `user_margins` are
        callable like the reference's nn.Embedding modules, so the reference's loop body (stosa/trainer.py:534-559 with
        bpr_optimization :358-391) runs unchanged; FusedStosaTrainer.step is the fast way to train."""
        ids = [self.ids(input_ids), self.ids(dec_ids)]
        B, L = ids[0].shape
        if custom_ops.wants_grad(self):
            outs = custom_ops.forward_with_grad(self, ids)
        else:
            with torch.no_grad():
                outs, _ = self._op_forward(ids, self.training)
        nl = self.num_layers
        margins = None if user_ids is None else self.user_margins(user_ids)
        pairs = [[outs[2 + 2 * i], outs[3 + 2 * i]] for i in range(3 * nl)]
        return outs[0], outs[1], None, margins, pairs[:nl], pairs[nl:2 * nl], pairs[2 * nl:]

    def _op_forward(self, ids, training):
        inp, dec = ids
        B, L = inp.shape
        d, H = self.hidden_units, self.num_heads
        if training:
            self.next_seed()
        tp = Tape(self, self.prec, training)
        m, c, enc_inputs, enc_recs, dec_outs = self._finetune(tp, inp.view(-1), dec.view(-1), B)
        acts = [m, c] + [x for pr in enc_inputs for x in pr] + [x for pr in enc_recs for x in pr] + [x for pr in dec_outs for x in pr]
        nl = self.num_layers
        outs = []
        for k, a in enumerate(acts):
            is_rec = 2 + 2 * nl <= k < 2 + 4 * nl
            outs.append(a.t.view(B, L, H, H) if is_rec else a.t.view(B, L, d))
        return outs, {"tp": tp, "acts": acts}

    def _op_backward(self, st, grads):
        self.flat_grad.zero_()
        for a, g in zip(st["acts"], grads):
            give(a, custom_ops.take_grad(g, tuple(a.t.shape)))
        st["tp"].backward()
        base = self.n_trained_floats
        return custom_ops.param_grads(self, lambda n: self._views[n][0] >= base)     # the reference leaves these at grad None

    @torch.no_grad()
    def predict_full(self, input_ids, dec_ids=None):
        """Full-sort scores (stosa/trainer.py:583-595, dist_predict_full :464-479): Wasserstein distance of the last state to
        every item, (B, item_size)."""
        inp = self.ids(input_ids)
        dec = inp if dec_ids is None else self.ids(dec_ids)
        B, L = inp.shape
        was = self.training
        self.eval()
        tp = Tape(self, self.prec, False)
        m, c, _, _, _ = self._finetune(tp, inp.view(-1), dec.view(-1), B)
        rows = torch.arange(L - 1, B * L, L, device=self.dev, dtype=torch.int32)
        sm, sc = ops.gather_rows(m.t, rows), ops.gather_rows(c.t, rows)
        self.train(was)
        return ops.wdist_full(sm, sc, self.P("item_mean_embeddings.weight"), self.P("item_cov_embeddings.weight"), self.item_size)

    # ------------------------------------------------------------------------------------------------------------------
    def stage(self, input_ids, dec_ids, pos_ids, neg_ids, n_target_global=None):
        inp, dec, pos, neg = (np.ascontiguousarray(np.asarray(a), dtype=np.int32) for a in (input_ids, dec_ids, pos_ids, neg_ids))
        nt = float(max(int((pos > 0).sum()), 1) if n_target_global is None else n_target_global)
        return {"B": inp.shape[0], "inp": self.ids(inp), "dec": self.ids(dec), "pos": self.ids(pos), "neg": self.ids(neg),
                "inv_count": torch.tensor([1.0 / nt], device=self.dev, dtype=torch.float32)}

    def loss_forward_backward(self, st, lambda1, lambda2, norms, loss_slots, b_offset=0):
        """Forward, the loss assembly of DistSAModelTrainer.iteration (stosa/trainer.py:534-556) and backward into flat_grad.
        norms: device {_, n_mse, n_nll}; loss_slots: (3 + 4*num_layers) x 64 floats {bpr, pvn, auc, mse (mean, cov) per
        layer.., nll (mean, cov) per layer..}."""
        B, L, d, H, nl = st["B"], self.maxlen, self.hidden_units, self.num_heads, self.num_layers
        tp = Tape(self, self.prec, self.training, row_offset=b_offset * L, b_offset=b_offset)
        inp, dec = st["inp"].view(-1), st["dec"].view(-1)
        m, c, enc_inputs, enc_recs, dec_outs = self._finetune(tp, inp, dec, B)
        dsm, dsc = ops.wdist_bpr(m.t, c.t, self.P("item_mean_embeddings.weight"), self.P("item_cov_embeddings.weight"), st["pos"].view(-1),
                                 st["neg"].view(-1), float(self.args.pvn_weight), st["inv_count"], self.G("item_mean_embeddings.weight"),
                                 self.G("item_cov_embeddings.weight"), loss_slots[0:3].view(-1))
        give(m, dsm)
        give(c, dsc)
        for l in range(nl):
            for t in (0, 1):
                a, bq = enc_inputs[l][t], dec_outs[nl - 1 - l][t]      # dec_outputs.reverse() (trainer.py:540)
                if a.g is None:
                    a.g = torch.zeros_like(a.t)
                g_b = torch.empty_like(bq.t)
                ops.mse_seed(a.t, bq.t, lambda1[l], norms, a.g, True, g_b, loss_slots[3 + 2 * l + t])
                give(bq, g_b)
        for l in range(nl):
            for t in (0, 1):
                r = enc_recs[l][t]
                r.g = torch.empty_like(r.t)
                ops.nll_seed(r.t, H, lambda2[l], norms, r.g, loss_slots[3 + 2 * nl + 2 * l + t])
        tp.backward()
