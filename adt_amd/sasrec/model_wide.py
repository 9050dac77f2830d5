"""SASRecADT at the widths the fused 64-wide executor does not cover -- in particular the reference's shipped ml-1m
template (sasrec/templates/ml-1m.json:11-12: hidden_units 256, 2 heads => head size 128).  Same constructor, forward 5-tuple,
predict and state_dict names/shapes as sasrec/model.py:SASRecADT; the layers run on the general stage kernels (dense layers
on MFMA: adt_gemm.cuh; causal attention: adt_attn_gen.cuh, chunked backward at head size 128; LayerNorm, head classifier,
embedding and logits kernels) strung together by the launch tape of adt_amd/wide.py.  `WideSasrecTrainer.step()` is the loop
body of sasrec/main.py:146-173 on the device (HIP-graph capturable, data-parallel with one gradient all-reduce).
"""
import math

import numpy as np
import torch

from .. import _lib, custom_ops, ops
from ..dp import GradBuckets, reduce_sum, capture
from ..wide import Act, FlatModule, Tape, give
from .model import REF_ORDER, param_table

LN_EPS = 1e-8
SITE_EMB_SEQ, SITE_EMB_DEC = 1, 2


def enc_sites(i):
    b = 16 + 8 * i
    return {"attn": b, "ffn1": b + 1, "ffn2": b + 2}


def dec_sites(i):
    b = 128 + 8 * i
    return {"slf": b, "enc": b + 1, "ffn1": b + 2, "ffn2": b + 3}


NREP_MAX = 16             # replicas of the item-table gradient (contention relief, as in the fused executor)
REP_BUDGET = 64 << 20     # bytes of zero-fill + reduce per step we are willing to spend on them


def n_replicas(table_floats):
    """Replica count for a (V+1, d) item table: 16 for ml-1m (3.5 MB at d=256), fewer as the table grows, 1 (scatter straight into
    the gradient, no replica traffic at all) from 32 MB up -- Amazon-Beauty's 54,542 x 256 table is 56 MB, and its ids are spread
    over 16x more rows, so the same-address atomic chains the replicas exist to break are short anyway."""
    return int(max(1, min(NREP_MAX, REP_BUDGET // (4 * int(table_floats)))))


class SASRecADTWide(FlatModule):
    def __init__(self, user_num, item_num, args):
        super().__init__()
        self.user_num, self.item_num = user_num, item_num
        self.num_heads, self.maxlen, self.num_layers = args.num_heads, args.maxlen, args.num_layers
        self.hidden_units, self.dropout = args.hidden_units, _lib.dropout_rate(args.dropout, "dropout")
        self.args = args
        self.prec = {"f32": ops.PREC_F32, "fp32": ops.PREC_F32, "bf16": ops.PREC_BF16}[getattr(args, "precision", "bf16")]
        d, H = self.hidden_units, self.num_heads
        if d % 64 or d > 256 or (d // H) not in (16, 32, 64, 128) or self.maxlen > 256:
            raise _lib.AdtError("SASRecADT (adt_amd, wide path): hidden_units in {64,128,192,256}, head size 16..128, maxlen <= 256; got d=%d H=%d L=%d"
                                % (d, H, self.maxlen))
        self._build_flat(param_table(item_num, args.maxlen, d, H, args.num_layers), args.device, REF_ORDER)   # item table first: adt_clip_adam's wd term
        g = torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (1 << 31))
        for name, shape in self.table:
            v = self.P(name)
            if name.endswith("norm.weight"):
                v.fill_(1.0)
            elif len(shape) >= 2:
                bound = (1.0 / max(shape[1] * (shape[2] if len(shape) > 2 else 1), 1)) ** 0.5
                v.copy_((torch.rand(shape, generator=g) * 2 - 1) * bound)

    # ---- layers (sasrec/modules.py:644-677) ------------------------------------------------------------------------------
    def _embed(self, tp, ids, site):
        P, G = self.P, self.G
        L = self.maxlen
        p = tp.p_eff(self.dropout)
        x = Act(ops.embed_fwd(ids, P("item_emb.weight"), P("pos_emb.weight"), L, p, self._seed, site, tp.row_offset))

        def bw():
            if x.g is None:
                return
            rep = getattr(tp, "item_rep", None)
            if rep is None:
                ops.embed_bwd(ids, x.g, L, p, self._seed, site, G("item_emb.weight"), G("pos_emb.weight"), tp.row_offset)
            else:      # item rows through the zeroed replicas (popular items: same-address float atomics serialise), positions directly
                ops.item_scatter(ids, x.g, None, float(self.hidden_units) ** 0.5, p, self._seed, site, tp.row_offset, rep, self._nrep, self._rep_stride)
                ops.posemb_bwd(ids, x.g, L, p, self._seed, site, tp.row_offset, G("pos_emb.weight"))
        tp.bw.append(bw)
        return x

    def _attn(self, tp, q, kv, B, site, qkv=None):
        d, H, L = self.hidden_units, self.num_heads, self.maxlen
        p = tp.p_eff(self.dropout)
        if qkv is not None:
            Q, K, V = qkv.t[:, :d], qkv.t[:, d:2 * d], qkv.t[:, 2 * d:]
        else:
            Q, K, V = q.t, kv.t[:, :d], kv.t[:, d:]
        fill = float("-inf")     # the float causal mask of sasrec/modules.py:504-507
        O, LSE = ops.attn_masked_fwd(self.prec, Q, K, V, B, H, L, True, None, fill, p, self._seed, site, tp.b_offset)
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
            ops.attn_masked_bwd(self.prec, Q, K, V, O, LSE, o.g, B, H, L, True, None, fill, p, self._seed, site, tp.b_offset, out=out)
        tp.bw.append(bw)
        return o

    def _conv(self, name, grad=False):
        d = self.hidden_units
        return (self.G(name) if grad else self.P(name)).view(d, d)

    def _ln(self, tp, x, p):
        return tp.layernorm(x, self.P(p + ".weight"), self.P(p + ".bias"), self.G(p + ".weight"), self.G(p + ".bias"), LN_EPS)

    def _lin(self, tp, x, p, rows=None, **kw):
        W, b, gW, gb = self.P(p + "weight"), self.P(p + "bias"), self.G(p + "weight"), self.G(p + "bias")
        if rows is not None:
            W, b, gW, gb = W[rows], b[rows], gW[rows], gb[rows]
        return tp.dense(x, W, b, gW, gb, **kw)

    def _ffn(self, tp, p, x, st, ids, R2=None):
        c1, c2 = p + ".conv1", p + ".conv2"
        f1 = tp.dense(x, self._conv(c1 + ".weight"), self.P(c1 + ".bias"), self._conv(c1 + ".weight", True), self.G(c1 + ".bias"), act=ops.ACT_RELU,
                      p=self.dropout, site=st["ffn1"])
        return tp.dense(f1, self._conv(c2 + ".weight"), self.P(c2 + ".bias"), self._conv(c2 + ".weight", True), self.G(c2 + ".bias"), p=self.dropout,
                        site=st["ffn2"], R=x, R2=R2, mask_ids=ids)

    def _enc_layer(self, tp, p, x, ids, B, st):
        d = self.hidden_units
        Q = self._ln(tp, x, p + ".attention_layernorm")
        ip = p + ".attention_layer.in_proj_"
        q = self._lin(tp, Q, ip, slice(0, d))                     # q from LN(x) ...
        kv = self._lin(tp, x, ip, slice(d, 3 * d))                # ... k, v from the un-normalised x (modules.py:646-647)
        o = self._attn(tp, q, kv, B, st["attn"])
        rec = tp.headcls(o, self.P(p + ".sparse.weight"), self.P(p + ".sparse.bias"), self.G(p + ".sparse.weight"), self.G(p + ".sparse.bias"))
        h = self._lin(tp, o, p + ".attention_layer.out_proj.", R=Q)
        h2 = self._ln(tp, h, p + ".forward_layernorm")
        return self._ffn(tp, p + ".forward_layer", h2, st, ids), rec

    def _dec_layer(self, tp, p, x, enc, ids, B, st):
        d = self.hidden_units
        D = self._ln(tp, x, p + ".layer_norm")
        qkv = self._lin(tp, D, p + ".slf_attn.in_proj_")
        a1 = self._lin(tp, self._attn(tp, None, None, B, st["slf"], qkv=qkv), p + ".slf_attn.out_proj.")
        q2 = self._lin(tp, a1, p + ".enc_attn.in_proj_", slice(0, d))
        kv2 = self._lin(tp, enc, p + ".enc_attn.in_proj_", slice(d, 3 * d))
        a2 = self._lin(tp, self._attn(tp, q2, kv2, B, st["enc"]), p + ".enc_attn.out_proj.")
        return self._ffn(tp, p + ".pos_ffn", a2, st, ids, R2=D)   # dec_input + (a2 + ffn(a2)), modules.py:672-673

    def _encode(self, tp, seq, B):
        x = self._embed(tp, seq, SITE_EMB_SEQ)
        enc_in, recs = [], []
        for i in range(self.num_layers):
            enc_in.append(x)
            x, rec = self._enc_layer(tp, "encoder.encoder_layers.%d" % i, x, seq, B, enc_sites(i))
            recs.append(rec)
        return self._ln(tp, x, "last_layernorm"), enc_in, recs       # sasrec/model.py:48

    def _decode(self, tp, dec, feats, B):
        y = self._embed(tp, dec, SITE_EMB_DEC)
        tp.mark_decoder_start()
        outs = []
        for i in range(self.num_layers):
            y = self._dec_layer(tp, "decoder.decoder_layers.%d" % i, y, feats, dec, B, dec_sites(i))
            outs.append(y)
        return outs

    # ---- reference API -----------------------------------------------------------------------------------------------------
    def forward(self, user_ids, log_seqs, dec_seqs, pos_seqs, neg_seqs):
        """sasrec/model.py:67-81 -> (pos_logits, neg_logits, encoder_layer_input, decoder_layer_output [reversed], rec_layer_ind
        [natural (b, l) row order]).  Under autograd (grad mode on, parameters requiring grad) the tensors are wired into it through
        the adt_amd::model_forward custom operator, so the reference's loop body (sasrec/main.py:146-173) runs on them unchanged;
        WideSasrecTrainer.step() is the fused, faster way to train."""
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
        """Reverse of _op_forward for the output gradients autograd hands over; returns one gradient per parameter."""
        tp, feats, pos, neg = st["tp"], st["feats"], st["pos"].view(-1), st["neg"].view(-1)
        self.flat_grad.zero_()
        tp.item_rep = None          # embedding rows straight into the table gradient (no replicas on this path)
        dpos, dneg = (custom_ops.take_grad(g, (pos.numel(),)) for g in grads[:2])
        give(feats, ops.logits_bwd(feats.t, self.P("item_emb.weight"), pos, neg, dpos, dneg, self.G("item_emb.weight")))
        for a, g in zip(st["acts"], grads[2:]):
            give(a, custom_ops.take_grad(g, tuple(a.t.shape)))
        tp.backward()
        H = self.num_heads
        return custom_ops.param_grads(self, lambda n: "pos_ffn_layernorm" in n or (H == 1 and ".sparse." in n))

    @torch.no_grad()
    def predict_rank(self, log_seqs, item_indices, want_rank=True):
        seq = self.ids(log_seqs)
        B, L = seq.shape
        was = self.training
        self.eval()
        tp = Tape(self, self.prec, False)
        feats, _, _ = self._encode(tp, seq.view(-1), B)
        self.train(was)
        d = self.hidden_units
        cand = None if item_indices is None else self.ids(item_indices)
        C = self.item_num + 1 if cand is None else cand.shape[1]
        return ops.score_rank(feats.t[L - 1:], L * d, self.P("item_emb.weight"), cand, B, C, want_rank)

    def predict(self, user_ids, log_seqs, item_indices, full=False):
        """sasrec/model.py:83-97."""
        return self.predict_rank(log_seqs, None if full else item_indices, want_rank=False)[0]

    def loss_forward_backward(self, ids, lambdas1, lambdas2, norms, loss_slots, b_offset=0):
        """Forward, loss seeds (sasrec/main.py:151-169: BCE, lambda1[i] * MSE, lambda2[stale i] * NLL) and backward into
        flat_grad; the wd * ||E||_F term is added by adt_clip_adam.  ids: device int32 (seq, dec, pos, neg) (B, L)."""
        seq, dec, pos, neg = ids
        B, L = seq.shape
        nl, H = self.num_layers, self.num_heads
        tp = Tape(self, self.prec, self.training, row_offset=b_offset * L, b_offset=b_offset)
        feats, enc_in, recs = self._encode(tp, seq.view(-1), B)
        dec_outs = self._decode(tp, dec.view(-1), feats, B)
        E, gE = self.P("item_emb.weight"), self.G("item_emb.weight")
        pl, nlg = ops.logits_fwd(feats.t, E, pos.view(-1), neg.view(-1))
        dpos, dneg = ops.bce_seed(pl, nlg, pos.view(-1), norms, loss_slots[0:2].view(-1))
        # item-table gradient through NREP zeroed replicas, reduced once after the backward (as the fused executor does)
        n_table = gE.numel()
        if getattr(self, "_rep", None) is None:
            self._nrep = n_replicas(n_table)
            self._rep_stride = (n_table + 3) // 4 * 4
            self._rep = gE.view(-1) if self._nrep == 1 else torch.zeros(self._nrep * self._rep_stride, device=self.dev, dtype=torch.float32)
        elif self._nrep > 1:
            self._rep.zero_()
        tp.item_rep = self._rep
        give(feats, ops.logits_bwd_df(E, pos.view(-1), neg.view(-1), dpos, dneg))
        ops.item_scatter(pos.view(-1), feats.t, dpos, 1.0, 0.0, None, 0, 0, self._rep, self._nrep, self._rep_stride)
        ops.item_scatter(neg.view(-1), feats.t, dneg, 1.0, 0.0, None, 0, 0, self._rep, self._nrep, self._rep_stride)
        i = 0
        for i in range(nl):
            a, bq = enc_in[i], dec_outs[nl - 1 - i]
            if a.g is None:
                a.g = torch.zeros_like(a.t)
            g_b = torch.empty_like(bq.t)
            ops.mse_seed(a.t, bq.t, lambdas1[i], norms, a.g, True, g_b, loss_slots[2 + i])
            give(bq, g_b)
        if H > 1:
            for l in range(nl):
                recs[l].g = torch.empty_like(recs[l].t)
                ops.nll_seed(recs[l].t, H, lambdas2[i], norms, recs[l].g, loss_slots[2 + nl + l])     # stale index (main.py:169)
        tp.backward()
        if self._nrep > 1:
            ops.replica_reduce(gE, self._rep, self._nrep, self._rep_stride)


class WideSasrecTrainer:
    def __init__(self, model, lambdas1, lambdas2, lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, clip=5.0, process_group=None,
                 use_graph=False, seed=23):
        self.model = model
        self.lambdas1, self.lambdas2 = [float(x) for x in lambdas1], [float(x) for x in lambdas2]
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.use_graph = use_graph       # data-parallel steps are captured too (RCCL collectives are graph nodes)
        dev = model.dev
        self._buckets = GradBuckets(model.flat_grad, model.offset_of("decoder.decoder_layers.0.layer_norm.weight"), process_group)
        self.m, self.v = torch.zeros_like(model.flat), torch.zeros_like(model.flat)
        self.scal = torch.zeros(192, device=dev, dtype=torch.float32)
        nl = model.num_layers
        self.loss_slots = torch.zeros(2 + 2 * nl, 64, device=dev, dtype=torch.float32)
        w = [1.0, 1.0] + self.lambdas1 + [self.lambdas2[nl - 1] if model.num_heads > 1 else 0.0] * nl
        self._loss_w = torch.tensor(w, device=dev, dtype=torch.float32)
        model.set_seed(seed * 1000003 + 12345)
        self._graph, self._st = None, None
        self.nstep = 0

    def stage(self, seq, dec, pos, neg, norms=None):
        m = self.model
        ids = tuple(m.ids(a) for a in (seq, dec, pos, neg))
        B, L = ids[0].shape
        if norms is None:
            norms = (float(np.count_nonzero(np.asarray(pos))), float(self.world * B * L * m.hidden_units), float(self.world * B * L * m.num_heads))
        return {"B": B, "seq": ids[0], "dec": ids[1], "pos": ids[2], "neg": ids[3], "norms": torch.tensor(norms, device=m.dev, dtype=torch.float32)}

    def _launch(self, b_offset):
        m, st = self.model, self._st
        m._seed.add_(-1640531535)
        self.loss_slots.zero_()
        m.flat_grad.zero_()
        m.dp_hook = self._buckets.tail_ready if self._buckets.active else None
        m.loss_forward_backward((st["seq"], st["dec"], st["pos"], st["neg"]), self.lambdas1, self.lambdas2, st["norms"], self.loss_slots, b_offset)
        self._buckets.finish()
        ops.clip_adam(m.flat, m.flat_grad, self.m, self.v, (m.item_num + 1) * m.hidden_units, self.wd, self.clip, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.scal)

    def step(self, seq, dec, pos, neg, norms=None, b_offset=0):
        self.model.train()
        self.nstep += 1
        st = self.stage(seq, dec, pos, neg, norms)
        if self._st is None or self._st["B"] != st["B"]:
            self._st = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
            self._graph = None
        else:
            for k, v in st.items():
                if isinstance(v, torch.Tensor):
                    self._st[k].copy_(v, non_blocking=True)
        if not self.use_graph:
            return self._launch(b_offset)
        if self._graph is None:
            self._launch(b_offset)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with capture(self._graph):
                self._launch(b_offset)
            return
        self._graph.replay()

    def loss(self):
        slots = self.loss_slots.sum(1)
        if self.world > 1:
            slots = reduce_sum(slots, self.pg)
        return (slots * self._loss_w).sum() + self.scal[3]

    def grad_norm(self):
        return self.scal[1].sqrt()
