"""SuperBertModel on the MI355X hot path -- drop-in for the reference's bert4rec/model/superbert.py (+ SuperEncoder / SuperDecoder,
bert4rec/model/modules.py:217-259, :356-393, base_super_modules.py): the weight-sharing BERT4Rec-ADT supernet that
bert4rec/evolution.py warms up and scores.

Differences from BertModel that the reference's supernet has and this one keeps: the vocabulary is itemnum + 2 (superbert.py:21), the
feed-forward width is 4 * hidden_units (:35), every depth holds rec_size * ind_size candidate encoder and decoder layers;
`set_choice(block_cand)` selects four of them per depth.  The encoder mixes the four candidates' outputs and head-classifier scores
with the bilinear weights and applies log_softmax to the mixed scores (modules.py:250-258); the DECODER sums its four candidates
WITHOUT the weights (modules.py:386-390: `seqs_list.append(c_logits)`).  The classifier scores are mixed as log-probabilities here:
log_softmax(sum_k w_k (z_k - lse_k)) == log_softmax(sum_k w_k z_k) because each lse_k is constant along the softmax axis.

Every candidate layer runs on the same stage kernels as BertModel (adt_amd/bert4rec/model.py); `SuperBertTrainer.step` is the loop
body of SearcherEvolution._train_warmup (bert4rec/evolution.py:266-296): CE over the masked positions + rec_weights[i] * MSE +
ind_weights[stale i] * NLL, clip_grad_norm_, torch.optim.AdamW (decoupled decay) with torch's bookkeeping for parameters whose grad is
None (candidates that were not selected keep their moments, their own step count, and are not decayed).
"""
import math

import numpy as np
import torch

from .. import ops
from ..supersearch import candidate_features, cand_to_block, get_shared
from ..wide import Act, Tape
from .model import SITE_EMB_DEC, SITE_EMB_SEQ, BertModel, dec_sites, enc_sites

CAND_SITE = 4096


def _cand_sites(sites, k):
    """Dropout sites of candidate slot k of a depth: every selected layer draws its own stream."""
    return {n: v + CAND_SITE * (k + 1) for n, v in sites.items()}


class SuperBertModel(BertModel):
    def __init__(self, usernum, itemnum, rec_choice, ind_choice, args):
        self.rec_choice, self.ind_choice = np.asarray(rec_choice, np.float64), np.asarray(ind_choice, np.float64)
        self.block = len(self.rec_choice) * len(self.ind_choice)
        super().__init__(usernum, itemnum, args, vocab=itemnum + 2, inner_units=4 * args.hidden_units, block=self.block)
        # SearcherEvolution._generate_supernet (bert4rec/evolution.py:104-112): trunc_normal_(std = initializer_range) on every tensor
        # whose name holds neither 'layer_norm' nor 'bias'; the others keep torch's constructor defaults (LayerNorm 1 / 0, mask_bias 0,
        # Linear bias U(-1/sqrt(fan_in), 1/sqrt(fan_in)))
        g = torch.Generator(device="cpu").manual_seed(torch.initial_seed() % (1 << 31))
        std = float(getattr(args, "initializer_range", 0.02))
        shapes = dict(self.table)
        for name, shape in self.table:
            v = self.P(name)
            if "layer_norm" in name:
                v.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name == "mask_bias":
                v.zero_()
            elif "bias" in name:
                fan_in = shapes[name[:-4] + "weight"][1]
                v.copy_((torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in))
            else:
                t = torch.empty(shape)
                torch.nn.init.trunc_normal_(t, std=std, generator=g)
                v.copy_(t)
        self.shared = [((0, 0, 0, 0), (0.0, 0.0, 0.0, 0.0)) for _ in range(self.num_layers)]

    def set_choice(self, cand):
        """superbert.py:121-123 / base_super_modules.py:42-57."""
        self.shared = get_shared(self.rec_choice, self.ind_choice, np.asarray(cand, np.float64))

    def layer_range(self, kind, depth, cand):
        """Flat [lo, hi) of one candidate layer (its tensors are consecutive in the table)."""
        p = "%s.%s_layers.%d.%d." % (kind, kind, depth, cand)
        names = [n for n, _ in self.table if n.startswith(p)]
        lo = self._views[names[0]][0]
        o, n, _ = self._views[names[-1]]
        return lo, o + (n + 3) // 4 * 4

    def shared_ranges(self):
        """Flat ranges torch's optimizer would step: everything outside the candidate layers + the selected candidates."""
        first = self._views["encoder.encoder_layers.0.0.multi_head_attention.query_transfer.weight"][0]
        tail = self._views["mask_trans_feat.weight"][0]
        ranges = [(0, first), (tail, self.flat.numel())]
        for depth, (idxs, _) in enumerate(self.shared):
            for idx in sorted(set(idxs)):
                ranges += [self.layer_range("encoder", depth, idx), self.layer_range("decoder", depth, idx)]
        return ranges

    # ------------------------------------------------------------------------------------------------------------------
    def _encode(self, tp, src, B):
        """log2feats + SuperEncoder.forward (superbert.py:66-73, modules.py:243-259)."""
        x = self._embed(tp, src, SITE_EMB_SEQ)
        enc_inputs, recs = [], []
        for i, (idxs, ws) in enumerate(self.shared):
            enc_inputs.append(x)
            outs, inds = [], []
            for k, (idx, w) in enumerate(zip(idxs, ws)):
                y, rec = self._enc_layer(tp, "encoder.encoder_layers.%d.%d" % (i, idx), x, src, B, _cand_sites(enc_sites(i), k))
                outs.append((y, float(w)))
                inds.append((rec, float(w)))
            x = tp.mix(outs)
            recs.append(tp.log_softmax(tp.mix(inds), self.num_heads))
        return x, enc_inputs, recs

    def _decode(self, tp, dec, src, enc, B):
        """decode + SuperDecoder.forward (superbert.py:75-84, modules.py:382-393): the four candidates are SUMMED, unweighted."""
        x = self._embed(tp, dec, SITE_EMB_DEC)
        tp.mark_decoder_start()
        outs = []
        for i, (idxs, _) in enumerate(self.shared):
            parts = [(self._dec_layer(tp, "decoder.decoder_layers.%d.%d" % (i, idx), x, dec, src, enc, B, _cand_sites(dec_sites(i), k)), 1.0)
                     for k, idx in enumerate(idxs)]
            x = tp.mix(parts)
            outs.append(x)
        return outs

    @torch.no_grad()
    def predict_rank_candidates(self, seqs, candidates, shared_list, stats=None):
        """Ranks of the positive (column 0 of `candidates`) at the last position under EVERY block choice of `shared_list`: (P, B).
        One pass: depth-0 layers shared between candidates, deeper layers once per distinct layer on the stacked inputs
        (supersearch.candidate_features)."""
        src = self.ids(seqs)
        cand = self.ids(candidates)
        B, L = src.shape
        was = self.training
        self.eval()
        tp = Tape(self, self.prec, False)
        flat = src.view(-1)
        x0 = self._embed(tp, flat, SITE_EMB_SEQ)

        def run_layer(depth, idx, x, n):
            ids = flat if n == 1 else flat.repeat(n)
            y, _ = self._enc_layer(tp, "encoder.encoder_layers.%d.%d" % (depth, idx), Act(x[0]), ids, B * n, enc_sites(depth))
            return (y.t,)
        feats = candidate_features(run_layer, (x0.t,), shared_list, self.num_layers, stats=stats)
        P = len(shared_list)
        F = feats[0][0] if P == 1 else torch.cat([f[0] for f in feats], 0)
        rows = torch.arange(L - 1, P * B * L, L, device=self.dev, dtype=torch.int32)
        h = self._head(tp, Act(ops.gather_rows(F, rows)))
        self.train(was)
        if P > 1:
            cand = cand.repeat(P, 1)
        _, rank = ops.score_rank_bias(h.t, self.hidden_units, self.P("item_emb.word_emb.weight"), self.P("mask_bias"), cand, P * B, cand.shape[1], True)
        return rank.view(P, B)


class SuperBertTrainer:
    """One warm-up optimisation step of the supernet (bert4rec/evolution.py:266-296) with torch.optim.AdamW's per-parameter bookkeeping."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, clip=5.0, seed=2022):
        self.model = model
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip
        dev = model.dev
        self.m, self.v = torch.zeros_like(model.flat), torch.zeros_like(model.flat)
        self.gn2 = torch.zeros(64, device=dev, dtype=torch.float32)
        self.loss_slots = torch.zeros(1 + 2 * model.num_layers, 64, device=dev, dtype=torch.float32)
        self.steps = {}       # (lo, hi) -> AdamW step count of that range
        self.rec_weights = [0.0] * model.num_layers
        self.ind_weights = [0.0] * model.num_layers
        model.set_seed(seed * 1000003 + 12345)

    def set_choice(self, cand):
        """SearcherEvolution._set_choice (bert4rec/evolution.py:119-133)."""
        m = self.model
        block, rw, iw = cand_to_block(m.rec_choice, m.ind_choice, cand)
        self.rec_weights[:], self.ind_weights[:] = rw, iw
        m.set_choice(block)

    def step(self, src, dec, labels):
        m = self.model
        m.train()
        st = m.stage(src, dec, labels)
        T = st["B"] * m.maxlen
        norms = torch.tensor([0.0, float(T * m.hidden_units), float(T * m.num_heads)], device=m.dev, dtype=torch.float32)
        m._seed.add_(-1640531535)
        self.loss_slots.zero_()
        m.flat_grad.zero_()
        nl = m.num_layers
        lam2 = [self.ind_weights[nl - 1]] * nl        # `ind_weights[i]` with the reconstruction loop's stale i (evolution.py:291)
        m.loss_forward_backward(st, self.rec_weights, lam2, norms, self.loss_slots)
        ops.grad_sumsq(m.flat_grad, self.gn2)
        for lo, hi in m.shared_ranges():
            t = self.steps.get((lo, hi), 0) + 1
            self.steps[(lo, hi)] = t
            ops.adamw_range(m.flat[lo:hi], m.flat_grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.wd, self.clip, self.lr, self.betas[0],
                            self.betas[1], self.eps, t, self.gn2)

    def loss(self):
        m = self.model
        nl = m.num_layers
        s = self.loss_slots.sum(1)
        w = [1.0] + list(self.rec_weights) + [self.ind_weights[nl - 1] if m.num_heads > 1 else 0.0] * nl
        return (s * torch.tensor(w, device=m.dev, dtype=torch.float32)).sum()

    def grad_norm(self):
        return self.gn2.sum().sqrt()
