"""DisenDistSASupernet on the MI355X hot path -- drop-in for the reference's stosa/supernet.py (+ SuperDistSAEncoder /
SuperDistSADecoder, stosa/super_modules.py:63-136): the weight-sharing STOSA-ADT supernet that stosa/searcher.py warms up and
scores.

Per depth there are rec_size * ind_size candidate encoder and decoder layers; `set_choice(block_cand)` selects four of them and their
bilinear weights.  Unlike the SASRec / BERT4Rec supernets the four selected layers are CHAINED: the reference's inner loop rebinds
`mean_hidden_states, cov_hidden_states` to each candidate's output, so candidate k+1 reads candidate k's output, and the four
intermediate results are mixed with the weights (super_modules.py:81-95, :123-133; encoder and decoder alike).  The head-classifier
scores are mixed and log_softmax'd (:96-97); they are mixed as log-probabilities here, which is the same function (each candidate's
log-sum-exp is constant along the softmax axis).  The supernet has no decLayerNorm (supernet.py:10-19).

Every candidate layer runs on the stage kernels of DisenDistSAModel (adt_amd/stosa/models.py); `SuperStosaTrainer.step` is the loop
body of SuperDistSAModelTrainer.iteration (stosa/super_trainer.py:205-235): BPR + pvn + lambda1[l] * MSE (mean and covariance) +
lambda2[l] * NLL (mean and covariance), Adam with coupled weight decay, torch's bookkeeping for parameters whose grad is None.
"""
import numpy as np
import torch

from .. import ops
from ..supersearch import candidate_features, cand_to_block, get_shared
from ..wide import Act, Tape
from .models import SITE_EMB, DisenDistSAModel, dec_sites, enc_sites

CAND_SITE = 4096


def _cand_sites(sites, k):
    return {n: v + CAND_SITE * (k + 1) for n, v in sites.items()}


class DisenDistSASupernet(DisenDistSAModel):
    def __init__(self, args, rec_choice, ind_choice):
        self.rec_choice, self.ind_choice = np.asarray(rec_choice, np.float64), np.asarray(ind_choice, np.float64)
        self.block = len(self.rec_choice) * len(self.ind_choice)
        super().__init__(args, block=self.block, dec_layernorm=False)     # init_weights is the plain model's (supernet.py:101-111)
        self.shared = [((0, 0, 0, 0), (0.0, 0.0, 0.0, 0.0)) for _ in range(self.num_layers)]

    def set_choice(self, cand):
        """supernet.py:113-115 / super_modules.py:45-58 (encoder and decoder receive the same choice)."""
        self.shared = get_shared(self.rec_choice, self.ind_choice, np.asarray(cand, np.float64))

    def layer_range(self, kind, depth, cand):
        """Flat [lo, hi) of the trained tensors of one candidate layer (consecutive in the table; a decoder layer's dec_attention
        lives in the untrained tail)."""
        p = "%s.layer.%d.%d." % (kind, depth, cand)
        names = [n for n, _ in self.table if n.startswith(p) and ".dec_attention." not in n]
        lo = self._views[names[0]][0]
        o, n, _ = self._views[names[-1]]
        return lo, o + (n + 3) // 4 * 4

    def shared_ranges(self):
        """Flat ranges torch's optimizer would step: the embeddings + LayerNorm, and the selected candidate layers."""
        ranges = [(0, self._views["item_encoder.layer.0.0.attention.mean_query.weight"][0])]
        for depth, (idxs, _) in enumerate(self.shared):
            for idx in sorted(set(idxs)):
                ranges += [self.layer_range("item_encoder", depth, idx), self.layer_range("item_decoder", depth, idx)]
        return ranges

    def _finetune(self, tp, inp, dec, B):
        """finetune's token-major body with the super encoder / decoder (supernet.py:55-99, super_modules.py:75-136)."""
        m, c = self._embed(tp, inp, "mean", SITE_EMB["seq_mean"]), self._embed(tp, inp, "cov", SITE_EMB["seq_cov"])
        dm, dc = self._embed(tp, dec, "mean", SITE_EMB["dec_mean"]), self._embed(tp, dec, "cov", SITE_EMB["dec_cov"])
        enc_inputs, enc_recs, dec_outs = [], [], []
        H = self.num_heads
        for i, (idxs, ws) in enumerate(self.shared):
            enc_inputs.append((m, c))
            ms, cs, rms, rcs = [], [], [], []
            for k, (idx, w) in enumerate(zip(idxs, ws)):
                m, c, rm, rc = self._enc_layer(tp, "item_encoder.layer.%d.%d" % (i, idx), m, c, inp, B, _cand_sites(enc_sites(i), k))   # chained
                ms.append((m, float(w))); cs.append((c, float(w))); rms.append((rm, float(w))); rcs.append((rc, float(w)))
            m, c = tp.mix(ms), tp.mix(cs)
            enc_recs.append((tp.log_softmax(tp.mix(rms), H), tp.log_softmax(tp.mix(rcs), H)))
        tp.mark_decoder_start()
        for i, (idxs, ws) in enumerate(self.shared):
            ms, cs = [], []
            for k, (idx, w) in enumerate(zip(idxs, ws)):
                dm, dc = self._dec_layer(tp, "item_decoder.layer.%d.%d" % (i, idx), dm, dc, m, c, inp, B, _cand_sites(dec_sites(i), k))  # chained
                ms.append((dm, float(w))); cs.append((dc, float(w)))
            dm, dc = tp.mix(ms), tp.mix(cs)
            dec_outs.append((dm, dc))
        return m, c, enc_inputs, enc_recs, dec_outs

    @torch.no_grad()
    def predict_full_candidates(self, input_ids, shared_list, stats=None):
        """Full-sort Wasserstein distances of the last state to every item under EVERY block choice of `shared_list`: (P * B, item_size),
        candidate-major.  One pass: chain prefixes of depth 0 are shared between candidates, deeper links run once per distinct layer on
        the stacked inputs (supersearch.candidate_features, chain=True)."""
        inp = self.ids(input_ids)
        B, L = inp.shape
        was = self.training
        self.eval()
        tp = Tape(self, self.prec, False)
        flat = inp.view(-1)
        m0, c0 = self._embed(tp, flat, "mean", SITE_EMB["seq_mean"]), self._embed(tp, flat, "cov", SITE_EMB["seq_cov"])

        def run_layer(depth, idx, x, n):
            ids = flat if n == 1 else flat.repeat(n)
            m, c, _, _ = self._enc_layer(tp, "item_encoder.layer.%d.%d" % (depth, idx), Act(x[0]), Act(x[1]), ids, B * n, enc_sites(depth))
            return (m.t, c.t)
        feats = candidate_features(run_layer, (m0.t, c0.t), shared_list, self.num_layers, chain=True, stats=stats)
        self.train(was)
        P = len(shared_list)
        M = feats[0][0] if P == 1 else torch.cat([f[0] for f in feats], 0)
        C = feats[0][1] if P == 1 else torch.cat([f[1] for f in feats], 0)
        rows = torch.arange(L - 1, P * B * L, L, device=self.dev, dtype=torch.int32)
        return ops.wdist_full(ops.gather_rows(M, rows), ops.gather_rows(C, rows), self.P("item_mean_embeddings.weight"),
                              self.P("item_cov_embeddings.weight"), self.item_size)


class SuperStosaTrainer:
    """One warm-up optimisation step of the supernet (stosa/super_trainer.py:205-235) with torch.optim.Adam's per-parameter bookkeeping:
    no gradient clipping, coupled weight decay, candidates that were not selected keep their moments and step counts."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, seed=42):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        dev = model.dev
        self.m, self.v = torch.zeros_like(model.flat), torch.zeros_like(model.flat)
        self.gn2 = torch.zeros(64, device=dev, dtype=torch.float32)
        nl = model.num_layers
        self.loss_slots = torch.zeros(3 + 4 * nl, 64, device=dev, dtype=torch.float32)
        self.steps = {}
        self.rec_weights, self.ind_weights = [0.0] * nl, [0.0] * nl
        model.set_seed(seed * 1000003 + 12345)

    def set_choice(self, cand):
        """SearcherEvolution._set_choice (stosa/searcher.py:88-102)."""
        m = self.model
        block, rw, iw = cand_to_block(m.rec_choice, m.ind_choice, cand)
        self.rec_weights[:], self.ind_weights[:] = rw, iw
        m.set_choice(block)

    def step(self, input_ids, dec_ids, pos_ids, neg_ids):
        m = self.model
        m.train()
        st = m.stage(input_ids, dec_ids, pos_ids, neg_ids)
        T = st["B"] * m.maxlen
        norms = torch.tensor([0.0, float(T * m.hidden_units), float(T * m.num_heads)], device=m.dev, dtype=torch.float32)
        m._seed.add_(-1640531535)
        self.loss_slots.zero_()
        m.flat_grad.zero_()
        m.loss_forward_backward(st, self.rec_weights, self.ind_weights, norms, self.loss_slots)
        ops.grad_sumsq(m.flat_grad, self.gn2)
        for lo, hi in m.shared_ranges():
            t = self.steps.get((lo, hi), 0) + 1
            self.steps[(lo, hi)] = t
            ops.adam_range(m.flat[lo:hi], m.flat_grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.wd, 1e30, self.lr, self.betas[0], self.betas[1],
                           self.eps, t, self.gn2)

    def loss(self):
        nl = self.model.num_layers
        s = self.loss_slots.sum(1)
        w = [1.0, 1.0, 0.0]
        for l in range(nl):
            w += [self.rec_weights[l]] * 2
        for l in range(nl):
            w += [self.ind_weights[l]] * 2
        return (s * torch.tensor(w, device=self.model.dev, dtype=torch.float32)).sum()

    def grad_norm(self):
        return self.gn2.sum().sqrt()
