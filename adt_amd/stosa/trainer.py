"""Fused training step and full-sort evaluation for STOSA-ADT: the train branch of the reference's
DistSAModelTrainer.iteration (stosa/trainer.py:525-560: finetune, bpr_optimization, reconstruction + independence terms,
backward, Adam -- no gradient clipping) as one device-side launch sequence, optionally replayed from a HIP graph, and the
full-sort branch (:583-612: distance of the last state to every item, seen items masked, 40 smallest).

Data-parallel (one process per GPU, RCCL over xGMI): batch rows shard across ranks with GLOBAL normalisers (sum of
istarget, B*L*d, B*L*H of the whole batch) and GLOBAL dropout indices; one sum all-reduce of the flat gradient buffer;
Adam runs identically on every rank on the reduced buffer over the trained prefix only (the parameters the reference
leaves at grad=None are outside it).
"""
import numpy as np
import torch

from .. import ops
from ..dp import GradBuckets, reduce_sum, capture


class FusedStosaTrainer:
    def __init__(self, model, lambda1, lambda2, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None, use_graph=False,
                 seed=42):
        self.model = model
        self.lambda1, self.lambda2 = [float(x) for x in lambda1], [float(x) for x in lambda2]
        nl = model.num_layers
        assert len(self.lambda1) == nl and len(self.lambda2) == nl
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        self.use_graph = use_graph       # data-parallel steps are captured too (RCCL collectives are graph nodes)
        self._buckets = GradBuckets(model.flat_grad, model.offset_of("item_decoder.layer.0.enc_attention.mean_query.weight"), process_group,
                                    n=model.n_trained_floats)
        dev = model.dev
        self.m = torch.zeros_like(model.flat)
        self.v = torch.zeros_like(model.flat)
        self.scal = torch.zeros(192, device=dev, dtype=torch.float32)
        self.loss_slots = torch.zeros(3 + 4 * nl, 64, device=dev, dtype=torch.float32)
        w = [1.0, 1.0, 0.0]
        for l in range(nl):
            w += [self.lambda1[l], self.lambda1[l]]
        for l in range(nl):
            w += [self.lambda2[l], self.lambda2[l]]
        self._loss_w = torch.tensor(w, device=dev, dtype=torch.float32)
        model.set_seed(seed * 1000003 + 12345)
        self.nstep = 0
        self._graph = None
        self._st = None

    def stage(self, input_ids, dec_ids, pos_ids, neg_ids, n_target_global=None, norms_scale=1):
        m = self.model
        st = m.stage(input_ids, dec_ids, pos_ids, neg_ids, n_target_global)
        T = st["B"] * m.maxlen
        st["norms"] = torch.tensor([0.0, float(norms_scale * T * m.hidden_units), float(norms_scale * T * m.num_heads)], device=m.dev,
                                   dtype=torch.float32)
        return st

    def _launch(self, b_offset):
        m, st = self.model, self._st
        m._seed.add_(-1640531535)    # += 0x9E3779B1 (mod 2^32): a fresh dropout stream every step, on the device
        self.loss_slots.zero_()
        m.flat_grad.zero_()
        m.dp_hook = self._buckets.tail_ready if self._buckets.active else None
        m.loss_forward_backward(st, self.lambda1, self.lambda2, st["norms"], self.loss_slots, b_offset)
        self._buckets.finish()
        n = m.n_trained_floats
        # no clip_grad_norm_ in the reference (trainer.py:557-559): clip = inf
        ops.clip_adam_l2(m.flat, m.flat_grad, self.m, self.v, self.wd, 1e30, self.lr, self.betas[0], self.betas[1], self.eps, self.scal, n=n)

    def _copy_stage(self, st):
        if self._st is None or self._st["B"] != st["B"]:
            self._st = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
            self._graph = None
            return
        for k, v in st.items():
            if isinstance(v, torch.Tensor):
                self._st[k].copy_(v, non_blocking=True)
            else:
                self._st[k] = v

    def step_staged(self, st, b_offset=0):
        self.model.train()
        self._copy_stage(st)
        self.nstep += 1
        if not self.use_graph:
            self._launch(b_offset)
            return
        if self._graph is None:
            self._launch(b_offset)          # warm up eagerly (hipFuncSetAttribute is not capturable), then capture
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with capture(self._graph):
                self._launch(b_offset)
            return
        self._graph.replay()

    def step(self, input_ids, dec_ids, pos_ids, neg_ids, n_target_global=None, b_offset=0, norms_scale=1):
        self.step_staged(self.stage(input_ids, dec_ids, pos_ids, neg_ids, n_target_global, norms_scale), b_offset)

    def loss(self):
        """Device scalar: the loss of the last step as the reference accumulates it (trainer.py:561)."""
        return (self.loss_parts() * self._loss_w).sum()

    def loss_parts(self):
        """{bpr, pvn (weighted), auc, mse.., nll..} of the last step (summed over the ranks: each holds its shard's partial sums)."""
        slots = self.loss_slots.sum(1)
        return reduce_sum(slots, self.pg) if self.world > 1 else slots

    def grad_norm(self):
        return self.scal[1].sqrt()

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def full_sort(self, batches, topk=40):
        """Full-sort evaluation (stosa/trainer.py:583-612) over an iterable of (input_ids (B, L), seen, answers (B, A)): rank all
        items by ascending distance with the seen items pushed to 1e24 and keep `topk`, all on the device (adt_wdist_full +
        adt_topk_masked); only the (B, topk) ids come back.  `seen` is the users' rows of the train/valid rating matrix as a scipy
        CSR matrix, a dense (B, item_size) 0/1 array, or None.  Returns (pred_list (N, topk), answers (N, A)) for
        get_full_sort_score."""
        preds, answers = [], []
        dev = self.model.dev
        for input_ids, seen, ans in batches:
            dist = self.model.predict_full(input_ids)
            indptr = indices = None
            if seen is not None:
                if hasattr(seen, "tocsr"):
                    csr = seen.tocsr()
                    ip, ix = csr.indptr, csr.indices
                else:
                    rows, cols = np.nonzero(np.asarray(seen))
                    ip = np.zeros(dist.shape[0] + 1, np.int64)
                    np.cumsum(np.bincount(rows, minlength=dist.shape[0]), out=ip[1:])
                    ix = cols
                indptr = torch.from_numpy(np.ascontiguousarray(ip, dtype=np.int32)).to(dev)
                indices = torch.from_numpy(np.ascontiguousarray(ix, dtype=np.int32)).to(dev)
                if indices.numel() == 0:
                    indptr = indices = None
            preds.append(ops.topk_masked(dist, topk, indptr, indices).cpu().numpy().astype(np.int64))
            answers.append(np.asarray(ans))
        return np.concatenate(preds), np.concatenate(answers)


def recall_at_k(actual, predicted, topk):
    """stosa/utils.py:228-242: mean over the users that have answers of |top-k hits| / |answers|."""
    s, n = 0.0, 0
    for a, p in zip(actual, predicted):
        a = set(int(x) for x in a)
        if a:
            s += len(a & set(int(x) for x in p[:topk])) / float(len(a))
            n += 1
    return s / n


def ndcg_k(actual, predicted, topk):
    """stosa/utils.py:327-345 (ideal DCG over min(topk, |answers|) positions, 1.0 when that is empty)."""
    res = 0.0
    for a, p in zip(actual, predicted):
        aset = set(int(x) for x in a)
        idcg = sum(1.0 / np.log2(i + 2) for i in range(min(topk, len(a)))) or 1.0
        res += sum(1.0 / np.log2(j + 2) for j in range(topk) if int(p[j]) in aset) / idcg
    return res / float(len(actual))


def cal_mrr(actual, predicted):
    """stosa/utils.py:244-267: reciprocal rank of the first hit in the predicted list, averaged over all users."""
    s = 0.0
    for a, p in zip(actual, predicted):
        aset = set(int(x) for x in a)
        hits = [j for j, it in enumerate(p) if int(it) in aset]
        if hits:
            s += 1.0 / (hits[0] + 1)
    return s / float(len(predicted))


def get_full_sort_score(answers, pred_list):
    """Trainer.get_full_sort_score (stosa/trainer.py:62-86): [HIT@1, NDCG@1, HIT@5, NDCG@5, ... @10, @15, @20, @40, MRR]."""
    out = []
    for k in (1, 5, 10, 15, 20, 40):
        out += [recall_at_k(answers, pred_list, k), ndcg_k(answers, pred_list, k)]
    return out + [cal_mrr(answers, pred_list)]
