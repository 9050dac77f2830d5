"""Fused training step for BERT4Rec-ADT: the loop body of the reference's BertTrainer.train (bert4rec/trainer.py:100-138:
forward, CE + reconstruction + independence losses, backward, clip_grad_norm_, Adam with coupled weight decay) as one
device-side launch sequence, optionally replayed from a HIP graph, plus the evaluate() ranking metrics (:49-86).

Data-parallel (one process per GPU, torch.distributed "nccl" = RCCL over xGMI): every rank runs the same sequence on
its contiguous slice of the batch with GLOBAL loss normalisers (count of labels != 0, B*L*d, B*L*H of the whole batch)
and GLOBAL dropout indices; the flat gradient buffer is summed with one all-reduce; clipping and Adam run identically
on every rank on the reduced buffer.
"""
import numpy as np
import torch

from .. import ops
from ..dp import GradBuckets, reduce_sum, capture


class FusedBertTrainer:
    def __init__(self, model, lambda1, lambda2, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=5.0, process_group=None,
                 use_graph=False, seed=23, mcap_frac=1.0):
        self.model = model
        self.lambda1, self.lambda2 = [float(x) for x in lambda1], [float(x) for x in lambda2]
        assert len(self.lambda1) == model.num_layers and len(self.lambda2) == model.num_layers
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        self.use_graph = use_graph       # data-parallel steps are captured too (RCCL collectives are graph nodes)
        # tail bucket: decoder layers + output head (their backward runs first); head bucket: embeddings (word_emb also receives
        # the all-item-logits gradient) + encoder, reduced at the end
        self._buckets = GradBuckets(model.flat_grad, model.offset_of("decoder.decoder_layers.0.dec_multi_head_attention.query_transfer.weight"),
                                    process_group)
        dev = model.dev
        self.m = torch.zeros_like(model.flat)
        self.v = torch.zeros_like(model.flat)
        self.scal = torch.zeros(192, device=dev, dtype=torch.float32)
        nl = model.num_layers
        self.loss_slots = torch.zeros(1 + 2 * nl, 64, device=dev, dtype=torch.float32)
        self._loss_w = torch.tensor([1.0] + self.lambda1 + self.lambda2, device=dev, dtype=torch.float32)
        self.norms = torch.zeros(3, device=dev, dtype=torch.float32)
        self.mcap_frac = float(mcap_frac)   # the masked-row GEMMs are launched for at most this fraction of B*L rows
        model.set_seed(seed * 1000003 + 12345)
        self.nstep = 0
        self._graph = None
        self._st = None

    # ------------------------------------------------------------------------------------------------------------------
    def stage(self, src, dec, labels, n_valid_global=None, norms_scale=1):
        """Upload one batch; n_valid_global / norms_scale give the GLOBAL normalisers under data parallelism."""
        m = self.model
        st = m.stage(src, dec, labels, n_valid_global)
        B = st["B"]
        T = B * m.maxlen
        st["norms"] = torch.tensor([0.0, float(norms_scale * T * m.hidden_units), float(norms_scale * T * m.num_heads)], device=m.dev,
                                   dtype=torch.float32)
        cap = int(np.ceil(self.mcap_frac * T))
        over = st["M_host"] > cap
        if self.world > 1:      # every rank must reach the same decision, or the ranks that go on hang in the gradient all-reduce
            t = torch.tensor([1.0 if over else 0.0], device=m.dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.pg)
            over = bool(t.item() > 0)
        if over:
            raise ValueError("a rank's shard has more masked rows than the capacity %d (this rank: %d; mcap_frac=%g)" % (cap, st["M_host"], self.mcap_frac))
        return st

    def _launch(self, b_offset):
        m, st = self.model, self._st
        m._seed.add_(-1640531535)    # += 0x9E3779B1 (mod 2^32): a fresh dropout stream every step, on the device
        self.loss_slots.zero_()
        m.flat_grad.zero_()
        T = st["B"] * m.maxlen
        m.dp_hook = self._buckets.tail_ready if self._buckets.active else None
        m.loss_forward_backward(st, self.lambda1, self.lambda2, st["norms"], self.loss_slots, b_offset, int(np.ceil(self.mcap_frac * T)))
        self._buckets.finish()
        ops.clip_adam_l2(m.flat, m.flat_grad, self.m, self.v, self.wd, self.clip, self.lr, self.betas[0], self.betas[1], self.eps, self.scal)

    def _copy_stage(self, st):
        """Graph replays read fixed buffers: copy the new batch into the captured ones."""
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

    def step(self, src, dec, labels, n_valid_global=None, b_offset=0, norms_scale=1):
        self.step_staged(self.stage(src, dec, labels, n_valid_global, norms_scale), b_offset)

    def loss(self):
        """Device scalar: the loss of the last step as BertTrainer prints it (trainer.py:139)."""
        return (self.loss_parts() * self._loss_w).sum()

    def loss_parts(self):
        slots = self.loss_slots.sum(1)
        return reduce_sum(slots, self.pg) if self.world > 1 else slots

    def grad_norm(self):
        return self.scal[1].sqrt()

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, batches, ks=(5, 10)):
        """BertTrainer.evaluate (trainer.py:49-86) over an iterable of (seq (B, L), candidates (B, C)) with the positive in
        column 0: rank = number of candidates scored above it, HR@k, NDCG@k, and the AUC with the reference's
        candidates_size = 1 + C."""
        self.model.eval()
        ks = tuple(ks)
        # additive statistics [n, sum of AUC terms, per k: hits, sum 1/log2(rank+2)]: under data parallelism rank r scores batches
        # r, r+W, ... and the statistics are sum-reduced in float64, so every rank ends up with the metrics of the whole user set
        stats = np.zeros(2 + 2 * len(ks), np.float64)
        for i, (seq, cand) in enumerate(batches):
            if i % self.world != self.rank:
                continue
            _, rank = self.model.predict(None, seq, None, None, cand, want_rank=True)
            r = rank.cpu().numpy().astype(np.int64)
            size = 1 + np.asarray(cand).shape[1]
            stats[0] += len(r)
            stats[1] += float(((size - (r + 1)) / (size - 1)).sum())
            for j, k in enumerate(ks):
                stats[2 + 2 * j] += float((r < k).sum())
                stats[3 + 2 * j] += float((1.0 / np.log2(r[r < k] + 2)).sum())
        if self.world > 1:
            t = torch.from_numpy(stats).to(self.model.dev)
            stats = reduce_sum(t, self.pg).cpu().numpy()
        n = stats[0]
        ndcg = {k: float(stats[3 + 2 * j] / n) for j, k in enumerate(ks)}
        hr = {k: float(stats[2 + 2 * j] / n) for j, k in enumerate(ks)}
        return (ndcg, hr), float(stats[1] / n)
