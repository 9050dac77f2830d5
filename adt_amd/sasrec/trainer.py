"""Fused training step for SASRecADT: the loop body of the reference's sasrec/main.py:143-173 (forward, loss
assembly, backward, weight-decay term, clip_grad_norm_, Adam) as ONE device-side launch sequence, optionally
replayed from a HIP graph.  Host work per step = one pinned-buffer fill + one H2D copy of the id batch.

Data-parallel (one process per GPU, torch.distributed "nccl" = RCCL over xGMI): every rank runs the same
sequence on its contiguous slice of the batch with GLOBAL loss normalisers and GLOBAL dropout indices, the
flat gradient buffer is summed with all-reduce in two buckets (decoder bucket overlaps the encoder's
backward), and the weight-decay term, clipping and Adam run identically on every rank on the reduced
buffer -- so N ranks reproduce the 1-rank step on the same global batch.
"""
import ctypes

import numpy as np
import torch

from .. import _lib, ops
from ..dp import GradBuckets, reduce_sum, capture
from .model import WS_LOSS, WS_NORMS


class FusedTrainer:
    def __init__(self, model, lambdas1, lambdas2, lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, clip=5.0,
                 process_group=None, use_graph=False, seed=23):
        self.model = model
        self.lambdas1, self.lambdas2 = [float(x) for x in lambdas1], [float(x) for x in lambdas2]
        assert len(self.lambdas1) == model.num_layers and len(self.lambdas2) == model.num_layers
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        self.use_graph = use_graph       # the data-parallel step is captured too: RCCL collectives are graph nodes
        dev = model.dev
        self.m = torch.zeros_like(model.flat)
        self.v = torch.zeros_like(model.flat)
        self.scal = torch.zeros(192, device=dev, dtype=torch.float32)   # adt_clip_adam scalars + partial-sum slots
        self.base_seed = seed
        self.nstep = 0
        self._B = -1
        self._bstate = {}
        nl, H = model.num_layers, model.num_heads
        w = [1.0, 1.0] + self.lambdas1 + [self.lambdas2[nl - 1] if H > 1 else 0.0] * nl   # stale-index NLL weight
        self._loss_w = torch.tensor(w, device=dev, dtype=torch.float32)
        # bucket boundary for the overlapped all-reduce: decoder parameters start at this flat offset
        self._dec_off = model.offsets[4 + 14 * nl]
        self._buckets = GradBuckets(model.flat_grad, self._dec_off, process_group)

    # ------------------------------------------------------------------------------------------
    def _alloc(self, B):
        """Per-batch-size state (id staging ring, device id buffer, captured graph), cached: the trailing partial batch of an
        epoch and the regular batch each keep theirs, so switching B neither re-captures nor invalidates a live graph."""
        m = self.model
        st = self._bstate.get(B)
        if st is None:
            T = B * m.maxlen
            n_int = 4 * T + 4   # seq, dec, pos, neg, 3 normalisers (float bits), pad
            st = {"T": T, "ring": [torch.empty(n_int, dtype=torch.int32).pin_memory() for _ in range(3)], "ring_ev": [None] * 3,
                  "devbuf": torch.empty(n_int, device=m.dev, dtype=torch.int32), "graph": None}
            self._bstate[B] = st
            m.workspace(B)
        # Pinned staging ring: the H2D copy of step n is asynchronous, so the host may only refill a buffer once the copy that
        # last read it has executed (its event).  Three buffers keep the host two steps ahead of the device without waiting.
        self._B, self._T = B, st["T"]
        self._ring, self._ring_ev = st["ring"], st["ring_ev"]
        self._ring_i = 0
        self._host = self._ring[0]
        self._devbuf = st["devbuf"]
        T = self._T
        self._ids = [self._devbuf[i * T:(i + 1) * T].view(B, m.maxlen) for i in range(4)]
        self._norms_dev = self._devbuf[4 * T:4 * T + 4].view(torch.float32)     # 3 normalisers + a zero pad word
        self._st = st

    def _launch(self, B, b_offset):
        """Everything after the H2D copy; capturable."""
        m = self.model
        seq, dec, pos, neg = self._ids
        # one launch: seed += 0x9E3779B1 (a fresh dropout stream every step, on the device), the normalisers into the workspace (a kernel, not
        # tensor.copy_: inside a captured step that would be a memcpy NODE, see DESIGN.md on captured memset nodes), zero_grad, loss slots,
        # ||E||^2 partials for the weight-decay term
        m.run_step_begin(B, self._norms_dev, self.scal)
        m.run_forward(seq, dec, pos, neg, B, True, b_offset)
        m.run_loss_seed(pos, B, self.lambdas1, self.lambdas2, zero_loss=False)
        if not self._buckets.active:
            m.run_backward(seq, dec, pos, neg, B, True, b_offset, phase=0)
        else:
            # two buckets: the decoder bucket's all-reduce (RCCL, its own stream) overlaps the encoder's backward.
            # NOTE: phase 1 also scatters the decoder-input embedding rows, which live in the encoder bucket
            # (item/pos tables at flat offset 0) -- that bucket is reduced after phase 2, so nothing is lost.
            m.run_backward(seq, dec, pos, neg, B, True, b_offset, phase=1)
            self._buckets.tail_ready()
            m.run_backward(seq, dec, pos, neg, B, True, b_offset, phase=2)
            self._buckets.finish()
        ops.clip_adam_pre(m.flat, m.flat_grad, self.m, self.v, (m.item_num + 1) * m.hidden_units, self.wd, self.clip, self.lr,
                          self.betas[0], self.betas[1], self.eps, self.scal)

    def _fill_host(self, seq, dec, pos, neg, norms):
        m = self.model
        seq = np.asarray(seq)
        B = seq.shape[0]
        if B != self._B:
            self._alloc(B)
            if self.nstep == 0:
                m.set_seed(self.base_seed * 1000003 + 12345)
        T = self._T
        self._ring_i = (self._ring_i + 1) % len(self._ring)
        if self._ring_ev[self._ring_i] is not None:
            self._ring_ev[self._ring_i].synchronize()
        self._host = self._ring[self._ring_i]
        hn = self._host.numpy()
        hn[0:T] = seq.reshape(-1)
        hn[T:2 * T] = np.asarray(dec).reshape(-1)
        hn[2 * T:3 * T] = np.asarray(pos).reshape(-1)
        hn[3 * T:4 * T] = np.asarray(neg).reshape(-1)
        if norms is None:
            n_bce = float(np.count_nonzero(hn[2 * T:3 * T]))
            if self.world > 1:   # global normalisers: sum of the per-rank counts
                t = torch.tensor([n_bce], device=m.dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, group=self.pg)
                n_bce = float(t)
            norms = (n_bce, float(self.world * T * m.hidden_units), float(self.world * T * m.num_heads))
        hn[4 * T:4 * T + 3] = np.array(norms, dtype=np.float32).view(np.int32)
        hn[4 * T + 3] = 0
        return B

    def stage(self, batch, norms=None):
        """Upload one id batch (+ normalisers) to HBM ahead of time; returns the device buffer for step_staged."""
        self._fill_host(*batch, norms)
        buf = torch.empty_like(self._devbuf)
        buf.copy_(self._host)
        torch.cuda.synchronize()
        self._ring_ev[self._ring_i] = None
        return buf

    def step_staged(self, buf, b_offset=0):
        """One optimisation step on an id batch already resident in HBM (see stage())."""
        self._devbuf.copy_(buf, non_blocking=True)
        self._run(self._B, b_offset)

    def step(self, seq, dec, pos, neg, norms=None, b_offset=0):
        """One optimisation step on numpy/torch int arrays (B_local, L).  `norms` = (n_bce, n_mse, n_nll) of
        the GLOBAL batch (defaults to this batch's own counts, all-reduced over the process group).  Returns
        nothing; see `loss()`."""
        B = self._fill_host(seq, dec, pos, neg, norms)
        self._devbuf.copy_(self._host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ring_ev[self._ring_i] = ev
        self._run(B, b_offset)

    def _run(self, B, b_offset):
        self.nstep += 1
        if not self.use_graph:
            self._launch(B, b_offset)
            return
        st = self._st
        if st["graph"] is None or st["b_offset"] != b_offset:
            # warm up once eagerly (hipFuncSetAttribute etc. are not capturable), then capture
            self._launch(B, b_offset)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture(g):
                self._launch(B, b_offset)
            st["graph"], st["b_offset"] = g, b_offset
            return
        st["graph"].replay()

    def loss(self):
        """Device scalar: the loss of the last step as the reference prints it (sasrec/main.py:174)."""
        m = self.model
        n = 2 + 2 * m.num_layers
        slots = m.ws_view(self._B, WS_LOSS, 0, 64 * n).view(n, 64).sum(1)   # 64 sub-slots per loss term
        if self.world > 1:          # every rank holds its shard's partial sums (already over the global normalisers)
            slots = reduce_sum(slots.clone(), self.pg)
        return (slots * self._loss_w).sum() + self.scal[3]

    def grad_norm(self):
        return self.scal[1].sqrt()
