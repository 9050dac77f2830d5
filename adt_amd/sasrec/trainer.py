"""Fused training step for SASRecADT: the loop body of the reference's sasrec/main.py:143-173 (forward, loss
assembly, backward, weight-decay term, clip_grad_norm_, Adam) as ONE device-side launch sequence, optionally
replayed from a HIP graph.  Host work per step = one packed write of the id batch into a slot of a pinned ring + ONE graph launch: the step's first kernel fetches
the slot itself (no copy-engine transfer, no event).

Data-parallel (one process per GPU, torch.distributed "nccl" = RCCL over xGMI): every rank runs the same
sequence on its contiguous slice of the batch with GLOBAL loss normalisers and GLOBAL dropout indices, the
flat gradient buffer is summed with all-reduce in two buckets (decoder bucket overlaps the encoder's
backward), and the weight-decay term, clipping and Adam run identically on every rank on the reduced
buffer -- so N ranks reproduce the 1-rank step on the same global batch.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _hostlib, _lib, ops
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
        self._dp_phases = 2 if os.environ.get("ADT_DP_PHASES", "1") == "2" else 1
        # (with the sorted table gradients, ADT_ITEM_SORT, the side stream carries the id sort: the logits kernel stays on the caller's stream there)
        self._bce_side = os.environ.get("ADT_BCE_SIDE", "1") != "0" and os.environ.get("ADT_ITEM_SORT", "0") == "0"

    # ------------------------------------------------------------------------------------------
    NSLOTS = 4          # pinned id ring: the producer may run up to three batches ahead of the step the GPU is executing
    WAIT_US = 30 * 1000 * 1000

    def _alloc(self, B):
        """Per-batch-size state (pinned id ring, device id buffer, ring counters, captured graphs), cached: the trailing partial batch of
        an epoch and the regular batch each keep theirs, so switching B neither re-captures nor invalidates a live graph."""
        m = self.model
        st = self._bstate.get(B)
        if st is None:
            T = B * m.maxlen
            n_int = 4 * T + 4   # seq, dec, pos, neg, 3 normalisers (float bits), pad
            ring = torch.empty(self.NSLOTS * n_int, dtype=torch.int32).pin_memory()
            consumed = torch.zeros(16, dtype=torch.int32).pin_memory()
            produced = torch.zeros(16, dtype=torch.int32).pin_memory()      # [0]: batches the producer has completely written (publish())
            st = {"T": T, "n_int": n_int, "ring": ring, "ring_np": ring.numpy(), "consumed": consumed, "consumed_np": consumed.numpy().view(np.uint32),
                  "produced": produced, "produced_np": produced.numpy().view(np.uint32),
                  "state": torch.zeros(8, device=m.dev, dtype=torch.int32), "nsub": 0,
                  "staging": torch.empty(n_int, device=m.dev, dtype=torch.int32),      # the next batch, prefetched beside this step's loss assembly
                  "devbuf": torch.empty(n_int, device=m.dev, dtype=torch.int32), "graphs": {}, "dev_rings": []}
            self._bstate[B] = st
            m.workspace(B)
        return st

    def _bind(self, B):
        """Make batch size B the one the next launch runs on."""
        if B != self._B:
            m = self.model
            st = self._alloc(B)
            self._B, self._T = B, st["T"]
            self._devbuf = st["devbuf"]
            T = self._T
            self._ids = [self._devbuf[i * T:(i + 1) * T].view(B, m.maxlen) for i in range(4)]
            self._norms_dev = self._devbuf[4 * T:4 * T + 4].view(torch.float32)     # 3 normalisers + a zero pad word
            self._st = st
            if self.nstep == 0:
                m.set_seed(self.base_seed * 1000003 + 12345)
        return self._st

    def _launch(self, B, b_offset, src):
        """The whole step; capturable.  src: None -- the ids (and normalisers) are already in the device id buffer; otherwise the id ring
        (ring tensor, slots, state, consumed or None) whose current slot the step's first kernel fetches."""
        m = self.model
        seq, dec, pos, neg = self._ids
        # one launch: [the id batch from the ring slot,] seed += 0x9E3779B1 (a fresh dropout stream every step, on the device), the normalisers into the
        # workspace (a kernel, not tensor.copy_: inside a captured step that would be a memcpy NODE, see DESIGN.md on captured memset nodes), zero_grad,
        # loss slots, ||E||^2 partials for the weight-decay term
        prefetch = None
        if src is None:
            m.run_step_begin(B, self._norms_dev, self.scal)
        elif len(src) > 4:      # pinned host ring: the next batch is prefetched beside this step's loss assembly when its producer is ahead
            ring, nslots, state, consumed, staging, produced = src
            m.run_step_begin_ring_staged(B, ring, self._st["n_int"], nslots, self._devbuf, state, consumed, staging, produced, self.scal)
            prefetch = (ring, self._st["n_int"], nslots, state, consumed, staging)
        else:
            m.run_step_begin_ring(B, src[0], self._st["n_int"], src[1], self._devbuf, src[2], src[3], self.scal)
        # forward + loss assembly (run_step_begin* above packed the weight images and zeroed the loss slots ...)
        # bce: logits + BCE seed left to the backward (True), or launched beside the loss pass on the side stream and joined by the backward ("fwd")
        bce = m.run_forward_loss(seq, dec, pos, neg, B, self.lambdas1, self.lambdas2, b_offset, prefetch=prefetch, bce_side=self._bce_side)
        if not self._buckets.active:
            # ... and zeroed the parameter-gradient replicas ; the last fold of the replicas happens inside the optimizer's first kernel
            m.run_backward(seq, dec, pos, neg, B, True, b_offset, phase=0, prezeroed=True, defer_fold=True, bce=bce)
            m.run_fold_clip_adam(B, self.m, self.v, self.wd, self.clip, self.lr, self.betas[0], self.betas[1], self.eps, self.scal)
            return
        elif self._dp_phases == 1:
            # data parallel, ONE exchange: the single-GPU backward (one phase, side stream, stored partial sums), the sums into the flat
            # gradient, one all-reduce of its 1.46 MB, then weight-decay term + clip + Adam on every rank.  The two-bucket form below hides
            # the decoder's 0.3 MB behind the encoder backward but runs the atomics-and-replicas backward on one stream: 0.663 against
            # 0.63 ms per step on one rank (ADT_DP_PHASES=2 selects it).
            m.run_backward(seq, dec, pos, neg, B, True, b_offset, phase=0, prezeroed=True, defer_fold=True, bce=bce)
            m.run_fold_grads(B, self.scal)
            self._buckets.whole(m.flat_grad)
        else:
            # two buckets: the decoder bucket's all-reduce (RCCL, its own stream) overlaps the encoder's backward.
            # NOTE: phase 1 also scatters the decoder-input embedding rows, which live in the encoder bucket
            # (item/pos tables at flat offset 0) -- that bucket is reduced after phase 2, so nothing is lost.
            m.run_backward(seq, dec, pos, neg, B, True, b_offset, phase=1, prezeroed=True, bce=bce)
            self._buckets.tail_ready()
            m.run_backward(seq, dec, pos, neg, B, True, b_offset, phase=2, prezeroed=True, bce=bce)      # the gradient buffer's table rows are still zero
            self._buckets.finish()
        ops.clip_adam_pre(m.flat, m.flat_grad, self.m, self.v, (m.item_num + 1) * m.hidden_units, self.wd, self.clip, self.lr,
                          self.betas[0], self.betas[1], self.eps, self.scal)

    # ---- the pinned id ring ---------------------------------------------------------------------------------------------------
    # Host -> device hand-over of a batch (sasrec/main.py:144-145) without a copy-engine transfer: the producer writes the packed batch into
    # slot (k % NSLOTS) of a pinned ring, the step's first kernel reads the slot over PCIe and stores the count of fetched batches into a
    # pinned word (adt_sasrec_step_begin_ring).  A slot written for step k may be refilled once that count exceeds k.  Per step the host
    # issues ONE call (the graph launch); a hipMemcpyAsync + event pair per step put two cross-queue waits (copy engine <-> compute) on the
    # GPU's critical path and three runtime calls on the host's.
    def slot(self, B, k=None):
        """numpy views (seq, dec, pos, neg: (B, L) int32) + the flat int32 view of ring slot k % NSLOTS for the k-th ring step of batch size
        B (default: the next one to be submitted), after waiting until the GPU has fetched the batch that last occupied it.  A producer
        thread may fill slots up to NSLOTS - 1 steps ahead of commit() (this method does not touch the trainer's launch state)."""
        st = self._alloc(B)
        k = st["nsub"] if k is None else k
        need = k - self.NSLOTS + 1
        if need > 0 and _hostlib.load().adt_host_wait_ge(st["consumed_np"].ctypes.data, need & 0xFFFFFFFF, self.WAIT_US) != 0:
            raise RuntimeError("FusedTrainer: the GPU has not fetched ring step %d after %d s" % (need - 1, self.WAIT_US // 1000000))
        n, T, L = st["n_int"], st["T"], self.model.maxlen
        flat = st["ring_np"][(k % self.NSLOTS) * n:(k % self.NSLOTS + 1) * n]
        return [flat[i * T:(i + 1) * T].reshape(B, L) for i in range(4)], flat

    def publish(self, B, k, norms):
        """Producer side: ring step k of batch size B is completely written (ids in its slot()); stores its normalisers (n_bce, n_mse, n_nll of
        the GLOBAL batch) and announces it to the GPU, which may then prefetch it beside the step in front of it.  Steps are published in order."""
        st = self._alloc(B)
        flat = st["ring_np"][(k % self.NSLOTS) * st["n_int"]:(k % self.NSLOTS + 1) * st["n_int"]]
        flat[4 * st["T"]:4 * st["T"] + 3] = np.asarray(norms, np.float32).view(np.int32)
        flat[4 * st["T"] + 3] = 0
        _hostlib.load().adt_host_store_release(st["produced_np"].ctypes.data, (k + 1) & 0xFFFFFFFF)

    def commit(self, B, norms=None, b_offset=0):
        """Submit the step whose batch the caller has written into slot() (ids in place); norms = (n_bce, n_mse, n_nll) of the GLOBAL batch,
        or None when the producer has already publish()ed this step."""
        st = self._bind(B)
        if norms is not None:
            self.publish(B, st["nsub"], norms)
        self._submit(st, B, b_offset)

    def _submit(self, st, B, b_offset):
        st["nsub"] += 1
        self._run(B, b_offset, ("host", (st["ring"], self.NSLOTS, st["state"], st["consumed"], st["staging"], st["produced"])))

    def _norms(self, pos_flat, T):
        m = self.model
        n_bce = float(np.count_nonzero(pos_flat))
        if self.world > 1:   # global normalisers: sum of the per-rank counts
            t = torch.tensor([n_bce], device=m.dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, group=self.pg)
            n_bce = float(t)
        return (n_bce, float(self.world * T * m.hidden_units), float(self.world * T * m.num_heads))

    def step(self, seq, dec, pos, neg, norms=None, b_offset=0):
        """One optimisation step on numpy/torch int arrays (B_local, L).  `norms` = (n_bce, n_mse, n_nll) of
        the GLOBAL batch (defaults to this batch's own counts, all-reduced over the process group).  Returns
        nothing; see `loss()`."""
        arrs = [np.ascontiguousarray(a.cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.int32) for a in (seq, dec, pos, neg)]
        B = arrs[0].shape[0]
        _, flat = self.slot(B)
        st = self._bind(B)
        if norms is None:
            norms = self._norms(arrs[2], st["T"])
        _hostlib.load().adt_host_pack_batch(flat.ctypes.data, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                            st["T"], norms[0], norms[1], norms[2])
        _hostlib.load().adt_host_store_release(st["produced_np"].ctypes.data, (st["nsub"] + 1) & 0xFFFFFFFF)
        self._submit(st, B, b_offset)

    # ---- batches resident in HBM ------------------------------------------------------------------------------------------------
    def stage(self, batch, norms=None):
        """Upload one id batch (+ normalisers) to HBM ahead of time; returns the device buffer for step_staged."""
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in batch]
        st = self._bind(arrs[0].shape[0])
        T = st["T"]
        if norms is None:
            norms = self._norms(arrs[2], T)
        host = np.empty(st["n_int"], np.int32)
        _hostlib.load().adt_host_pack_batch(host.ctypes.data, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                            T, norms[0], norms[1], norms[2])
        return torch.from_numpy(host).to(self.model.dev)

    def stage_ring(self, batches, norms=None):
        """A ring of len(batches) staged batches in HBM: step_staged(handle) then runs them round-robin with ONE host call per step (the
        graph launch) -- the id fetch is the same first kernel as in step(), reading HBM instead of pinned host memory."""
        bufs = [self.stage(b, None if norms is None else norms[i]) for i, b in enumerate(batches)]
        st = self._st
        ring = torch.cat(bufs)
        h = ("dev%d" % len(st["dev_rings"]), (ring, len(bufs), torch.zeros(8, device=self.model.dev, dtype=torch.int32), None))
        st["dev_rings"].append(h)
        torch.cuda.synchronize()
        return h

    def step_staged(self, buf, b_offset=0):
        """One optimisation step on an id batch already resident in HBM: a buffer of stage() (copied device-to-device into the id buffer) or
        the next slot of a stage_ring() handle."""
        if isinstance(buf, tuple):
            return self._run(self._B, b_offset, buf)
        self._devbuf.copy_(buf, non_blocking=True)
        self._run(self._B, b_offset, None)

    def _run(self, B, b_offset, src):
        self.nstep += 1
        key, ring = (None, None) if src is None else src
        if not self.use_graph:
            self._launch(B, b_offset, ring)
            return
        st = self._st
        g = st["graphs"].get((key, b_offset))
        if g is None:
            # warm up once eagerly (hipFuncSetAttribute etc. are not capturable) -- this IS the step -- then capture for the next ones
            self._launch(B, b_offset, ring)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture(g):
                self._launch(B, b_offset, ring)
            st["graphs"][(key, b_offset)] = g
            return
        g.replay()

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


class RingFeeder:
    """Producer side of the trainer's pinned id ring: a background thread samples the batches of an epoch with the native sampler
    (libadt_host.so) STRAIGHT into ring slots, two steps ahead of the step being submitted, so the training thread's work per step is one
    commit() = one graph launch.  A data-parallel rank samples only its own rows [lo, hi) of every global batch (the row streams of the
    native sampler depend on the global row index, so the union over ranks is the batch a single process draws) and takes the global BCE
    normaliser from the history lengths.  Replaces DataLoader(WarpDataset, num_workers=4) + the per-batch ndarray -> LongTensor -> device
    conversions of sasrec/main.py:88,141-145."""

    def __init__(self, trainer, warp, rank=0, world=1, depth=2):
        assert warp._native is not None, "RingFeeder needs libadt_host.so (python -m adt_amd.csrc.build)"
        assert 1 <= depth <= trainer.NSLOTS - 2
        self.tr, self.warp, self.rank, self.world, self.depth = trainer, warp, rank, world, depth

    def epoch(self, batch_size, rng):
        """Runs one epoch; yields the number of sequences of each global batch after its step has been submitted."""
        import queue
        import threading
        from ..dp import shard_bounds, skip_batch
        tr, warp, m = self.tr, self.warp, self.tr.model
        plan = []                    # drawn up front on the calling thread: the random stream does not depend on thread timing
        for users in warp.epoch_users(batch_size, rng):
            plan.append((users, warp.next_seed(rng)))
        q = queue.Queue(maxsize=self.depth)
        err = []

        def produce():
            try:
                nxt = {}             # next ring step index per local batch size
                for users, seed in plan:
                    n = len(users)
                    if skip_batch(n, self.world):
                        continue
                    lo, hi = shard_bounds(n, self.rank, self.world)
                    B = hi - lo
                    k = nxt.get(B, tr._alloc(B)["nsub"])
                    views, _ = tr.slot(B, k)
                    warp.sample_rows_into(users[lo:hi], seed, views, b0=lo)
                    nxt[B] = k + 1
                    L = m.maxlen
                    tr.publish(B, k, (float(warp.count_targets(users)), float(n * L * m.hidden_units), float(n * L * m.num_heads)))
                    q.put((B, lo, n, None))
            except BaseException as e:      # surfaces on the training thread
                err.append(e)
            q.put(None)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is None:
                break
            B, lo, n, norms = item
            tr.commit(B, norms, b_offset=lo)
            yield n
        th.join()
        if err:
            raise err[0]
