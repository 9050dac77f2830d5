"""Data-parallel layer shared by every fused trainer (sasrec/trainer.py, sasrec/model_wide.py, bert4rec/trainer.py,
stosa/trainer.py) and the three entry points.  One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on
ROCm); the same code runs on "gloo" in the CPU / shared-GPU tests.

The path shards by user sequence and has exactly one exchange step: a sum all-reduce of the flat fp32 gradient buffer
(SURVEY.md 8e).  Exactness rules that make N ranks reproduce the 1-rank step on the same global batch:
  1. loss normalisers are GLOBAL (count of pos != 0, B*L*d, B*L*H of the whole batch), never per-rank means;
  2. dropout masks are indexed by GLOBAL sequence index (b_offset), so sharding does not change them;
  3. the weight-decay term wd*||E||_F, gradient clipping and Adam run AFTER the all-reduce, identically on every
     rank, on the reduced buffer.

xGMI is point-to-point (7 links per GPU), so a ring all-reduce of a small buffer is latency bound: the gradient goes out in
TWO buckets, not many.  The flat layout puts embeddings and encoder first and the decoder (and, for BERT, the output head) last;
the backward pass finishes the decoder's parameters first, so the tail bucket's all-reduce runs on RCCL's stream while the
encoder's backward and the embedding scatter still execute (`GradBuckets.tail_ready()`), and the head bucket goes out at the
end (`GradBuckets.finish()`).  Both collectives are captured into the step's HIP graph together with the kernels.
"""
import os

import numpy as np


def init_from_env(backend="nccl"):
    """Process-group set-up of an entry point launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (process_group or None, rank, world, local_rank); with ADT_FORCE_DP=1 a single process still gets a 1-rank group
    so that the data-parallel code path (bucketed all-reduce on RCCL's stream) is the one that runs."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("ADT_FORCE_DP", "0") == "1"
    if world == 1 and not force:
        return None, 0, 1, local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a pid-derived port only makes sense for the 1-rank group (ADT_FORCE_DP): with world > 1 every rank must agree on the port, so an
    # unset MASTER_PORT (a manual launch without torchrun) falls back to torch's fixed default
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 2000) if world == 1 else "29500")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
        if torch.cuda.device_count() > 0:      # gloo ranks may share a GPU (the one-GPU test box): fold the local rank onto the devices present
            local = local % torch.cuda.device_count()
    return dist.group.WORLD, rank, world, local


def shard_bounds(B, rank, world):
    """Contiguous rows [lo, hi) of a global batch of B sequences owned by `rank`: balanced, the first B % world ranks get one
    more row, so a rank is empty only when B < world (callers skip such a trailing batch: `skip_batch`)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def skip_batch(B, world):
    """A trailing batch with fewer sequences than ranks would leave a rank without rows (zero-sized launches on that rank
    while the others wait in the all-reduce): every rank drops it (identically, so nobody hangs)."""
    return B < world


def global_norms(pos, hidden, num_heads):
    """(n_bce, n_mse, n_nll) of the GLOBAL batch: sasrec/main.py:150-153 (BCE over pos != 0), :158 (MSE mean over all
    B*L*d elements), :169 (NLL mean over B*L*H)."""
    pos = np.asarray(pos)
    return float(np.count_nonzero(pos)), float(pos.size * hidden), float(pos.size * num_heads)


class GradBuckets:
    """The gradient exchange of one step: flat[boundary:n) as soon as the decoder's backward is done, flat[0:boundary) at the
    end.  With no process group (single GPU) both calls are no-ops.  torch.distributed issues each collective on the
    backend's own stream behind an event of the current stream, and `wait()` makes the current stream wait for it: under HIP
    graph capture both become cross-stream edges of the captured graph."""

    def __init__(self, flat_grad, boundary, group=None, n=None):
        n = flat_grad.numel() if n is None else int(n)
        boundary = max(0, min(int(boundary), n))
        self.group = group
        self.tail = flat_grad[boundary:n] if boundary < n else None
        self.head = flat_grad[:boundary] if boundary > 0 else None
        self._h = []

    @property
    def active(self):
        return self.group is not None

    def tail_ready(self):
        if self.group is None or self.tail is None:
            return
        import torch.distributed as dist
        self._h.append(dist.all_reduce(self.tail, group=self.group, async_op=True))

    def whole(self, flat_grad):
        """One sum of the whole flat gradient (the one-phase data-parallel step: nothing to overlap it with, but the backward in front of
        it is the single-GPU one -- side stream, stored partial sums)."""
        if self.group is None:
            return
        import torch.distributed as dist
        h = dist.all_reduce(flat_grad, group=self.group, async_op=True)
        if h is not None:
            h.wait()

    def finish(self):
        if self.group is None:
            return
        import torch.distributed as dist
        if self.head is not None:
            self._h.append(dist.all_reduce(self.head, group=self.group, async_op=True))
        for h in self._h:
            if h is not None:
                h.wait()
        self._h = []


def reduce_sum(t, group=None):
    """Sum a small device tensor (loss partial sums, normaliser counts) over the ranks, in place; no-op without a group."""
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(t, group=group)
    return t


def capture(graph):
    """torch.cuda.graph with capture_error_mode="thread_local".  The default ("global") makes EVERY thread's HIP calls illegal while the
    step is being captured, and ProcessGroupNCCL's watchdog thread polls the events of earlier collectives with hipEventQuery whenever it
    likes: a poll that lands inside the capture raises `operation not permitted when stream is capturing` in the watchdog, which takes
    the process down (seen in tests/test_dp_nccl.py as soon as the steps got short enough for the race to hit).  Thread-local mode only
    polices the capturing thread."""
    import torch
    return torch.cuda.graph(graph, capture_error_mode="thread_local")
