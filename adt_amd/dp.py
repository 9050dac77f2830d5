"""Data-parallel step logic shared by the HIP trainer (adt_amd/sasrec/trainer.py, RCCL over xGMI) and the CPU
gloo tests.  The path shards by user sequence and has exactly one exchange step: a sum all-reduce of the flat
fp32 gradient buffer.  Exactness rules that make N ranks reproduce the 1-rank step on the same
global batch:
  1. loss normalisers are GLOBAL (count of pos != 0, B*L*d, B*L*H of the whole batch), never per-rank means;
  2. dropout masks are indexed by GLOBAL sequence index (b_offset), so sharding does not change them;
  3. the weight-decay term wd*||E||_F, gradient clipping and Adam run AFTER the all-reduce, identically on every
     rank, on the reduced buffer.
"""
import numpy as np


def shard_bounds(B, rank, world):
    """Contiguous rows [lo, hi) of a global batch of B sequences owned by `rank`."""
    per = (B + world - 1) // world
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def global_norms(pos, hidden, num_heads):
    """(n_bce, n_mse, n_nll) of the GLOBAL batch: sasrec/main.py:150-153 (BCE over pos != 0), :158 (MSE mean over all
    B*L*d elements), :169 (NLL mean over B*L*H)."""
    pos = np.asarray(pos)
    return float(np.count_nonzero(pos)), float(pos.size * hidden), float(pos.size * num_heads)


def allreduce_buckets(flat, boundaries, group=None, async_op=True):
    """Sum-all-reduce `flat` in the buckets delimited by `boundaries` (offsets, ascending, within (0, len)).
    Returns the list of work handles (empty when torch.distributed is not initialised: single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return []
    edges = [0] + [int(b) for b in boundaries] + [flat.numel()]
    handles = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        if hi > lo:
            handles.append(dist.all_reduce(flat[lo:hi], group=group, async_op=async_op))
    return [h for h in handles if h is not None]


class DPStep:
    """Engine-agnostic data-parallel optimisation step.  `engine` provides
         forward_backward(shard_batch, norms, b_offset) -> flat gradient tensor of THIS shard's contribution
                                                           (no weight-decay term), to be summed over ranks
         apply(flat_gradient)                           -> weight-decay term + clip + Adam on the reduced buffer
    """

    def __init__(self, engine, hidden, num_heads, rank=0, world=1, group=None, bucket_boundaries=()):
        self.engine, self.hidden, self.num_heads = engine, hidden, num_heads
        self.rank, self.world, self.group, self.boundaries = rank, world, group, list(bucket_boundaries)

    def step(self, seq, dec, pos, neg):
        B = len(seq)
        lo, hi = shard_bounds(B, self.rank, self.world)
        norms = global_norms(pos, self.hidden, self.num_heads)
        flat = self.engine.forward_backward((seq[lo:hi], dec[lo:hi], pos[lo:hi], neg[lo:hi]), norms, lo)
        for h in allreduce_buckets(flat, self.boundaries, self.group, async_op=True):
            h.wait()
        self.engine.apply(flat)
        return flat
