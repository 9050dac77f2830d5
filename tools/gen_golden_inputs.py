"""Seeded synthetic id batches shared by tools/gen_golden.py (here) and the tests (anywhere).  No reference
imports: this module travels to the GPU box."""
import numpy as np


def make_batch(r, B, L, V, max_pad_frac=0.9):
    """ids shaped like WarpDataset.sample_data output (sasrec/utils.py:288-307): right-aligned history,
    dec[k] = seq[k-1], pos = next item, neg = random item where pos != 0."""
    seq = np.zeros((B, L), np.int32)
    dec = np.zeros((B, L), np.int32)
    pos = np.zeros((B, L), np.int32)
    neg = np.zeros((B, L), np.int32)
    for b in range(B):
        n = L if b == 0 else max(1, int(L * (1.0 - max_pad_frac * r.rand())))
        items = r.randint(1, V + 1, size=n + 1)
        seq[b, L - n:] = items[:-1]
        pos[b, L - n:] = items[1:]
        neg[b, L - n:] = r.randint(1, V + 1, size=n)
        dec[b, 1:] = seq[b, :-1]
    return seq, dec, pos, neg


def sample_idx(n, k=256):
    return (np.arange(min(k, n), dtype=np.int64) * 7919) % n
