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


def compact(out, keep=(), k=1024, thresh=8192):
    """Fixtures at the BASELINE configurations' own shapes would be hundreds of MB: every float array larger than `thresh` elements
    becomes its Frobenius norm (`key@norm`) plus `k` strided samples (`key@sample`, indices sample_idx(size, k))."""
    res = {}
    for key, v in out.items():
        a = np.asarray(v)
        if key in keep or a.dtype.kind != "f" or a.size <= thresh:
            res[key] = v
            continue
        t = a.reshape(-1)
        res[key + "@norm"] = np.float64(np.sqrt((t.astype(np.float64) ** 2).sum()))
        res[key + "@sample"] = t[sample_idx(t.size, k)].copy()
    return res


def golden_err(got, g, key, k=1024):
    """Relative error of `got` against fixture entry `key`, stored whole or compacted (see compact()): max |diff| over the stored
    entries relative to the reference's magnitude, and -- for compacted entries -- also the relative error of the norm."""
    got = np.asarray(got, np.float64)
    if key in g.files:
        want = np.asarray(g[key], np.float64)
        assert got.shape == want.shape, (key, got.shape, want.shape)
        return np.abs(got - want).max() / max(np.abs(want).max(), 1e-6)
    want = np.asarray(g[key + "@sample"], np.float64)
    norm = float(g[key + "@norm"])
    t = got.reshape(-1)
    s = t[sample_idx(t.size, k)]
    scale = max(np.abs(want).max(), norm / np.sqrt(max(t.size, 1)), 1e-9)
    e_s = np.abs(s - want).max() / scale
    e_n = abs(np.sqrt((t ** 2).sum()) - norm) / max(norm, 1e-9)
    return max(e_s, e_n)


def seeded_params(named_shapes, seed):
    """Deterministic fp32 fill of a {name: shape} table, identical wherever it is called (the golden generator fills the REFERENCE
    model with it, the tests fill ours): tensors in sorted-name order from one RandomState; LayerNorm scales ~1, everything else
    N(0, 0.06) (biases included, so that their gradients are exercised)."""
    r = np.random.RandomState(seed)
    out = {}
    for name in sorted(named_shapes):
        shape = tuple(named_shapes[name])
        v = 0.06 * r.standard_normal(shape)
        low = name.lower()
        if ("layer_norm" in low or "layernorm" in low) and name.endswith("weight"):
            v = 1.0 + v
        out[name] = v.astype(np.float32)
    return out
