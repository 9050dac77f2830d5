"""Deterministic item-table / positional-table gradient (adt_item_sort / adt_item_segsum / adt_posemb_sum; sasrec/model.py:34-41, :53-59,
:72-76 reversed) against a numpy restatement (np.add.at over the same entries) and against itself: two runs give the same bits.
Tolerance: fp32 sums in a different order than numpy's -- 2e-5 of the largest magnitude."""
import numpy as np
import pytest
import torch

from oracle import rng

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def seed_t(seed):
    return torch.from_numpy(np.array([seed], dtype=np.uint32).view(np.int32)).to(DEV)


def make_ids(r, B, L, V, mode):
    if mode == "zipf":
        pop = 1.0 / np.arange(1, V + 1)
        pop /= pop.sum()
        ids = (r.choice(V, size=(B, L), p=pop) + 1).astype(np.int32)
    elif mode == "one":            # every token the same item: one segment over every workgroup
        ids = np.full((B, L), 3, np.int32)
    elif mode == "two":            # two items, long chains with one break
        ids = np.where(r.rand(B, L) < 0.5, 2, V).astype(np.int32)
    else:
        ids = r.randint(1, V + 1, size=(B, L)).astype(np.int32)
    for b in range(1, B):
        ids[b, : r.randint(0, L)] = 0
    return ids


def reference(ids_list, rows, coef, kind, site, p, seed, row_offset, scale, V1, mask):
    dE = np.zeros((V1, 64), np.float64)
    for s, ids in enumerate(ids_list):
        if not (mask >> s) & 1:
            continue
        flat = ids.reshape(-1)
        T = flat.size
        if kind[s] == 1:
            val = rows[s].astype(np.float64) * coef[s].astype(np.float64)[:, None]
        else:
            idx = (np.arange(T, dtype=np.int64)[:, None] + row_offset) * 64 + np.arange(64)[None, :]
            keep = rng.keep_mask(seed, site[s], idx, p) if p > 0 else np.ones((T, 64), bool)
            val = rows[s].astype(np.float64) * scale * keep / (1.0 - rng.drop_prob(p))
        nz = flat != 0
        np.add.at(dE, flat[nz], val[nz])
    return dE


@pytest.mark.parametrize("B,L,V,mode,p", [(256, 200, 3416, "zipf", 0.5), (8, 52, 40, "uniform", 0.2), (64, 200, 3416, "one", 0.0),
                                          (33, 100, 7, "two", 0.5), (3, 16, 15000, "uniform", 0.0), (256, 200, 15999, "zipf", 0.5)])
def test_item_segsum_vs_numpy_and_deterministic(B, L, V, mode, p):
    from adt_amd import ops
    r = np.random.RandomState(B * 7 + L)
    T, V1 = B * L, V + 1
    ids = [make_ids(r, B, L, V, mode) for _ in range(4)]
    rows = [r.randn(T, 64).astype(np.float32) for _ in range(4)]
    rows[3] = rows[2]                                  # pos / neg share log_feats
    coef = [None, None, r.randn(T).astype(np.float32), r.randn(T).astype(np.float32)]
    kind, site, seed, ro, scale = [0, 0, 1, 1], [1, 2, 0, 0], 1234567, 5 * L, 8.0
    d_ids = [T_(a.reshape(-1)) for a in ids]
    d_rows = [T_(a) for a in rows]
    d_coef = [None, None, T_(coef[2]), T_(coef[3])]
    sd = seed_t(seed)
    outs = []
    for rep in range(2):
        work = ops.item_sort(d_ids, V1, d_rows, d_coef, kind, ro)
        dE = torch.zeros(V1, 64, device=DEV)
        ops.item_segsum(work, 4, T, V1, 0b1110, site, p, sd, scale, dE)      # decoder ids + pos + neg: plain stores ...
        ops.item_segsum(work, 4, T, V1, 0b0001, site, p, sd, scale, dE, accumulate=True)      # ... the encoder ids on top
        if rep == 0:
            one = torch.zeros(V1, 64, device=DEV)
            ops.item_segsum(work, 4, T, V1, 0b1111, site, p, sd, scale, one)      # everything in one pass
            torch.cuda.synchronize()
            one = one.cpu().numpy()
        torch.cuda.synchronize()
        outs.append(dE.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]), "two runs of the sorted sums differ"
    assert np.abs(one - outs[0]).max() <= 2e-5 * max(np.abs(one).max(), 1e-6)
    want = reference(ids, rows, coef, kind, site, p, seed, ro, scale, V1, 0b1111)
    err = np.abs(outs[0] - want).max() / max(np.abs(want).max(), 1e-6)
    assert err < 2e-5, err
    assert np.all(outs[0][0] == 0.0), "padding row touched"
    # the sort itself: a stable permutation of the non-zero entries
    N = 4 * T
    base = work[ops._lib.load().adt_item_sort_work_ints(4, T, V1) * 0:].cpu().numpy()      # whole buffer
    up = lambda x: (x + 63) // 64 * 64
    o_base = up(256 * V1)
    o_perm = o_base + up(V1 + 1 + 256)
    o_item = o_perm + up(N)
    n_ent = int(base[o_base + V1])
    allids = np.concatenate([a.reshape(-1) for a in ids])
    assert n_ent == int(np.count_nonzero(allids))
    perm, pitem = base[o_perm:o_perm + n_ent], base[o_item:o_item + n_ent]
    assert np.array_equal(allids[perm], pitem)
    order = np.lexsort((perm, pitem))
    assert np.array_equal(order, np.arange(n_ent)), "not sorted by (item, entry)"


@pytest.mark.parametrize("B,L,p", [(256, 200, 0.5), (5, 52, 0.0), (31, 16, 0.3)])
def test_posemb_sum_vs_numpy_and_deterministic(B, L, p):
    from adt_amd import ops
    r = np.random.RandomState(B + L)
    T = B * L
    ids = [make_ids(r, B, L, 50, "uniform") for _ in range(2)]
    dX = [r.randn(T, 64).astype(np.float32) for _ in range(2)]
    site, seed, ro = [1, 2], 99, 3 * L
    want = np.zeros((L, 64), np.float64)
    for s in range(2):
        idx = (np.arange(T, dtype=np.int64)[:, None] + ro) * 64 + np.arange(64)[None, :]
        keep = rng.keep_mask(seed, site[s], idx, p) if p > 0 else np.ones((T, 64), bool)
        val = dX[s].astype(np.float64) * keep / (1.0 - rng.drop_prob(p)) * (ids[s].reshape(-1) != 0)[:, None]
        want += val.reshape(B, L, 64).sum(0)
    outs = []
    for rep in range(2):
        dP = torch.zeros(L, 64, device=DEV)
        ops.posemb_sum([T_(a.reshape(-1)) for a in ids], [T_(a) for a in dX], site, B, L, p, seed_t(seed), ro, dP)
        torch.cuda.synchronize()
        outs.append(dP.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0] - want).max() / max(np.abs(want).max(), 1e-6) < 2e-5
