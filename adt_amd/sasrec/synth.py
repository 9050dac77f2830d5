"""Seeded synthetic interaction files in the reference's "<user> <item>" line format (sasrec/utils.py:328-335).

The reference ships neither ml-1m nor ml-20m (.MISSING_LARGE_BLOBS), so the north-star configuration runs on a
generated dataset with ml-1m's public shape: 6,040 users, 3,416 items, history length ~ lognormal clipped to
[20, 2314] (mean ~165), Zipf(1.0) item popularity.  Unlike i.i.d. popularity draws the sequences carry a
learnable first-order structure (each item has a few preferred successors), so that NDCG@10 depends on the
model actually working and is a meaningful parity signal between implementations.
"""
import numpy as np

PRESETS = {
    "ml1m": dict(users=6040, items=3416, len_mu=4.6, len_sigma=0.95, len_min=20, len_max=2314),
    "ml1m-small": dict(users=1200, items=800, len_mu=4.0, len_sigma=0.8, len_min=12, len_max=400),
    "tiny": dict(users=64, items=120, len_mu=3.0, len_sigma=0.6, len_min=5, len_max=80),
    # BASELINE configs[2]: ml-20m's public shape (138,493 users, 26,744 items, mean history ~144).  Histories are cut at
    # 900 drawn items: the models read the last 200 (BERT4Rec's sliding windows a few more).  (len_mu is set so that the mean
    # AFTER dropping repeated items is the target: 146 here, 9.0 and 9.4 for the two Beauty shapes.)
    "ml20m": dict(users=138493, items=26744, len_mu=5.2, len_sigma=0.85, len_min=20, len_max=900, fast=True),
    # BASELINE configs[3]: the shape of the reference's sasrec/data/beauty.txt (40,226 users, 54,542 items, 353,962 actions: mean 8.8)
    "beauty": dict(users=40226, items=54542, len_mu=2.15, len_sigma=0.55, len_min=5, len_max=200, fast=True),
    # BASELINE configs[4]: the shape of stosa/data/Beauty.txt (22,363 users, 12,101 items, 5-core)
    "beauty-stosa": dict(users=22363, items=12101, len_mu=2.2, len_sigma=0.55, len_min=5, len_max=200, fast=True),
}


def _generate_fast(p, seed, follow, n_succ):
    """The same generative model as generate() -- Zipf popularity, a few preferred successors per item, follow the chain with
    probability `follow` -- vectorised over users (one numpy step per position) for the large presets; repeated items are dropped
    afterwards (first occurrence kept) instead of re-drawn, so histories come out slightly shorter than drawn."""
    r = np.random.RandomState(seed)
    U, V = p["users"], p["items"]
    pop = 1.0 / np.arange(1, V + 1)
    pop = pop[r.permutation(V)]
    cdf = np.cumsum(pop / pop.sum())

    def draw(n):
        return np.minimum(np.searchsorted(cdf, r.rand(n)), V - 1)
    succ = draw(V * n_succ).reshape(V, n_succ)
    lens = np.clip(np.exp(r.normal(p["len_mu"], p["len_sigma"], size=U)), p["len_min"], p["len_max"]).astype(np.int64)
    out = np.zeros((U, int(lens.max())), np.int32)
    cur = draw(U)
    for t in range(out.shape[1]):
        act = np.nonzero(lens > t)[0]
        nxt = np.where(r.rand(act.size) < follow, succ[cur[act], r.randint(n_succ, size=act.size)], draw(act.size))
        out[act, t] = nxt
        cur[act] = nxt
    hist = {}
    for u in range(U):
        row = out[u, :lens[u]]
        _, first = np.unique(row, return_index=True)
        hist[u + 1] = (row[np.sort(first)] + 1).tolist()
    return hist, U, V


def generate(preset="ml1m", seed=23, follow=0.6, n_succ=4):
    p = PRESETS[preset]
    if p.get("fast"):
        return _generate_fast(p, seed, follow, n_succ)
    r = np.random.RandomState(seed)
    V = p["items"]
    pop = 1.0 / np.arange(1, V + 1)
    pop = pop[r.permutation(V)]
    pop /= pop.sum()
    succ = np.stack([r.choice(V, size=n_succ, replace=False, p=pop) for _ in range(V)])   # preferred successors
    lens = np.clip(np.exp(r.normal(p["len_mu"], p["len_sigma"], size=p["users"])), p["len_min"], p["len_max"]).astype(int)
    hist = {}
    for u in range(1, p["users"] + 1):
        n = int(lens[u - 1])
        items = np.empty(n, np.int64)
        seen = set()
        cur = int(r.choice(V, p=pop))
        k = 0
        tries = 0
        while k < n:
            nxt = int(succ[cur, r.randint(n_succ)]) if r.rand() < follow else int(r.choice(V, p=pop))
            tries += 1
            if nxt in seen and tries < 50 * n:
                cur = nxt if r.rand() < 0.3 else cur
                continue
            seen.add(nxt)
            items[k] = nxt
            cur = nxt
            k += 1
        hist[u] = (items + 1).tolist()
    return hist, p["users"], V


def write(path, hist):
    with open(path, "w") as f:
        for u in sorted(hist):
            for it in hist[u]:
                f.write("%d %d\n" % (u, it))


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="ml1m")
    ap.add_argument("--seed", type=int, default=23)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    h, nu, ni = generate(a.preset, a.seed)
    write(a.out, h)
    print("users %d items %d actions %d" % (nu, ni, sum(len(v) for v in h.values())))
