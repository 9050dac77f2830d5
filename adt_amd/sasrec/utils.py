"""Config surface, data split, batch sampling and ranking metrics of the reference's sasrec/utils.py, rebuilt
for a trainer that consumes a batch every millisecond: the per-sample Python loops of the reference
(WarpDataset.sample_data, PopularSampler.get_negative_samples, evaluate_loader) are vectorised over whole
batches, scoring + ranking runs on the GPU (adt_score_rank), and evaluation negatives can be frozen so that
two implementations are scored on the same split.  Names, argument meaning and return formats follow the
reference so that its callers read the same.
"""
import json
import os
from collections import Counter, defaultdict

import numpy as np


# ---- config surface ------------------------------------------------------------------------------------
def set_template(args, template_folder=None):
    """sasrec/utils.py:842-848: every key of templates/<dataset>.json OVERRIDES the CLI value of the same name."""
    folder = template_folder or os.path.join(os.path.dirname(os.path.abspath(__file__)), "templates")
    with open(os.path.join(folder, "%s.json" % args.dataset)) as f:
        for k, v in json.load(f).items():
            setattr(args, k, v)
    return args


_LAMBDAS = {  # sasrec/utils.py:855-862 (reconstruction lambda1 per layer, independence lambda2 per layer)
    "ml-1m": ([0.104292, 0.065892], [0.100833, 0.000607]),
    "beauty": ([0.0124, 0.122], [0.0001, 0.0]),
    "Beauty": ([0.0124, 0.122], [0.0001, 0.0]),
    "steam": ([0.0001, 0.0005], [0.00134, 0.00028]),
    "ml-20m": ([0.005, 0.1], [0.00186667, 0.075]),
}


def get_lambdas(dataset, tp=-1):
    """sasrec/utils.py:850-862.  Unknown datasets return None, as the reference falls off its if-chain."""
    if dataset in _LAMBDAS:
        l1, l2 = _LAMBDAS[dataset]
        return list(l1), list(l2)
    return None


def candidate_to_lambda(choices, prob):
    """candidates_to_lambdas.py:1-9: piecewise-linear interpolation of a search coordinate in [0, 1]."""
    split = 1.0 / (len(choices) - 1)
    idx = 0
    while prob > split:
        idx += 1
        prob -= split
    rel = prob / split
    return choices[idx] * (1 - rel) + choices[idx + 1] * rel


# ---- data ------------------------------------------------------------------------------------------------
def data_partition(fname, data_dir="data"):
    """sasrec/utils.py:320-350: one "user item" pair per line; users with < 3 actions are train-only, otherwise the
    last two actions become the validation and the test item."""
    path = fname if os.path.isfile(fname) else os.path.join(data_dir, "%s.txt" % fname)
    users = defaultdict(list)
    usernum = itemnum = 0
    with open(path) as f:
        for line in f:
            u, i = line.rstrip().split(" ")
            u, i = int(u), int(i)
            usernum, itemnum = max(usernum, u), max(itemnum, i)
            users[u].append(i)
    train, valid, test = {}, {}, {}
    for u, items in users.items():
        if len(items) < 3:
            train[u], valid[u], test[u] = items, [], []
        else:
            train[u], valid[u], test[u] = items[:-2], [items[-2]], [items[-1]]
    return train, valid, test, usernum, itemnum


class WarpDataset:
    """sasrec/utils.py:281-317: right-aligned history of a user (all but the last training item) as `seq`, the
    same shifted right by one as `dec` (dec[0] = 0), the next item as `pos` and a random unseen item as `neg`
    (0 where pos == 0).  sample_batch() builds a whole (B, L) batch with numpy; the negative is redrawn until it
    is not in the user's training set, like random_neq (sasrec/utils.py:73-77)."""

    def __init__(self, user_train, usernum, itemnum, maxlen, native=True):
        self.user_train, self.usernum, self.itemnum, self.maxlen = user_train, usernum, itemnum, maxlen
        self._sets = {}
        self._native = None
        if native:
            self._init_native()

    def _init_native(self):
        """CSR copy of the histories for libadt_host.so (adt_amd/csrc/adt_hostdata.cpp).  When the library has not been
        built the numpy sampler below is used: same semantics, ~100x slower."""
        from .. import _hostlib
        if not _hostlib.available():
            return
        lib = _hostlib.load()
        off = np.zeros(self.usernum + 2, np.int64)
        for u in range(1, self.usernum + 1):
            off[u + 1] = off[u] + len(self.user_train.get(u, []))
        items = np.zeros(max(int(off[-1]), 1), np.int32)
        for u in range(1, self.usernum + 1):
            h = self.user_train.get(u, [])
            items[off[u]:off[u + 1]] = h
        self._native = (lib, off, items)
        self._calls = 0

    def __len__(self):
        return self.usernum

    def _hist(self, user):
        return self.user_train.get(user, [])

    def sample_data(self, user, rng=np.random):
        L = self.maxlen
        seq = np.zeros(L, np.int32)
        dec = np.zeros(L + 1, np.int32)
        pos = np.zeros(L, np.int32)
        neg = np.zeros(L, np.int32)
        items = self._hist(user)
        n = min(len(items) - 1, L)
        if n > 0:
            hist = np.asarray(items[-(n + 1):], np.int32)
            seq[L - n:] = hist[:-1]
            pos[L - n:] = hist[1:]
            dec[L - n + 1:] = hist[:-1]
            ts = self._sets.setdefault(user, set(items))
            ng = rng.randint(1, self.itemnum + 1, size=n)
            for k in range(n):
                while ng[k] in ts:
                    ng[k] = rng.randint(1, self.itemnum + 1)
            neg[L - n:] = ng
        return user, seq, dec[:-1], pos, neg

    def __getitem__(self, i):
        user = i % self.usernum + 1
        while len(self._hist(user)) < 1:
            user = np.random.randint(1, self.usernum + 1)
        return self.sample_data(user), 0

    def sample_batch(self, users, rng=np.random):
        B, L = len(users), self.maxlen
        if self._native is not None:
            lib, off, items = self._native
            us = np.ascontiguousarray(users, dtype=np.int32)
            out = [np.empty((B, L), np.int32) for _ in range(4)]
            seed = self.next_seed(rng)
            rc = lib.adt_host_sample_batch(off.ctypes.data, items.ctypes.data, us.ctypes.data, B, L, self.itemnum, seed,
                                           out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data, 1)   # 1 thread: 0.2 ms / 256-sequence batch
            if rc != 0:
                raise RuntimeError("adt_host_sample_batch failed")
            return np.asarray(users), out[0], out[1], out[2], out[3]
        seq = np.zeros((B, L), np.int32)
        dec = np.zeros((B, L), np.int32)
        pos = np.zeros((B, L), np.int32)
        neg = np.zeros((B, L), np.int32)
        for b, u in enumerate(users):
            _, seq[b], dec[b], pos[b], neg[b] = self.sample_data(u, rng)
        return np.asarray(users), seq, dec, pos, neg

    def epoch_users(self, batch_size, rng=np.random, shuffle=True, drop_last=False):
        """The user ids of every batch of one epoch = usernum samples (sasrec/utils.py:316-317), shuffled like DataLoader(shuffle=True)."""
        order = np.arange(self.usernum)
        if shuffle:
            rng.shuffle(order)
        for s in range(0, len(order), batch_size):
            idx = order[s:s + batch_size]
            if drop_last and len(idx) < batch_size:
                break
            users = []
            for i in idx:
                u = int(i) % self.usernum + 1
                while len(self._hist(u)) < 1:
                    u = rng.randint(1, self.usernum + 1)
                users.append(u)
            yield users

    def epoch_batches(self, batch_size, rng=np.random, shuffle=True, drop_last=False):
        for users in self.epoch_users(batch_size, rng, shuffle, drop_last):
            yield self.sample_batch(users, rng)

    # ---- in-place sampling for the trainer's pinned id ring (adt_amd/sasrec/trainer.py: RingFeeder) --------------------------------
    def next_seed(self, rng=np.random):
        """The per-batch seed of the native sampler, drawn exactly as sample_batch() draws it."""
        self._calls += 1
        return (int(rng.randint(0, 2 ** 31 - 1)) * 2654435761 + self._calls) & (2 ** 64 - 1)

    def sample_rows_into(self, users, seed, out, b0=0):
        """Rows [b0, b0 + len(users)) of a global batch, written straight into the four (B, L) int32 arrays `out` (views of a ring slot).
        Native sampler only; callable from a producer thread (the C call releases the GIL)."""
        lib, off, items = self._native
        us = np.ascontiguousarray(users, dtype=np.int32)
        rc = lib.adt_host_sample_rows(off.ctypes.data, items.ctypes.data, us.ctypes.data, len(us), b0, self.maxlen, self.itemnum, seed,
                                      out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data, 1)
        if rc != 0:
            raise RuntimeError("adt_host_sample_rows failed")

    def count_targets(self, users):
        """Number of positions with a target in the batch of `users` (the BCE normaliser, sasrec/main.py:150-153), without sampling it."""
        lib, off, _ = self._native
        us = np.ascontiguousarray(users, dtype=np.int32)
        return int(lib.adt_host_count_targets(off.ctypes.data, us.ctypes.data, len(us), self.maxlen))


class PopularSampler:
    """sasrec/utils.py:19-69: negatives drawn by item popularity (p[i] = count of item id i for i in
    range(itemnum), exactly as the reference indexes it), excluding what the user has seen."""

    def __init__(self, train, val, test, usernum, itemnum, sample_size):
        self.train, self.val, self.test = train, val, test
        self.usernum, self.itemnum, self.sample_size = usernum, itemnum, sample_size
        pop = Counter()
        for u in range(1, usernum + 1):
            pop.update(train.get(u, []))
            pop.update(val.get(u, []))
            pop.update(test.get(u, []))
        self.popular_items = sorted(pop, key=pop.get, reverse=True)
        p = np.array([pop[i] for i in range(itemnum)], np.float64)
        self.popular_p = p / p.sum()

    def get_negative_samples(self, user, mode="valid", rng=np.random):
        seen = set(self.train.get(user, []))
        seen.update(self.val.get(user, []))
        if mode == "test":
            seen.update(self.test.get(user, []))
        out = []
        while len(out) < self.sample_size:
            ids = rng.choice(self.itemnum, 2 * self.sample_size, replace=False, p=self.popular_p)
            for x in ids:
                x = int(x)
                if x not in seen and x not in out:
                    out.append(x)
        return out[:self.sample_size]


class EvalDataset:
    """sasrec/utils.py:138-205.  frozen=True draws every user's negatives once (RandomState(seed)) so that
    repeated evaluations -- and different implementations -- see the same candidate sets; the reference redraws
    them on every access."""

    def __init__(self, user_train, user_val, user_test, usernum, itemnum, maxlen, negative_sampler, mode="val", eval_set=-1,
                 frozen=False, seed=23):
        self.user_train, self.user_val, self.user_test = user_train, user_val, user_test
        self.usernum, self.itemnum, self.maxlen, self.mode = usernum, itemnum, maxlen, mode
        self.negative_sampler = negative_sampler
        cand = range(1, usernum + 1) if eval_set is None or eval_set < 0 else \
            np.random.choice(np.arange(1, usernum + 1), eval_set, replace=False)
        tgt = user_val if mode == "val" else user_test
        self.users = [int(u) for u in cand if len(tgt.get(int(u), [])) != 0 and len(user_train.get(int(u), [])) != 0]
        self._frozen = None
        if frozen:
            r = np.random.RandomState(seed)
            self._frozen = {u: self._candidates(u, r) for u in self.users}

    def __len__(self):
        return len(self.users)

    def _candidates(self, user, rng):
        first = self.user_val[user][0] if self.mode == "val" else self.user_test[user][0]
        return np.array([first] + self.negative_sampler.get_negative_samples(user, mode=self.mode, rng=rng), np.int32)

    def sample_data(self, user, rng=np.random):
        L = self.maxlen
        seq = np.zeros(L, np.int32)
        hist = list(self.user_train[user])
        if self.mode == "test":
            hist = hist + [self.user_val[user][0]]
        n = min(len(hist), L)
        seq[L - n:] = hist[-n:]
        item_idx = self._frozen[user] if self._frozen is not None else self._candidates(user, rng)
        label = np.zeros(len(item_idx), np.int64)
        label[0] = 1
        return user, seq, item_idx, label

    def __getitem__(self, i):
        user, seq, item_idx, label = self.sample_data(self.users[i])
        return (user, seq, item_idx), label

    def batches(self, batch_size):
        for s in range(0, len(self.users), batch_size):
            us = self.users[s:s + batch_size]
            rows = [self.sample_data(u) for u in us]
            yield (np.array([r[0] for r in rows]), np.stack([r[1] for r in rows]), np.stack([r[2] for r in rows])), \
                np.stack([r[3] for r in rows])


def metrics_from_ranks(ranks, n_candidates, ks=(5, 10)):
    """The accumulation of sasrec/utils.py:412-427: HR@k = #(rank < k)/N, NDCG@k = sum 1/log2(rank+2)/N,
    AUC = mean((S - (rank+1)) / (S - 1)) with S = 1 + n_candidates (the +1 double-counts the positive)."""
    ranks = np.asarray(ranks, np.int64)
    n = float(len(ranks))
    ndcg, hr = {}, {}
    for k in ks:
        hit = ranks < k
        hr[k] = float(hit.sum()) / n
        ndcg[k] = float((1.0 / np.log2(ranks[hit] + 2.0)).sum()) / n
    S = 1 + n_candidates
    return (ndcg, hr), float(np.mean((S - (ranks + 1)) / (S - 1)))


def rank_stats(ranks, n_candidates, ks=(5, 10)):
    """Additive sufficient statistics of metrics_from_ranks: [N, sum((S-(rank+1))/(S-1)), then per k: hits, sum 1/log2(rank+2)]
    as float64 -- what data-parallel evaluation sum-reduces across ranks."""
    ranks = np.asarray(ranks, np.int64)
    S = 1 + n_candidates
    out = [float(len(ranks)), float(((S - (ranks + 1)) / (S - 1)).sum())]
    for k in ks:
        hit = ranks < k
        out += [float(hit.sum()), float((1.0 / np.log2(ranks[hit] + 2.0)).sum())]
    return np.array(out, np.float64)


def metrics_from_stats(stats, ks=(5, 10)):
    n = float(stats[0])
    ndcg = {k: float(stats[3 + 2 * i]) / n for i, k in enumerate(ks)}
    hr = {k: float(stats[2 + 2 * i]) / n for i, k in enumerate(ks)}
    return (ndcg, hr), float(stats[1]) / n


def reduce_rank_stats(stats, process_group=None):
    """Sum the per-rank statistics over the group (gloo on CPU tensors, RCCL needs device tensors: float64 either way)."""
    import torch
    import torch.distributed as dist
    if process_group is None or dist.get_world_size(process_group) == 1:
        return stats
    t = torch.from_numpy(np.ascontiguousarray(stats))
    if dist.get_backend(process_group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=process_group)
    return t.cpu().numpy()


def evaluate_loader(model, loader, args=None, mode="val", ks=(5, 10), process_group=None):
    """sasrec/utils.py:395-428 -> ((NDCG, HT), AUC).  `loader` yields ((u, seq, item_idx), label) batches (an
    EvalDataset.batches() generator or a torch DataLoader); scoring and the rank of the positive run on the GPU.
    With a process group, rank r scores batches r, r+W, ... and the additive statistics are sum-reduced, so every rank
    returns the metrics of the whole user set (identical to the single-process result: the statistics are sums)."""
    r, W = 0, 1
    if process_group is not None:
        import torch.distributed as dist
        r, W = dist.get_rank(process_group), dist.get_world_size(process_group)
    ranks, ncand = [], None
    for i, ((u, seq, item_idx), _) in enumerate(loader):
        item_idx = np.asarray(item_idx)
        ncand = item_idx.shape[1]
        if i % W != r:
            continue
        _, rank = model.predict_rank(np.asarray(seq), item_idx)
        ranks.append(rank.cpu().numpy())
    ranks = np.concatenate(ranks) if ranks else np.zeros(0, np.int64)
    if W == 1:
        return metrics_from_ranks(ranks, ncand, ks)
    return metrics_from_stats(reduce_rank_stats(rank_stats(ranks, ncand, ks), process_group), ks)
