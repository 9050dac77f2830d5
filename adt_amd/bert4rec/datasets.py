"""Host data pipeline of BERT4Rec-ADT -- the counterpart of the reference's bert4rec/datasets/dataset.py: the same split
(data_partition :16-47), the same masked-sequence generation (BertTrainDataset :49-156: per user `dupe_factor` randomly
masked copies of every sliding window plus one copy with only the last item masked; 80 % [MASK] / 10 % random item / 10 %
kept; the decoder input always ends in [MASK]) and the same evaluation rows (BertEvalDataset :176-222: history + [MASK],
positive + popularity negatives).  Arrays are built once with numpy instead of per-sample LongTensors, and batches are
plain int32 arrays for FusedBertTrainer."""
import random
from collections import Counter, defaultdict

import numpy as np


def data_partition(fname, data_dir="data"):
    """bert4rec/datasets/dataset.py:16-47 (users with < 3 interactions are train-only)."""
    usernum = itemnum = 0
    User = defaultdict(list)
    with open("%s/%s.txt" % (data_dir, fname)) as f:
        for line in f:
            u, i = line.rstrip().split(" ")
            u, i = int(u), int(i)
            usernum, itemnum = max(u, usernum), max(i, itemnum)
            User[u].append(i)
    train, valid, test = {}, {}, {}
    for user, items in User.items():
        if len(items) < 3:
            train[user], valid[user], test[user] = items, [], []
        else:
            train[user], valid[user], test[user] = items[:-2], [items[-2]], [items[-1]]
    return [train, valid, test, usernum, itemnum]


class PopularSampler:
    """bert4rec/datasets/negative_sampler.py PopularSampler: `sample_size` negatives per user drawn by item popularity
    among the items the user has not interacted with."""

    def __init__(self, train, val, test, usernum, itemnum, sample_size, seed=23):
        cnt = Counter()
        for d in (train, val, test):
            for items in d.values():
                cnt.update(items)
        self.items = np.array(sorted(cnt), np.int64)
        p = np.array([cnt[i] for i in self.items], np.float64)
        self.p = p / p.sum()
        self.train, self.val, self.test, self.sample_size = train, val, test, sample_size
        self.rng = np.random.RandomState(seed)
        self._cache = {}

    def get_negative_samples(self, user, mode="val"):
        key = (user, mode)
        if key not in self._cache:      # frozen per (user, mode): two evaluations rank the same candidates
            seen = set(self.train.get(user, [])) | set(self.val.get(user, [])) | set(self.test.get(user, []))
            out = []
            while len(out) < self.sample_size:
                for it in self.rng.choice(self.items, size=2 * self.sample_size, p=self.p):
                    if it not in seen and it not in out:
                        out.append(int(it))
                        if len(out) == self.sample_size:
                            break
            self._cache[key] = out
        return self._cache[key]


class BertTrainDataset:
    def __init__(self, user_train, usernum, itemnum, maxlen, mask_prob, seed, dupe_factor=10, prop_sliding_window=0.5):
        self.itemnum, self.maxlen, self.mask_prob = itemnum, maxlen, mask_prob
        self.mask_token = itemnum + 1
        self.rng = random.Random(seed)
        self.dupe_factor, self.prop_sliding_window = dupe_factor, prop_sliding_window
        src, dec, lab = [], [], []
        for user in range(1, usernum + 1):
            seqs = user_train.get(user, [])
            if len(seqs) < 1:
                continue
            if len(seqs) <= maxlen:
                windows = [seqs]
            else:       # dataset.py:84-94
                step = int(prop_sliding_window * maxlen) if prop_sliding_window != -1 else maxlen
                beg = list(range(len(seqs) - maxlen, 0, -step)) + [0]
                windows = [seqs[i:i + maxlen] for i in beg[::-1]]
            for wdw in windows:
                for _ in range(dupe_factor):
                    t, d, l = self.sample_data(wdw)
                    src.append(t), dec.append(d), lab.append(l)
            t, d, l = self._mask_last(seqs)
            src.append(t), dec.append(d), lab.append(l)
        self.src, self.dec, self.labels = (np.array(x, np.int32) for x in (src, dec, lab))

    def _pad(self, x):
        x = x[-self.maxlen:]
        return [0] * (self.maxlen - len(x)) + x

    def _mask_last(self, seq):
        """dataset.py:100-123."""
        tokens, labels = list(seq), [0] * len(seq)
        labels[-1] = seq[-1]
        tokens[-1] = self.mask_token
        return self._pad(tokens), self._pad(list(tokens)), self._pad(labels)

    def sample_data(self, seq):
        """dataset.py:125-156."""
        tokens, dec_tokens, labels = [], [], []
        for s in seq:
            prob = self.rng.random()
            if prob < self.mask_prob:
                prob /= self.mask_prob
                if prob < 0.8:
                    tok = self.mask_token
                elif prob < 0.9:
                    tok = self.rng.randint(1, self.itemnum)
                else:
                    tok = s
                tokens.append(tok), dec_tokens.append(tok), labels.append(s)
            else:
                tokens.append(s), dec_tokens.append(s), labels.append(0)
        dec_tokens[-1] = self.mask_token
        return self._pad(tokens), self._pad(dec_tokens), self._pad(labels)

    def __len__(self):
        return len(self.src)

    def epoch_batches(self, batch_size, rng, shuffle=True):
        order = rng.permutation(len(self)) if shuffle else np.arange(len(self))
        for s in range(0, len(order), batch_size):
            idx = order[s:s + batch_size]
            yield self.src[idx], self.dec[idx], self.labels[idx]


class BertEvalDataset:
    def __init__(self, user_train, user_val, user_test, usernum, itemnum, maxlen, negative_sampler, mode="val", eval_set=-1):
        self.user_train, self.user_val, self.user_test = user_train, user_val, user_test
        self.maxlen, self.sampler, self.mode, self.mask_token = maxlen, negative_sampler, mode, itemnum + 1
        pool = random.sample(range(1, usernum + 1), eval_set) if eval_set >= 0 else range(1, usernum + 1)
        tgt = user_val if mode == "val" else user_test
        self.users = [u for u in pool if len(tgt.get(u, [])) != 0 and len(user_train.get(u, [])) != 0]

    def sample_data(self, user):
        """dataset.py:201-215."""
        answer = [(self.user_val if self.mode == "val" else self.user_test)[user][0]]
        cand = answer + self.sampler.get_negative_samples(user, mode=self.mode)
        seq = (list(self.user_train[user]) + [self.mask_token])[-self.maxlen:]
        return [0] * (self.maxlen - len(seq)) + seq, cand

    def __len__(self):
        return len(self.users)

    def batches(self, batch_size):
        for s in range(0, len(self.users), batch_size):
            rows = [self.sample_data(u) for u in self.users[s:s + batch_size]]
            yield np.array([r[0] for r in rows], np.int32), np.array([r[1] for r in rows], np.int32)
