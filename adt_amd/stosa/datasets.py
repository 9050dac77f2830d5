"""Host data pipeline of STOSA-ADT -- the counterpart of the reference's stosa/utils.py:get_user_seqs (:132-149),
generate_rating_matrix_{valid,test} (:96-130), neg_sample (:32-36) and stosa/datasets.py:DisenDataset (:202-300): the
same leave-two-out layout (train / valid / test views of one sequence), left padding, negatives outside the user's item
set.  Batches are built with numpy (one vectorised negative draw per batch with rejection) instead of per-sample tensors."""
import numpy as np
from scipy.sparse import csr_matrix


def get_user_seqs(data_file):
    """(user_seq, max_item, valid_rating_matrix, test_rating_matrix, num_users); each line is `user item item ...`."""
    user_seq, max_item = [], 0
    with open(data_file) as f:
        for line in f:
            _, items = line.strip().split(" ", 1)
            items = [int(x) for x in items.split(" ")]
            user_seq.append(items)
            max_item = max(max_item, max(items))
    num_users, num_items = len(user_seq), max_item + 2
    return user_seq, max_item, rating_matrix(user_seq, num_users, num_items, 2), rating_matrix(user_seq, num_users, num_items, 1), num_users


def rating_matrix(user_seq, num_users, num_items, holdout):
    """generate_rating_matrix_valid (holdout 2) / _test (holdout 1): the items a user has seen before the answer."""
    row, col = [], []
    for u, items in enumerate(user_seq):
        seen = items[:-holdout]
        row += [u] * len(seen)
        col += seen
    return csr_matrix((np.ones(len(row)), (np.array(row), np.array(col))), shape=(num_users, num_items))


class DisenDataset:
    def __init__(self, args, user_seq, data_type="train", eval_set=-1, seed=42):
        assert data_type in ("train", "valid", "test")
        self.args, self.user_seq, self.data_type, self.max_len = args, user_seq, data_type, args.maxlen
        self.n = len(user_seq) if eval_set == -1 else min(eval_set, len(user_seq))
        self.sets = [set(s) for s in user_seq]
        self.rng = np.random.RandomState(seed)

    def __len__(self):
        return self.n

    def _views(self, items):
        """datasets.py:230-246."""
        if self.data_type == "train":
            return items[:-3], items[1:-2], items[:-4], [0]
        if self.data_type == "valid":
            return items[:-2], items[1:-1], items[:-3], [items[-2]]
        return items[:-1], items[1:], items[:-2], [items[-1]]

    def batch(self, users):
        L, size = self.max_len, self.args.item_size
        B = len(users)
        inp, dec, pos, neg = (np.zeros((B, L), np.int32) for _ in range(4))
        ans = np.zeros((B, 1), np.int64)
        for r, u in enumerate(users):
            input_ids, target_pos, dec_ids, answer = self._views(self.user_seq[u])
            input_ids, target_pos, dec_ids = input_ids[-L:], target_pos[-L:], dec_ids[-L:]
            n = len(input_ids)
            if n:
                inp[r, L - n:], pos[r, L - n:] = input_ids, target_pos
                tn = self.rng.randint(1, size, size=n)      # neg_sample: uniform in [1, item_size - 1], outside the user's items
                bad = np.array([t in self.sets[u] for t in tn])
                while bad.any():
                    tn[bad] = self.rng.randint(1, size, size=int(bad.sum()))
                    bad = np.array([t in self.sets[u] for t in tn])
                neg[r, L - n:] = tn
            if len(dec_ids):
                dec[r, L - len(dec_ids):] = dec_ids
            ans[r, 0] = answer[0]
        return np.asarray(users, np.int64), inp, dec, pos, neg, ans

    def epoch_batches(self, batch_size, shuffle=True):
        order = self.rng.permutation(self.n) if shuffle else np.arange(self.n)
        for s in range(0, self.n, batch_size):
            yield self.batch(order[s:s + batch_size])
