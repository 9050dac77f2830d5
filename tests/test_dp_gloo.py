"""CPU, world_size 2, gloo: the data-parallel layer every fused trainer uses (adt_amd/dp.py: shard_bounds, global_norms,
GradBuckets) reproduces the single-process step on the same global batch -- with dropout ON (global mask indices), an uneven
split (5 sequences over 2 ranks) and the two-bucket all-reduce with the tail bucket issued before the backward has finished.
The compute engine here is the numpy oracle (test infrastructure standing in for the HIP engine, which needs a GPU); the
same GradBuckets object drives RCCL in the HIP trainers (tests/test_dp_gpu.py, tests/test_dp_nccl.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

LAM1, LAM2, WD = [0.104292, 0.065892], [0.100833, 0.000607], 1e-3


def _setup():
    from oracle import sasrec_oracle as so
    from tools.gen_golden_inputs import make_batch
    cfg = so.Cfg(60, 12, 32, 2, 2, dropout=0.5)
    P = so.init_params(cfg, seed=4)
    batch = make_batch(np.random.RandomState(5), 5, cfg.maxlen, cfg.item_num)
    return so, cfg, P, batch


class OracleEngine:
    def __init__(self, so, cfg, P, seed):
        self.so, self.cfg, self.P, self.seed, self.state = so, cfg, P, seed, {}
        self.names = [n for n, _ in so.param_shapes(cfg)]
        self.sizes = [int(np.prod(s)) for _, s in so.param_shapes(cfg)]

    def forward_backward(self, shard, norms, b_offset):
        so, cfg = self.so, self.cfg
        flat = np.zeros(sum(self.sizes), np.float32)
        if len(shard[0]):
            out = so.forward(self.P, cfg, *shard, training=True, seed=self.seed, b_offset=b_offset)
            _, _, seeds = so.loss_and_seeds(self.P, cfg, out, shard[2], LAM1, LAM2, WD, norms)
            G = so.backward(self.P, cfg, out[5], seeds, WD, add_wd=False)
            o = 0
            for n, sz in zip(self.names, self.sizes):
                if G[n] is not None:
                    flat[o:o + sz] = G[n].reshape(-1)
                o += sz
        return torch.from_numpy(flat)

    def apply(self, flat):
        so = self.so
        G, o = {}, 0
        for (n, shp), sz in zip(so.param_shapes(self.cfg), self.sizes):
            G[n] = None if so.is_unused(n, self.cfg.num_heads) else flat[o:o + sz].numpy().reshape(shp).copy()
            o += sz
        E = self.P["item_emb.weight"]
        nrm = np.sqrt((E.astype(np.float64) ** 2).sum())
        G["item_emb.weight"] = G["item_emb.weight"] + (WD / nrm * E).astype(np.float32)   # once, after the reduce
        so.clip_adam(self.P, G, self.state, lr=1e-3, clip=5.0)


def dp_step(eng, cfg, batch, rank, world, group):
    """What each fused trainer's _launch does, with the oracle as the engine: own shard with global normalisers -> gradient ->
    tail bucket out, head bucket out, wait -> weight-decay term + clip + Adam on the reduced buffer."""
    from adt_amd.dp import GradBuckets, global_norms, shard_bounds
    seq, dec, pos, neg = batch
    lo, hi = shard_bounds(len(seq), rank, world)
    flat = eng.forward_backward((seq[lo:hi], dec[lo:hi], pos[lo:hi], neg[lo:hi]), global_norms(pos, cfg.hidden_units, cfg.num_heads), lo)
    buckets = GradBuckets(flat, flat.numel() // 3, group)
    buckets.tail_ready()
    buckets.finish()
    eng.apply(flat)
    return flat


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    from adt_amd.dp import init_from_env
    pg, r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world) and pg is not None
    so, cfg, P, batch = _setup()
    eng = OracleEngine(so, cfg, P, seed=77)
    for _ in range(2):
        flat = dp_step(eng, cfg, batch, rank, world, pg)
    if rank == 0:
        q.put(({k: v.copy() for k, v in P.items()}, flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_single_process_step():
    so, cfg, P, batch = _setup()
    from adt_amd.dp import shard_bounds, global_norms, skip_batch
    assert [shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(256, r, 8) for r in range(8)][-1] == (224, 256)
    # the 34-row trailing batch of Amazon-Beauty (40,226 % 256) over 8 ranks: balanced, nobody empty, rows covered once
    b34 = [shard_bounds(34, r, 8) for r in range(8)]
    assert b34[0] == (0, 5) and b34[-1] == (30, 34) and all(hi > lo for lo, hi in b34) and all(b34[i][1] == b34[i + 1][0] for i in range(7))
    assert skip_batch(5, 8) and not skip_batch(8, 8)
    assert global_norms(batch[2], 32, 2)[1:] == (5 * 12 * 32.0, 5 * 12 * 2.0)
    eng = OracleEngine(so, cfg, P, seed=77)
    for _ in range(2):
        flat1 = dp_step(eng, cfg, batch, 0, 1, None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    P2, flat2 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g1 = flat1.numpy()
    assert np.abs(g1 - flat2).max() <= 2e-6 * max(1.0, np.abs(g1).max())
    o = 0
    for (k, shp), sz in zip(so.param_shapes(cfg), eng.sizes):
        # two Adam steps: entries whose true gradient is zero (key bias: softmax is shift-invariant) move by +-lr on
        # rounding noise (see test_oracle_golden); everything else must agree to rounding
        d = np.abs(P[k] - P2[k]).reshape(-1)
        noisy = np.abs(g1[o:o + sz]) < 1e-6
        o += sz
        assert d.max() <= 2 * 1e-3 * 1.01, k
        assert d[~noisy].max(initial=0.0) <= 2e-5, k


# ---- sharded evaluation (SURVEY.md 8(e) rule 5) -------------------------------------------------------------------------
class _RankModel:
    """Stands in for the HIP model on CPU ranks: the 'rank' of a user is a fixed function of its sequence."""

    def predict_rank(self, seq, item_idx):
        import torch
        return None, torch.from_numpy((np.asarray(seq).sum(1) * 7919 % 101).astype(np.int64))


def _eval_batches():
    r = np.random.RandomState(3)
    for n in (64, 64, 64, 64, 37):          # ragged last batch, odd number of batches
        seq = r.randint(0, 50, size=(n, 12))
        yield (np.arange(n), seq, np.zeros((n, 101), np.int64)), None


def _eval_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adt_amd.sasrec import utils as U
    out = U.evaluate_loader(_RankModel(), _eval_batches(), ks=(5, 10), process_group=dist.group.WORLD)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_equals_single_process():
    from adt_amd.sasrec import utils as U
    single = U.evaluate_loader(_RankModel(), _eval_batches(), ks=(5, 10))
    ranks = np.concatenate([_RankModel().predict_rank(b[0][1], None)[1].numpy() for b in _eval_batches()])
    (nd, hr), auc = U.metrics_from_stats(U.rank_stats(ranks, 101))
    assert nd == pytest.approx(single[0][0]) and hr == pytest.approx(single[0][1]) and auc == pytest.approx(single[1])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):          # every rank holds the metrics of the whole user set
        (nd2, hr2), auc2 = outs[r]
        for k in (5, 10):
            assert nd2[k] == pytest.approx(single[0][0][k], abs=1e-12) and hr2[k] == pytest.approx(single[0][1][k], abs=1e-12)
        assert auc2 == pytest.approx(single[1], abs=1e-12)
