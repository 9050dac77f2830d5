"""GPU, 2 processes: the data-parallel FusedTrainer path (two-phase backward, two-bucket async all-reduce, global
normalisers, global dropout indices, clip + Adam after the reduce) reproduces the single-process step on the same global
batch.  Both ranks share cuda:0 here (the test box has one GPU), so the process group uses gloo on device tensors --
RCCL refuses two ranks on one device; everything above the collective call is the code that runs over RCCL/xGMI."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAM1, LAM2, WD = [0.104292, 0.065892], [0.100833, 0.000607], 1e-3


def _build(prec="f32"):
    sys.path.insert(0, REPO)
    from oracle import sasrec_oracle as so
    from tests.test_hip_model import build
    from tools.gen_golden_inputs import make_batch
    cfg = so.Cfg(300, 52, 64, 2, 2, dropout=0.5)      # L % 4 == 0: in bf16 the per-sequence lean kernels bench.py times
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), 6, cfg.maxlen, cfg.item_num)
    return cfg, build(cfg, P, prec, dropout=0.5), batch


def _nsteps(prec):
    """f32: two steps.  bf16: one -- after a step the key-projection biases (exactly-zero gradient: the softmax is shift-invariant) differ
    by +-lr of Adam-amplified rounding noise between the two runs, which in bf16 re-rounds every key and moves the second step by ~1e-3."""
    return 2 if prec == "f32" else 1


def _worker(rank, world, port, q, prec="f32"):
    import torch.distributed as dist
    if prec.endswith("-2phase"):      # the two-bucket exchange (decoder bucket overlapped with the encoder backward) instead of the one-phase default
        os.environ["ADT_DP_PHASES"] = "2"
        prec = prec[:-7]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adt_amd.dp import shard_bounds, global_norms
    from adt_amd.sasrec.trainer import FusedTrainer
    cfg, m, batch = _build(prec)
    m.train()
    tr = FusedTrainer(m, LAM1, LAM2, lr=1e-3, weight_decay=WD, clip=5.0, process_group=dist.group.WORLD, seed=5)
    for _ in range(_nsteps(prec)):
        lo, hi = shard_bounds(len(batch[0]), rank, world)
        tr.step(*[a[lo:hi] for a in batch], norms=global_norms(batch[2], 64, 2), b_offset=lo)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((m.flat.cpu().numpy(), m.flat_grad.cpu().numpy(), float(tr.grad_norm())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16-2phase"])
def test_two_rank_trainer_matches_single_process(prec):
    import torch.multiprocessing as mp
    from adt_amd.sasrec.trainer import FusedTrainer
    wprec, prec = prec, prec.split("-")[0]
    cfg, m, batch = _build(prec)
    if prec == "bf16":
        assert m.lib.adt_seq_layer_supported(1, cfg.maxlen, 64, 32) == 1
    m.train()
    tr = FusedTrainer(m, LAM1, LAM2, lr=1e-3, weight_decay=WD, clip=5.0, seed=5)
    for _ in range(_nsteps(prec)):
        tr.step(*batch)
    torch.cuda.synchronize()
    w1, g1, n1 = m.flat.cpu().numpy(), m.flat_grad.cpu().numpy(), float(tr.grad_norm())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, wprec)) for r in range(2)]
    for p in procs:
        p.start()
    w2, g2, n2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # one workgroup = one sequence in both runs: the shards compute the same per-sequence arithmetic, only the order of the
    # fp32 sums over sequences (partials / atomics / all-reduce) differs
    assert abs(n1 - n2) <= 1e-4 * n1
    assert np.abs(g1 - g2).max() <= 5e-5 * max(np.abs(g1).max(), 1e-6)
    d = np.abs(w1 - w2)
    noisy = np.abs(g1) < 1e-6           # Adam turns rounding noise on exactly-zero gradients into +-lr (see test_oracle_golden)
    assert d.max() <= 2 * 1e-3 * 1.01 and d[~noisy].max() <= 3e-5
