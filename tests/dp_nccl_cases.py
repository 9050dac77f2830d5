"""GPU: the data-parallel code path of every fused trainer on the `nccl` backend (= RCCL) -- a 1-rank group on the single test
GPU, so the collectives really are RCCL calls on RCCL's stream: two-phase backward, tail bucket issued mid-backward
(adt_amd/dp.py:GradBuckets), head bucket at the end, clip + Adam after the reduce -- eagerly and captured in a HIP graph.
With one rank the all-reduce is the identity, so the DP step must land on the weights of the plain single-GPU step (same seeds,
same dropout stream); what this test guards is stream ordering: a missing wait() on RCCL's stream, or a collective that does
not survive graph capture, shows up as stale or torn gradients.
(Run by tests/test_dp_nccl.py in a child process; the file name keeps it out of the default collection.)"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

LR = 1e-3


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(23400 + os.getpid() % 2000)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _far(a, b):
    d = (a - b).abs()
    return float((d > 1e-5).float().mean()), float(d.max())


def _run(maker, pg, use_graph, nsteps=4):
    from tests import test_checkpoint_hip as T
    m, tr0, batches = getattr(T, maker)(11)
    kw = {}
    cls = type(tr0)
    # rebuild the trainer with the process group / graph flag (same hyper-parameters as the maker's)
    if maker == "_sasrec":
        tr = cls(m, [0.1, 0.05], [0.1, 0.01], lr=LR, weight_decay=1e-3, clip=5.0, process_group=pg, use_graph=use_graph, seed=5)
    elif maker == "_wide":
        tr = cls(m, [0.1], [0.1], lr=LR, weight_decay=1e-3, clip=5.0, process_group=pg, use_graph=use_graph, seed=5)
    elif maker == "_bert":
        tr = cls(m, [0.01, 0.01], [0.01, 0.01], lr=LR, weight_decay=1e-4, clip=5.0, process_group=pg, use_graph=use_graph, seed=5)
    else:
        tr = cls(m, [0.002], [0.001], lr=LR, process_group=pg, use_graph=use_graph, seed=5)
    for i in range(nsteps):
        tr.step(*batches[i % len(batches)], **kw)
    torch.cuda.synchronize()
    return m.flat.clone(), float(tr.loss()), float(tr.grad_norm())


@pytest.mark.parametrize("maker", ["_sasrec", "_wide", "_bert", "_stosa"])
def test_one_rank_rccl_step_equals_plain_step(maker, nccl_group):
    want, loss0, gn0 = _run(maker, None, False)
    for use_graph in (False, True):
        got, loss, gn = _run(maker, nccl_group, use_graph)
        frac, worst = _far(got, want)
        # float atomics reorder between runs: Adam turns a sign flip of a ~0 gradient into a +-lr move (see test_checkpoint_hip)
        assert frac < 1e-3 and worst <= 4 * 4 * LR, (maker, use_graph, frac, worst)
        assert abs(loss - loss0) <= 1e-3 * abs(loss0) and abs(gn - gn0) <= 1e-3 * gn0, (maker, use_graph, loss, loss0, gn, gn0)


def test_buckets_reduce_on_rccl(nccl_group):
    """GradBuckets itself on RCCL: both buckets of a flat buffer go through all_reduce (identity on one rank) and come back intact,
    eagerly and from a captured graph whose input changes between replays."""
    from adt_amd.dp import GradBuckets
    flat = torch.arange(10000, device="cuda:0", dtype=torch.float32)
    b = GradBuckets(flat, 3000, nccl_group, n=9000)
    assert b.active and b.tail.numel() == 6000 and b.head.numel() == 3000
    b.tail_ready()
    flat[:3000] += 1.0           # "the encoder backward" keeps writing the head bucket while the tail is in flight
    b.finish()
    torch.cuda.synchronize()
    ref = torch.arange(10000, device="cuda:0", dtype=torch.float32)
    ref[:3000] += 1.0
    assert torch.equal(flat, ref)
    g = torch.cuda.CUDAGraph()
    src = torch.zeros_like(flat)
    with torch.cuda.graph(g):
        flat.copy_(src)
        flat.mul_(2.0)
        b.tail_ready()
        flat[:3000] += 1.0
        b.finish()
        flat.add_(1.0)
    for k in (1.0, 5.0):
        src.fill_(k)
        g.replay()
        torch.cuda.synchronize()
        want = torch.full_like(flat, 2 * k + 1)
        want[:3000] += 1.0
        assert torch.equal(flat, want), k
