"""bench.py's multi-rank branch on ONE GPU: two ranks over gloo (ADT_DIST_BACKEND=gloo) through the driver's launch line
(python -m torch.distributed.run ... bench.py --gpus 2).  One JSON line, world_size 2, both scalings in it, and the strong-scaling loss (global
batch 256 split over two ranks) equal to the loss one rank computes on the same global batches after the same number of steps
(sasrec/main.py:146-173 is the step; SURVEY 8e the exactness rules)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d: %r" % (len(lines), lines[:3])
    return json.loads(lines[0])


def test_bench_two_gloo_ranks_one_gpu():
    env = dict(os.environ, ADT_BENCH_BLOCKS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-ndcg", "--no-roofline"]
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, env)
    assert one["n_gpus"] == 1 and one["config"]["world_size"] == 1 and one["scaling"] == "weak"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2"] + common, dict(env, ADT_DIST_BACKEND="gloo"))
    assert two["n_gpus"] == 2 and two["config"]["world_size"] == 2 and two["config"]["backend"] == "gloo"
    assert two["scaling"] == "weak" and two["config"]["global_batch"] == 512
    assert "strong" in two and two["strong"]["global_batch"] == 256 and two["strong"]["sequences_per_gpu"] == 128
    for key in ("value", "ms_per_step", "value_incl_h2d", "ms_per_step_incl_h2d"):
        assert two[key] > 0 and two["strong"][key] > 0
    assert two["timing"]["blocks"] == 2
    # the same global batches, the same number of steps: two ranks on half the rows each reproduce the one-rank loss (bf16 kernels: atomics
    # in the item-table gradient reorder sums, 2e-3)
    assert abs(two["strong"]["loss_last_step"] - one["loss_last_step"]) < 2e-3, (two["strong"]["loss_last_step"], one["loss_last_step"])
