"""GPU parity of the fused all-item logits + cross-entropy kernels (adt_amd/csrc/adt_lce.cuh, C ABI adt_lce_fwd_bwd) against the
oracle's tape (oracle/tape.py: linear + cross_entropy(ignore_index=0), the restatement of bert4rec/model/bert.py:80-90 and
bert4rec/trainer.py:45,113-115) on seeded inputs: loss, per-row log-sum-exp, dh (scattered to the masked rows), dE, dbias.

Cases: ragged row / item counts (not multiples of 32 / 256 / 512), a device-side row count below the capacity, a single row, the
BASELINE config-3 vocabulary (V + 100 = 26,844, d = 256), a small grid (every workgroup walks several work items), repeated labels.
Tolerance (bf16 operands, fp32 accumulation; the probabilities are rounded to bf16 before the gradient products): loss / lse 2e-3
absolute on values of ~ln V, gradients 2e-2 of the tensor magnitude -- the bound of every bf16-mode kernel test in this repo."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import tape as tp  # noqa: E402


def dev():
    return torch.device("cuda:0")


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)


def bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float32).numpy()


def run_case(T, K, V, M, mcap, seed, slots=0, scale=1.0, few_labels=False):
    from adt_amd import _lib, ops
    lib = _lib.load()
    prev = lib.adt_lce_slots(0)
    if slots:
        lib.adt_lce_slots(slots)
    try:
        r = np.random.RandomState(seed)
        h = (scale * r.standard_normal((T, K))).astype(np.float32)
        E = (scale * r.standard_normal((V, K)) / np.sqrt(K)).astype(np.float32)
        bias = (0.3 * r.standard_normal(V)).astype(np.float32)
        rows = np.sort(r.choice(T, size=M, replace=False)).astype(np.int32)
        labels = (r.randint(1, 4, size=M) if few_labels else r.randint(1, V, size=M)).astype(np.int32)
        rows_p = np.zeros(mcap, np.int32)
        rows_p[:M] = rows
        lab_p = np.zeros(mcap, np.int32)
        lab_p[:M] = labels
        inv = 1.0 / max(M, 1)
        # oracle: the operands the matrix cores see are the bf16-rounded ones
        vh, vE, vb = tp.leaf(bf16_round(h[rows])), tp.leaf(bf16_round(E)), tp.leaf(bias)
        logits = tp.linear(vh, vE, vb)
        ce = tp.cross_entropy(logits, labels, ignore_index=0)
        tp.backward(ce)
        z = logits.v.astype(np.float64)
        lse_ref = np.log(np.exp(z - z.max(1, keepdims=True)).sum(1)) + z.max(1)
        # kernel
        hg, Eg, bg = T_(h), T_(E), T_(bias)
        dh = torch.full((T, K), 7.0, device=dev())          # rows that are not masked must not be touched
        dE = torch.full((V, K), 0.5, device=dev())          # accumulated into
        db = torch.full((V,), -0.25, device=dev())
        loss64 = torch.zeros(64, device=dev())
        lse = torch.zeros(mcap, device=dev())
        m_dev = torch.tensor([M], device=dev(), dtype=torch.int32)
        invc = torch.tensor([inv], device=dev(), dtype=torch.float32)
        ops.lce_fwd_bwd(hg, T_(rows_p), T_(lab_p), mcap, m_dev, Eg, bg, invc, loss64, dh, dE, db, lse_out=lse)
        torch.cuda.synchronize()
        assert abs(float(loss64.sum()) - float(ce.v)) < 2e-3, (float(loss64.sum()), float(ce.v))
        assert np.abs(lse[:M].cpu().numpy() - lse_ref).max() < 2e-3
        dh_c = dh.cpu().numpy()
        untouched = np.setdiff1d(np.arange(T), rows)
        assert (dh_c[untouched] == 7.0).all()
        assert rel(dh_c[rows], vh.g) < 2e-2, rel(dh_c[rows], vh.g)
        assert rel(dE.cpu().numpy() - 0.5, vE.g) < 2e-2, rel(dE.cpu().numpy() - 0.5, vE.g)
        assert rel(db.cpu().numpy() + 0.25, vb.g) < 2e-2, rel(db.cpu().numpy() + 0.25, vb.g)
    finally:
        lib.adt_lce_slots(prev)


@pytest.mark.parametrize("T,K,V,M,mcap,slots", [
    (64, 256, 100, 1, 64, 0),              # one row
    (300, 256, 1000, 37, 300, 0),          # ragged everything
    (700, 128, 2077, 600, 700, 0),         # two forward X blocks, K = 128
    (1400, 256, 3416, 1111, 1400, 0),      # several blocks, ml-1m vocabulary
    (1400, 256, 3416, 1111, 1400, 8),      # 8 workgroups: every one walks several (X block, Y range) items
    (900, 128, 517, 520, 900, 16),
    (513, 256, 300, 300, 513, 0),          # device count below the capacity
])
def test_lce_against_oracle(T, K, V, M, mcap, slots):
    run_case(T, K, V, M, mcap, seed=T + V + M, slots=slots)


def test_lce_repeated_labels_and_large_logits():
    run_case(400, 256, 260, 333, 400, seed=5, few_labels=True)       # many rows share a label: the rank-one label terms collide
    run_case(400, 256, 700, 200, 400, seed=6, scale=3.0)             # logits of +-30: the running maximum matters


def test_lce_config3_vocabulary():
    """BASELINE config 3's output layer: V + 100 = 26,844 items, d = 256.  1,500 masked rows of a 6,000-row activation slice keep the
    oracle's (M, V) logits and their tape small enough for the CPU side of the test."""
    run_case(6000, 256, 26844, 1500, 6000, seed=11)


def test_lce_zero_rows_is_a_no_op():
    from adt_amd import ops
    T, K, V = 64, 256, 500
    h, E, b = torch.randn(T, K, device=dev()), torch.randn(V, K, device=dev()), torch.zeros(V, device=dev())
    dh, dE, db = torch.zeros(T, K, device=dev()), torch.zeros(V, K, device=dev()), torch.zeros(V, device=dev())
    loss64 = torch.zeros(64, device=dev())
    z = torch.zeros(T, device=dev(), dtype=torch.int32)
    ops.lce_fwd_bwd(h, z, z, T, torch.tensor([0], device=dev(), dtype=torch.int32), E, b, torch.tensor([1.0], device=dev()), loss64, dh, dE, db)
    torch.cuda.synchronize()
    assert float(loss64.abs().sum()) == 0 and float(dh.abs().sum()) == 0 and float(dE.abs().sum()) == 0 and float(db.abs().sum()) == 0
