"""CPU: pins the C++ / OpenMP restatement (oracle/csrc/adt_cpu.cpp -> oracle/libadt_cpu.so; the `adt_cpu_*` twins of the model-level C ABI and
bench.py's cpu_baseline) to (a) the golden vectors recorded from the imported reference (tools/gen_golden.py): forward tensors, loss, every
parameter gradient incl. which stay untouched, clip norm, weights after 1 and 3 Adam steps; (b) the numpy oracle with dropout ON -- both
draw their masks from the shared hash RNG, so the whole step is comparable; (c) itself across thread counts and data-parallel shards.
fp32 vs fp32: 2e-5 on O(1) tensors (stated per check)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cpu_restatement as cr
from oracle import sasrec_oracle as so
from tools.gen_golden_inputs import make_batch, sample_idx

from test_oracle_golden import SMALL, close, load, weights

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def model(cfg, P, dropout=0.0, threads=0):
    m = cr.CpuSasrec(cfg.item_num, cfg.maxlen, cfg.hidden_units, cfg.num_heads, cfg.num_layers, dropout, threads)
    m.load_params(P)
    return m


@pytest.mark.parametrize("name", SMALL)
def test_forward_loss_grads_match_reference(golden_dir, name):
    z, cfg = load(golden_dir, name)
    m = model(cfg, weights(z))
    B, L, d, H, nl = z["seq"].shape[0], cfg.maxlen, cfg.hidden_units, cfg.num_heads, cfg.num_layers
    m.forward(z["seq"], z["dec"], z["pos"], z["neg"], training=True)      # dropout p == 0
    close(m.out(B, cr.WS_POS_LOGITS, 0, (B, L)), z["pos_logits"], 2e-5, what="pos_logits")
    close(m.out(B, cr.WS_NEG_LOGITS, 0, (B, L)), z["neg_logits"], 2e-5, what="neg_logits")
    for i in range(nl):
        close(m.out(B, cr.WS_ENC_X, i, (B, L, d)), z["enc_in.%d" % i], 2e-5, what="enc_in%d" % i)
        close(m.out(B, cr.WS_DEC_X, nl - i, (B, L, d)), z["dec_out.%d" % i], 2e-5, what="dec_out%d (reversed)" % i)
        if H > 1:
            close(m.out(B, cr.WS_REC, i, (B, L, H, H)), z["rec_ind.%d" % i], 2e-5, what="rec%d (reference row order)" % i)
    loss = m.loss_seed(list(z["lam1"]), list(z["lam2"]))
    m.backward()
    tn, wdterm = m.clip_adam(float(z["wd"]))
    assert abs(loss + wdterm - float(z["loss"])) < 2e-5 * max(1.0, abs(float(z["loss"])))
    G = m.grads()          # clip_adam left the un-clipped gradient incl. the weight-decay term in G
    for k, _ in so.param_shapes(cfg):
        if "gnone." + k in z.files:
            assert np.abs(G[k]).max() == 0.0, k
        else:
            close(G[k], z["g." + k], 2e-6, rtol=2e-4, what="grad " + k)
    assert abs(tn - float(z["total_norm"])) < 1e-5 * float(z["total_norm"]) + 1e-6


@pytest.mark.parametrize("name", SMALL)
def test_three_adam_steps_match_reference(golden_dir, name):
    z, cfg = load(golden_dir, name)
    m = model(cfg, weights(z))
    batch = (z["seq"], z["dec"], z["pos"], z["neg"])
    for step in range(3):
        loss, tn = m.train_step(batch, list(z["lam1"]), list(z["lam2"]), float(z["wd"]))
        if step in (0, 2):
            ref = weights(z, "w%d." % (step + 1))
            assert abs(loss - float(z["loss_step%d" % (step + 1)])) < 5e-5
            P = m.params()
            for k in ref:
                g = z["g." + k] if "g." + k in z.files else np.zeros_like(ref[k])
                noisy = np.abs(g) < 1e-6        # Adam turns rounding noise on exactly-zero gradients into +-lr (see test_oracle_golden)
                err = np.abs(P[k] - ref[k])
                assert err.max() <= (step + 1) * 1e-3 * 1.01 + 1e-6, k
                assert err[~noisy].max(initial=0.0) <= 2e-5, (k, step)


def test_cfga_slice_matches_reference(golden_dir):
    """cfg-A shape (L=200, d=64, H=2, 2 blocks, V=3416), B=8: norms + strided samples recorded from the reference."""
    z, cfg = load(golden_dir, "sasrec_cfga_b8")
    seed, B = int(z["seed"]), int(z["B"])
    P = so.init_params(cfg, seed=seed)
    batch = make_batch(np.random.RandomState(seed + 1), B, cfg.maxlen, cfg.item_num)
    m = model(cfg, P)
    m.forward(*batch, training=True)
    close(m.out(B, cr.WS_POS_LOGITS, 0, (B, cfg.maxlen)), z["pos_logits"], 2e-5, what="pos_logits")
    loss = m.loss_seed(list(z["lam1"]), list(z["lam2"]))
    m.backward()
    tn, wdterm = m.clip_adam(float(z["wd"]))
    assert abs(loss + wdterm - float(z["loss"])) < 1e-4
    assert abs(tn - float(z["total_norm"])) < 1e-3 * float(z["total_norm"])
    G = m.grads()
    for k, _ in so.param_shapes(cfg):
        if "gnone." + k in z.files:
            continue
        want = float(z["gnorm." + k])
        assert abs(float(np.linalg.norm(G[k])) - want) <= 2e-3 * want + 1e-7, k
        got = G[k].reshape(-1)[sample_idx(G[k].size)]
        scale = max(float(np.abs(z["gsample." + k]).max()), want / np.sqrt(max(G[k].size, 1)), 1e-9)
        assert np.abs(got - z["gsample." + k]).max() <= 1e-3 * scale + 1e-8, k


@pytest.mark.parametrize("H,L,B", [(2, 52, 5), (4, 48, 3), (1, 40, 4)])
def test_training_step_with_dropout_matches_numpy_oracle(H, L, B):
    cfg = so.Cfg(300, L, 64, H, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), B, L, cfg.item_num)
    lam1, lam2, wd, seed = [0.104292, 0.065892], [0.100833, 0.000607], 1e-3, 123457
    out = so.forward(P, cfg, *batch, training=True, seed=seed, b_offset=2)
    loss, parts, seeds = so.loss_and_seeds(P, cfg, out, batch[2], lam1, lam2, wd)
    G = so.backward(P, cfg, out[5], seeds, wd)
    m = model(cfg, P, dropout=0.5)
    m.forward(*batch, training=True, seed=seed, b_offset=2)
    for i in range(2):
        close(m.out(B, cr.WS_ENC_X, i, (B, L, 64)), out[2][i], 2e-5, what="enc_in%d" % i)
        close(m.out(B, cr.WS_DEC_X, 2 - i, (B, L, 64)), out[3][i], 2e-5, what="dec_out%d" % i)
    close(m.out(B, cr.WS_POS_LOGITS, 0, (B, L)), out[0], 2e-5, what="pos_logits")
    l2 = m.loss_seed(lam1, lam2)
    m.backward()
    tn, wdterm = m.clip_adam(wd)
    assert abs(l2 + wdterm - loss) < 2e-5 * max(1.0, abs(loss))
    Gc = m.grads()
    for k, _ in so.param_shapes(cfg):
        if G[k] is None:
            assert np.abs(Gc[k]).max() == 0.0, k
        else:
            close(Gc[k], G[k].reshape(Gc[k].shape), 3e-6, rtol=3e-4, what="grad " + k)
    assert abs(tn - so.grad_norm(G)) < 1e-5 * tn


def test_threads_and_shards_agree():
    """1 thread == all threads (thread-private accumulators folded in order; fp32 sums re-associate), and two shards with b_offset and global
    normalisers add up to the whole batch (the data-parallel contract)."""
    cfg = so.Cfg(120, 24, 64, 2, 2, dropout=0.5)
    P = so.init_params(cfg, seed=5)
    B = 6
    batch = make_batch(np.random.RandomState(6), B, cfg.maxlen, cfg.item_num)
    lam1, lam2 = [0.1, 0.05], [0.1, 0.01]
    res = []
    for th in (1, 0):
        m = model(cfg, P, dropout=0.5, threads=th)
        m.forward(*batch, training=True, seed=9)
        m.loss_seed(lam1, lam2)
        m.backward()
        res.append(m.G.copy())
    cr.load().adt_cpu_set_threads(0)
    assert np.abs(res[0] - res[1]).max() <= 2e-6 * max(1.0, np.abs(res[0]).max())
    norms = (float(np.count_nonzero(batch[2])), float(B * cfg.maxlen * 64), float(B * cfg.maxlen * 2))
    tot = np.zeros_like(res[0])
    for lo, hi in ((0, 4), (4, 6)):
        m = model(cfg, P, dropout=0.5)
        m.forward(*[a[lo:hi] for a in batch], training=True, seed=9, b_offset=lo)
        m.loss_seed(lam1, lam2, norms)
        m.backward()
        tot += m.G
    assert np.abs(tot - res[1]).max() <= 3e-6 * max(1.0, np.abs(res[1]).max())


def test_predict_matches_reference(golden_dir):
    z, cfg = load(golden_dir, "sasrec_small_h4")
    m = model(cfg, weights(z))
    close(m.predict(z["seq"], z["cand"]), z["predict_cand"], 2e-5, what="predict cand")
    close(m.predict(z["seq"]), z["predict_full"], 2e-5, what="predict full")


def test_restatement_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """SURVEY 5 (race detection / sanitizers): the same source built with -fsanitize=address,undefined runs a dropout-on training step
    (forward, loss seeds, backward, clip + Adam, predict) in a child process with the sanitizer runtimes preloaded; any report fails it."""
    lib = str(tmp_path / "libadt_cpu_san.so")
    try:
        cr.build(force=True, out=lib, extra=["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-O1"])
    except subprocess.CalledProcessError:
        pytest.skip("the sanitizer runtimes are not installed with this g++")
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["g++", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import cpu_restatement as cr, sasrec_oracle as so\n"
        "cr.LIB = %r\n"
        "cr.build = lambda *a, **k: cr.LIB\n"
        "from tools.gen_golden_inputs import make_batch\n"
        "cfg = so.Cfg(90, 20, 64, 2, 2, dropout=0.5)\n"
        "m = cr.CpuSasrec(90, 20, 64, 2, 2, 0.5, threads=3)\n"
        "m.load_params(so.init_params(cfg, 1))\n"
        "b = make_batch(np.random.RandomState(2), 5, 20, 90)\n"
        "for s in range(2): print(m.train_step(b, [0.1, 0.05], [0.1, 0.01], 1e-3, seed=7 + s))\n"
        "print(m.predict(b[0], b[2][:, :7]).shape)\n" % (REPO, lib))
    env = dict(os.environ, LD_PRELOAD="%s:%s" % (asan, ubsan), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               OMP_NUM_THREADS="3")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    assert "(5, 7)" in out.stdout
