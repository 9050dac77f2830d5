"""GPU parity of the SASRec-ADT supernet (adt_amd/sasrec/supersasrec.py, through the C ABI) against the golden tensors
recorded from the imported reference (SuperSASRecModel + the _train_warmup loop body, dropout 0) and against the numpy
oracle with dropout ON.  Tolerances: exact-fp32 MFMA mode 1e-4 (activations) / 5e-4 (gradients) of the tensor magnitude;
bf16-operand mode 3e-2 on activations."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import super_oracle as su  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Args:
    pass


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)


def load_case(tag, dropout=0.0):
    g = np.load(os.path.join(GOLD, "super_%s.npz" % tag))
    V, L, d, H, nl = [int(x) for x in g["cfg"]]
    cfg = su.Cfg(V, L, d, H, nl, g["rec_choice"], g["ind_choice"], dropout)
    return g, cfg, su.init_params(cfg, int(g["seed"]))


def build(cfg, P, prec):
    from adt_amd.sasrec.supersasrec import SuperSASRecModel
    a = Args()
    a.device, a.num_heads, a.maxlen, a.num_layers, a.hidden_units, a.dropout, a.precision = "cuda:0", cfg.num_heads, cfg.maxlen, cfg.num_layers, cfg.hidden_units, cfg.dropout, prec
    m = SuperSASRecModel(1, cfg.item_num, cfg.rec_choice, cfg.ind_choice, a)
    m.load_numpy(P)
    return m


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_forward_and_predict_match_reference(tag, prec):
    from adt_amd.sasrec.supersasrec import SuperTrainer
    g, cfg, P = load_case(tag)
    m = build(cfg, P, prec)
    tr = SuperTrainer(m)
    tr.set_choice([float(x) for x in g["cand"]])
    assert [list(s[0]) for s in m.shared] == g["shared_idx"].tolist()
    m.eval()
    pl, nl, ei, do, rc = m(None, g["seq"], g["dec"], g["pos"], g["neg"])
    tol = 1e-4 if prec == "f32" else 3e-2
    assert rel(pl.cpu().numpy(), g["pos_logits"]) < tol and rel(nl.cpu().numpy(), g["neg_logits"]) < tol
    H2 = cfg.num_heads ** 2
    for i in range(cfg.num_layers):
        assert rel(ei[i].cpu().numpy(), g["enc_in_%d" % i]) < tol and rel(do[i].cpu().numpy(), g["dec_out_%d" % i]) < tol
        a, b = rc[i].cpu().numpy().reshape(-1, H2), g["rec_%d" % i].reshape(-1, H2)     # reference rows are permuted (modules.py:518)
        if prec == "f32":
            assert rel(a[np.lexsort(a.T)], b[np.lexsort(b.T)]) < tol
        else:
            assert abs(a.mean() - b.mean()) < tol
    assert rel(m.predict(None, g["seq"], g["items"]).cpu().numpy(), g["predict"]) < tol
    sd = m.state_dict()
    assert set(sd) == set(P) and all(tuple(sd[k].shape) == P[k].shape for k in P)


@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_warmup_step_matches_reference_fp32(tag):
    from adt_amd.sasrec.supersasrec import SuperTrainer
    g, cfg, P = load_case(tag)
    m = build(cfg, P, "f32")
    tr = SuperTrainer(m, lr=float(g["lr"]), weight_decay=float(g["wd"]), clip=float(g["clip"]))
    tr.set_choice([float(x) for x in g["cand"]])
    tr.step(g["seq"], g["dec"], g["pos"], g["neg"])
    torch.cuda.synchronize()
    assert abs(float(tr.loss()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert abs(float(tr.grad_norm()) - float(g["grad_norm"])) < 3e-4 * float(g["grad_norm"])
    none = set(str(x) for x in g["grad_none"])
    gmax = max(float(np.abs(g["grad." + k]).max()) for k in P if k not in none)
    lr = float(g["lr"])
    for k in P:
        got_g = m.G(k).cpu().numpy()
        if k in none:
            assert np.all(got_g == 0.0), k
            assert np.array_equal(m.P(k).cpu().numpy(), P[k]), k      # untouched: no decay, no step
            continue
        want = g["grad." + k]
        assert np.abs(got_g - want).max() < 5e-4 * max(np.abs(want).max(), 1e-3 * gmax), k
        if ("w1." + k) in g.files:
            diff = np.abs(m.P(k).cpu().numpy().astype(np.float64) - g["w1." + k])
            big = np.abs(want) > 1e-5
            assert (diff[big].max() if big.any() else 0.0) < 0.05 * lr, k
            assert diff.max() < 1.01 * lr, k


def test_warmup_step_with_dropout_matches_oracle():
    g, cfg, P = load_case("l2", dropout=0.3)
    from adt_amd.sasrec.supersasrec import SuperTrainer
    m = build(cfg, P, "f32")
    tr = SuperTrainer(m, lr=1e-3, weight_decay=1e-4)
    cand = [float(x) for x in g["cand"]]
    tr.set_choice(cand)
    tr.step(g["seq"], g["dec"], g["pos"], g["neg"])
    torch.cuda.synchronize()
    seed = int(m._seed.cpu().numpy().view(np.uint32)[0])
    loss, G = su.loss_and_grads(P, cfg, cand, g["seq"], g["dec"], g["pos"], g["neg"], training=True, seed=seed)
    assert abs(float(tr.loss()) - loss) < 1e-4 * abs(loss)
    gmax = max(float(np.abs(v).max()) for v in G.values() if v is not None)
    for k in P:
        if G[k] is not None:
            assert np.abs(m.G(k).cpu().numpy() - G[k]).max() < 5e-4 * max(np.abs(G[k]).max(), 1e-3 * gmax), k


@pytest.mark.parametrize("dropout", [0.0, 0.3])
@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_grouped_candidate_layers_match_stage_kernels(tag, dropout, monkeypatch):
    """bf16 mode runs the four selected candidate layers of a depth on the per-sequence fused layer kernels (mixing weight as output
    epilogue / gradient prologue, adt_seq_{enc,dec}_layer_{fwd,bwd}); ADT_SUPER_FUSED=0 keeps one stage-kernel sequence per candidate.
    Same weights, batch and dropout seed: loss within 1e-2 relative, whole gradient within 6 % relative Frobenius (two bf16 roundings of
    the same arithmetic), every tensor the reference leaves at grad None exactly zero in both."""
    from adt_amd.sasrec.supersasrec import SuperTrainer
    g, cfg, P = load_case(tag, dropout)
    res = []
    for fused in ("1", "0"):
        monkeypatch.setenv("ADT_SUPER_FUSED", fused)
        m = build(cfg, P, "bf16")
        assert m.fused_layers() == (fused == "1" and cfg.maxlen % 4 == 0)
        tr = SuperTrainer(m, lr=float(g["lr"]), weight_decay=float(g["wd"]), clip=float(g["clip"]), seed=7)
        tr.set_choice([float(x) for x in g["cand"]])
        m.train()
        ids = tuple(m.ids(g[k]) for k in ("seq", "dec", "pos", "neg"))
        B, L = ids[0].shape
        norms = torch.tensor([float(np.count_nonzero(g["pos"])), float(B * L * cfg.hidden_units), float(B * L * cfg.num_heads)], device=m.dev)
        tr.loss_slots.zero_()
        m.flat_grad.zero_()
        m.loss_forward_backward(ids, tr.rec_weights, tr.ind_weights, norms, tr.loss_slots)
        torch.cuda.synchronize()
        res.append((float(tr.loss()), m.flat_grad.clone(), m))
    (l1, g1, m1), (l0, g0, m0) = res
    if not m1.fused_layers():
        pytest.skip("maxlen %d: the fused layer kernels need L %% 4 == 0" % cfg.maxlen)
    assert abs(l1 - l0) < 1e-2 * abs(l0)
    assert float((g1 - g0).norm()) < 0.06 * float(g0.norm())
    for name in [str(x) for x in g["grad_none"]]:
        assert float(m1.G(name).abs().max()) == 0.0, name
