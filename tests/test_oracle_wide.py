"""CPU: the BERT4Rec-ADT and STOSA-ADT oracles (oracle/bert_oracle.py, oracle/stosa_oracle.py) against the golden vectors
recorded from the imported reference (tools/gen_golden_wide.py).  fp32 tolerances are written next to each check."""
import os

import numpy as np
import pytest

from oracle import bert_oracle as bo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    assert err < tol, "%s: rel err %.3g (tol %.1g)" % (what, err, tol)


def _adam_close(got, want, lr, nsteps, what, gref=None):
    """Adam moves every entry by about lr per step, by lr * g / (|g| + 1e-8) on the first: entries whose gradient is of the
    order of eps move by a rounding-dependent fraction of lr, so post-step weights are compared in units of lr (a wrong
    bias correction, beta or weight-decay coupling is off by >= 0.1 lr on most entries)."""
    diff = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    if gref is not None:      # first step: entries with |g| >> eps must agree tightly; eps-dominated ones within one lr
        big = np.abs(gref) > 1e-6
        assert (diff[big].max() if big.any() else 0.0) < 0.02 * lr, "weights after 1 step: %s: %.3g lr" % (what, diff[big].max() / lr)
        assert diff.max() < lr, what
    else:
        assert diff.max() < 0.3 * lr, "weights after %d steps: %s: abs err %.3g lr" % (nsteps, what, diff.max() / lr)


def _bert_case(tag):
    g = np.load(os.path.join(GOLD, "bert_%s.npz" % tag))
    V, L, d, H, nl, inner = [int(x) for x in g["cfg"]]
    cfg = bo.Cfg(V, L, d, H, nl, inner)
    P = bo.init_params(cfg, int(g["seed"]))
    r = np.random.RandomState(int(g["seed"]) + 1)
    for k in P:   # same perturbation as tools/gen_golden_wide.py:gen_bert
        if k.endswith("head_classifier.bias") or k == "mask_bias" or (k.endswith(".bias") and "layer_norm" not in k):
            P[k] = (0.02 * r.standard_normal(P[k].shape)).astype(np.float32)
    return g, cfg, P


@pytest.mark.parametrize("tag", ["small", "h4", "hd64"])
def test_bert_forward_matches_reference(tag):
    g, cfg, P = _bert_case(tag)
    logits, enc_in, dec_out, rec = bo.forward(P, cfg, g["src"], g["dec"], training=False)
    _close(logits, g["logits"], 2e-5, "logits")
    for i in range(cfg.num_layers):
        _close(enc_in[i], g["enc_in_%d" % i], 2e-5, "enc_in %d" % i)
        _close(dec_out[i], g["dec_out_%d" % i], 2e-5, "dec_out %d" % i)
        _close(rec[i], g["rec_%d" % i], 2e-5, "rec %d" % i)
    _close(bo.predict(P, cfg, g["src"], g["cand"]), g["predict"], 2e-5, "predict")


@pytest.mark.parametrize("tag", ["small", "h4", "hd64"])
def test_bert_train_step_matches_reference(tag):
    g, cfg, P = _bert_case(tag)
    lam1, lam2 = list(g["lambda1"]), list(g["lambda2"])
    loss, parts, G = bo.loss_and_grads(P, cfg, g["src"], g["dec"], g["labels"], lam1, lam2, training=True, seed=0)
    assert abs(loss - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    for k in P:
        _close(G[k], g["grad." + k], 5e-5, "grad " + k)
    state = {}
    Pw = {k: v.copy() for k, v in P.items()}
    for step in range(3):
        _, tn = bo.train_step(Pw, cfg, state, g["src"], g["dec"], g["labels"], lam1, lam2, lr=float(g["lr"]),
                              weight_decay=float(g["wd"]), clip=float(g["clip"]), training=True, seed=0)
        if step == 0:
            assert abs(tn - float(g["grad_norm"])) < 2e-5 * float(g["grad_norm"])
        if step in (0, 2) and ("w%d.mask_bias" % (step + 1)) in g.files:
            for k in P:
                # Adam divides by |g| + 1e-8: entries with |g| ~ 1e-8 move by a rounding-dependent fraction of lr = 1e-3
                _adam_close(Pw[k], g["w%d." % (step + 1) + k], float(g["lr"]), step + 1, k, g["grad." + k] if step == 0 and ("grad." + k) in g.files else None)


# ---- STOSA-ADT ----------------------------------------------------------------------------------------------------------
from oracle import stosa_oracle as so  # noqa: E402


def _stosa_case(tag):
    g = np.load(os.path.join(GOLD, "stosa_%s.npz" % tag))
    V, L, d, H, nl, nu = [int(x) for x in g["cfg"]]
    cfg = so.Cfg(V, L, d, H, nl, num_users=nu, pvn_weight=float(g["pvn_weight"]))
    P = so.init_params(cfg, int(g["seed"]))
    r = np.random.RandomState(int(g["seed"]) + 1)
    for k in P:   # same perturbation as tools/gen_golden_stosa.py:gen_stosa
        if k.endswith(".bias") and "LayerNorm" not in k:
            P[k] = (0.02 * r.standard_normal(P[k].shape)).astype(np.float32)
    return g, cfg, P


@pytest.mark.parametrize("tag", ["small", "l2h2", "h1"])
def test_stosa_finetune_matches_reference(tag):
    g, cfg, P = _stosa_case(tag)
    m, c, enc_in, enc_rec, dec_out = so.finetune(P, cfg, g["input_ids"], g["dec_ids"], training=False)
    _close(m, g["mean_out"], 2e-5, "mean_out")
    _close(c, g["cov_out"], 2e-5, "cov_out")
    for i in range(cfg.num_layers):
        _close(enc_in[i][0], g["enc_in_mean_%d" % i], 2e-5, "enc_in mean")
        _close(enc_in[i][1], g["enc_in_cov_%d" % i], 2e-5, "enc_in cov")
        _close(enc_rec[i][0], g["rec_mean_%d" % i], 2e-5, "rec mean")
        _close(enc_rec[i][1], g["rec_cov_%d" % i], 2e-5, "rec cov")
        _close(dec_out[i][0], g["dec_out_mean_%d" % i], 2e-5, "dec_out mean")
        _close(dec_out[i][1], g["dec_out_cov_%d" % i], 2e-5, "dec_out cov")
    _close(so.predict_full(P, cfg, g["input_ids"], g["dec_ids"]), g["full_dist"], 2e-5, "full-sort distances")


@pytest.mark.parametrize("tag", ["small", "l2h2", "h1"])
def test_stosa_train_step_matches_reference(tag):
    g, cfg, P = _stosa_case(tag)
    lam1, lam2 = list(g["lambda1"]), list(g["lambda2"])
    loss, parts, G = so.loss_and_grads(P, cfg, g["input_ids"], g["dec_ids"], g["pos_ids"], g["neg_ids"], lam1, lam2, training=True, seed=0)
    assert abs(loss - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert abs(parts["bpr"] - float(g["bpr"])) < 2e-5 * abs(float(g["bpr"])) and abs(parts["auc"] - float(g["auc"])) < 1e-6
    assert abs(parts["pvn"] - float(g["pvn"])) < 2e-5 * max(abs(float(g["pvn"])), 1e-6)
    assert sorted(k for k in P if G[k] is None) == sorted(str(x) for x in g["grad_none"])
    for k in P:
        if G[k] is not None:
            _close(G[k], g["grad." + k], 1e-4, "grad " + k)
    state = {}
    Pw = {k: v.copy() for k, v in P.items()}
    for step in range(3):
        so.train_step(Pw, cfg, state, g["input_ids"], g["dec_ids"], g["pos_ids"], g["neg_ids"], lam1, lam2, lr=float(g["lr"]), training=True, seed=0)
        if step in (0, 2) and ("w%d.LayerNorm.weight" % (step + 1)) in g.files:
            for k in P:
                _adam_close(Pw[k], g["w%d." % (step + 1) + k], float(g["lr"]), step + 1, k, g["grad." + k] if step == 0 and ("grad." + k) in g.files else None)


# ---- SASRec-ADT supernet ----------------------------------------------------------------------------------------------------
from oracle import super_oracle as su  # noqa: E402


def _super_case(tag):
    g = np.load(os.path.join(GOLD, "super_%s.npz" % tag))
    V, L, d, H, nl = [int(x) for x in g["cfg"]]
    cfg = su.Cfg(V, L, d, H, nl, g["rec_choice"], g["ind_choice"])
    return g, cfg, su.init_params(cfg, int(g["seed"]))


@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_supernet_matches_reference(tag):
    g, cfg, P = _super_case(tag)
    cand = [float(x) for x in g["cand"]]
    block, rec_w, ind_w = su.cand_to_block(cfg, cand)
    shared = su.get_shared(cfg, block)
    assert [list(s[0]) for s in shared] == g["shared_idx"].tolist()
    np.testing.assert_allclose(np.array([s[1] for s in shared]), g["shared_weights"], rtol=1e-12)
    pl, nl, ei, do, rc = su.forward(P, cfg, block, g["seq"], g["dec"], g["pos"], g["neg"])
    _close(pl, g["pos_logits"], 2e-5, "pos_logits")
    _close(nl, g["neg_logits"], 2e-5, "neg_logits")
    for i in range(cfg.num_layers):
        _close(ei[i], g["enc_in_%d" % i], 2e-5, "enc_in")
        _close(do[i], g["dec_out_%d" % i], 2e-5, "dec_out")
        # the reference's rec rows are permuted by the (L,B,E)->(B,L,..) view (sasrec/modules.py:518): compare row-sorted
        a, b = rc[i].reshape(-1, cfg.num_heads ** 2), g["rec_%d" % i].reshape(-1, cfg.num_heads ** 2)
        _close(a[np.lexsort(a.T)], b[np.lexsort(b.T)], 2e-5, "rec")
    _close(su.predict(P, cfg, block, g["seq"], g["items"]), g["predict"], 2e-5, "predict")
    loss, G = su.loss_and_grads(P, cfg, cand, g["seq"], g["dec"], g["pos"], g["neg"], training=True, seed=0)
    assert abs(loss - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    none = set(str(x) for x in g["grad_none"])
    assert set(k for k in P if G[k] is None) == none
    for k in P:
        if G[k] is not None:
            _close(G[k], g["grad." + k], 1e-4, "grad " + k)
    state = {}
    Pw = {k: v.copy() for k, v in P.items()}
    _, tn = su.train_step(Pw, cfg, state, cand, g["seq"], g["dec"], g["pos"], g["neg"], lr=float(g["lr"]), weight_decay=float(g["wd"]),
                          clip=float(g["clip"]), training=True, seed=0)
    assert abs(tn - float(g["grad_norm"])) < 2e-5 * float(g["grad_norm"])
    for k in [f[3:] for f in g.files if f.startswith("w1.")]:
        _adam_close(Pw[k], g["w1." + k], float(g["lr"]), 1, k, g["grad." + k])
    for k in none:
        assert np.array_equal(Pw[k], P[k])      # grad None: untouched, not even by the weight decay


# ---- the oracles at the BASELINE configurations' own shapes (compacted fixtures: norms + strided samples) ------------------
from tools.gen_golden_inputs import golden_err  # noqa: E402


def test_stosa_oracle_at_beauty_shape():
    """configs[4]: item_size 12,103, L=100, H=4, d=64, 1 layer, B=8 (tests/golden/stosa_cfg5_beauty.npz)."""
    g, cfg, P = _stosa_case("cfg5_beauty")
    m, c, enc_in, enc_rec, dec_out = so.finetune(P, cfg, g["input_ids"], g["dec_ids"], training=False)
    assert golden_err(m, g, "mean_out") < 2e-5 and golden_err(c, g, "cov_out") < 2e-5
    assert golden_err(dec_out[0][0], g, "dec_out_mean_0") < 2e-5 and golden_err(enc_rec[0][1], g, "rec_cov_0") < 2e-5
    assert golden_err(so.predict_full(P, cfg, g["input_ids"], g["dec_ids"]), g, "full_dist") < 2e-5
    lam1, lam2 = list(g["lambda1"]), list(g["lambda2"])
    loss, parts, G = so.loss_and_grads(P, cfg, g["input_ids"], g["dec_ids"], g["pos_ids"], g["neg_ids"], lam1, lam2, training=True, seed=0)
    assert abs(loss - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert sorted(k for k in P if G[k] is None) == sorted(str(x) for x in g["grad_none"])
    for k in P:
        if G[k] is not None:
            assert golden_err(G[k], g, "grad." + k) < 2e-4, k


def test_bert_oracle_at_ml20m_shape():
    """configs[2]: vocabulary 26,844, d=256, H=4, inner=1024, L=200, 2 layers (tests/golden/bert_cfg3_ml20m.npz); the first
    two sequences of the recorded batch (the oracle is numpy: the full B=4 backward takes minutes)."""
    g, cfg, P = _bert_case("cfg3_ml20m")
    logits, enc_in, dec_out, rec = bo.forward(P, cfg, g["src"][:1], g["dec"][:1], training=False)
    # batch rows are independent in eval mode: row 0 of the recorded tensors is every sample with flat index < row size
    for got, key in ((logits, "logits"), (enc_in[1], "enc_in_1"), (dec_out[0], "dec_out_0"), (rec[1], "rec_1")):
        n_total = 4 * got.size
        idx = (np.arange(1024, dtype=np.int64) * 7919) % n_total
        keep = idx < got.size
        assert keep.sum() > 100, key
        want = g[key + "@sample"][keep].astype(np.float64)
        err = np.abs(got.reshape(-1)[idx[keep]].astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-6)
        assert err < 5e-5, (key, err)
