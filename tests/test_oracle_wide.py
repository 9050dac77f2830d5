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
                _close(Pw[k], g["w%d." % (step + 1) + k], 3e-4 if step else 1e-4, "weights after %d: %s" % (step + 1, k))
