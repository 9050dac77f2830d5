"""GPU parity of the BERT4Rec-ADT and STOSA-ADT supernets (adt_amd/bert4rec/superbert.py, adt_amd/stosa/supernet.py, through the C ABI)
against golden tensors recorded from the imported reference supernets (tools/gen_golden_super_wide.py: forward, predict / full-sort
distances under two block choices, the warm-up loss, every gradient, one optimizer step; dropout 0), and of the BATCHED candidate
evaluation (adt_amd/supersearch.candidate_features) against one-candidate-at-a-time evaluation for all three supernets.

Tolerances: exact-fp32 MFMA mode 2e-4 of the tensor magnitude on activations, 1e-3 on gradients (sums over B*L tokens in another
order; fixtures hold norms + strided samples for large tensors); bf16-operand mode 4e-2 on activations."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tools.gen_golden_inputs import golden_err, seeded_params  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = 96      # samples per compacted tensor (tools/gen_golden_super_wide.py)


class Args:
    pass


def err(got, g, key):
    return golden_err(got.detach().cpu().numpy() if hasattr(got, "detach") else got, g, key, K)


def build_bert(g, prec):
    from adt_amd.bert4rec.superbert import SuperBertModel
    V, L, d, H, nl = [int(x) for x in g["cfg"]]
    a = Args()
    a.device, a.maxlen, a.num_heads, a.num_layers, a.hidden_units, a.inner_units = "cuda:0", L, H, nl, d, 4 * d
    a.dropout, a.attention_dropout, a.type_vocab_size, a.precision = 0.0, 0.0, 2, prec
    m = SuperBertModel(1, V, g["rec_choice"], g["ind_choice"], a)
    m.load_numpy(seeded_params({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"])))
    return m


def build_stosa(g, prec):
    from adt_amd.stosa.supernet import DisenDistSASupernet
    V, L, d, H, nl, nu = [int(x) for x in g["cfg"]]
    a = Args()
    a.device, a.item_size, a.maxlen, a.hidden_units, a.num_heads, a.num_layers, a.num_users = "cuda:0", V, L, d, H, nl, nu
    a.dropout, a.attention_dropout, a.pvn_weight, a.precision, a.distance_metric = 0.0, 0.0, float(g["pvn_weight"]), prec, "wasserstein"
    m = DisenDistSASupernet(a, g["rec_choice"], g["ind_choice"])
    m.load_numpy(seeded_params({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"])))
    return m


def ranks_from_scores(s):
    """evaluate_loader's rank of column 0 (number of candidates scored strictly above it)."""
    return (s[:, 1:] > s[:, :1]).sum(1)


# ---- BERT4Rec-ADT supernet --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_superbert_forward_and_predict(tag, prec):
    from adt_amd.bert4rec.superbert import SuperBertTrainer
    g = np.load(os.path.join(GOLD, "superbert_%s.npz" % tag))
    m = build_bert(g, prec)
    tr = SuperBertTrainer(m)
    tr.set_choice([float(x) for x in g["cand"]])
    assert [list(s[0]) for s in m.shared] == g["shared_idx"].tolist()
    assert np.allclose([list(s[1]) for s in m.shared], g["shared_weights"], atol=1e-12)
    m.eval()
    logits, ei, do, rc = m(g["src"], g["dec"])
    tol = 2e-4 if prec == "f32" else 4e-2
    assert err(logits, g, "logits") < tol
    nl = int(g["cfg"][4])
    for i in range(nl):
        assert err(ei[i], g, "enc_in_%d" % i) < tol and err(do[i], g, "dec_out_%d" % i) < tol and err(rc[i], g, "rec_%d" % i) < tol
    assert err(m.predict(None, g["src"], None, None, g["items"]), g, "predict") < tol
    # the reference's state_dict names and shapes
    assert "encoder.encoder_layers.0.1.head_classifier.weight" in m.state_dict() and m.vocab == int(g["cfg"][0]) + 2


@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_superbert_warmup_step_fp32(tag):
    from adt_amd.bert4rec.superbert import SuperBertTrainer
    g = np.load(os.path.join(GOLD, "superbert_%s.npz" % tag))
    m = build_bert(g, "f32")
    tr = SuperBertTrainer(m, lr=float(g["lr"]), weight_decay=float(g["wd"]), clip=float(g["clip"]))
    tr.set_choice([float(x) for x in g["cand"]])
    tr.step(g["src"], g["dec"], g["labels"])
    torch.cuda.synchronize()
    assert abs(float(tr.loss()) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
    assert abs(float(tr.grad_norm()) - float(g["grad_norm"])) < 5e-4 * float(g["grad_norm"])
    none = set(str(x) for x in g["grad_none"])
    checked = 0
    for name, _ in m.table:
        key = "grad." + name
        if key in g.files or key + "@norm" in g.files:
            assert err(m.G(name), g, key) < 1e-3, name
            checked += 1
        elif name in none:
            assert float(m.G(name).abs().max()) == 0.0, name        # the reference leaves these at grad None
    assert checked > 20
    for key in [k for k in g.files if k.startswith("w1.")]:
        name = key[3:].replace("@norm", "").replace("@sample", "")
        if name.endswith("key_transfer.bias"):
            continue      # softmax is invariant to a key bias: its gradient is rounding noise, which Adam normalises to +-lr steps
        assert err(m.P(name), g, "w1." + name) < 2e-4, name


# ---- STOSA-ADT supernet -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_superstosa_forward_and_full_sort(tag, prec):
    from adt_amd.stosa.supernet import SuperStosaTrainer
    g = np.load(os.path.join(GOLD, "superstosa_%s.npz" % tag))
    m = build_stosa(g, prec)
    tr = SuperStosaTrainer(m)
    tr.set_choice([float(x) for x in g["cand"]])
    assert [list(s[0]) for s in m.shared] == g["shared_idx"].tolist()
    m.eval()
    mo, co, _, _, ei, er, do = m.finetune(g["input_ids"], g["dec_ids"], None)
    tol = 2e-4 if prec == "f32" else 4e-2
    assert err(mo, g, "mean_out") < tol and err(co, g, "cov_out") < tol
    nl = int(g["cfg"][4])
    for i in range(nl):
        assert err(ei[i][0], g, "enc_in_mean_%d" % i) < tol and err(ei[i][1], g, "enc_in_cov_%d" % i) < tol
        assert err(er[i][0], g, "rec_mean_%d" % i) < tol and err(er[i][1], g, "rec_cov_%d" % i) < tol
        assert err(do[i][0], g, "dec_out_mean_%d" % i) < tol and err(do[i][1], g, "dec_out_cov_%d" % i) < tol
    assert err(m.predict_full(g["input_ids"]), g, "full_dist") < (5e-4 if prec == "f32" else 6e-2)


@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_superstosa_warmup_step_fp32(tag):
    from adt_amd.stosa.supernet import SuperStosaTrainer
    g = np.load(os.path.join(GOLD, "superstosa_%s.npz" % tag))
    m = build_stosa(g, "f32")
    tr = SuperStosaTrainer(m, lr=float(g["lr"]))
    tr.set_choice([float(x) for x in g["cand"]])
    tr.step(g["input_ids"], g["dec_ids"], g["pos_ids"], g["neg_ids"])
    torch.cuda.synchronize()
    assert abs(float(tr.loss()) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
    assert abs(float(tr.grad_norm()) - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    none = set(str(x) for x in g["grad_none"])
    checked = 0
    for name, _ in m.table:
        key = "grad." + name
        if key in g.files or key + "@norm" in g.files:
            assert err(m.G(name), g, key) < 2e-3, name
            checked += 1
        elif name in none:
            assert float(m.G(name).abs().max()) == 0.0, name
    assert checked > 20
    for key in [k for k in g.files if k.startswith("w1.")]:
        name = key[3:].replace("@norm", "").replace("@sample", "")
        assert err(m.P(name), g, "w1." + name) < 2e-4, name


# ---- batched candidate evaluation ---------------------------------------------------------------------------------------------------
def _cands(nl, n, seed):
    r = np.random.RandomState(seed)
    return [[float(x) for x in r.rand(2 * nl)] for _ in range(n)]


@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_batched_candidates_superbert(tag):
    from adt_amd.supersearch import cand_to_block, get_shared
    g = np.load(os.path.join(GOLD, "superbert_%s.npz" % tag))
    m = build_bert(g, "f32")
    rc, ic, nl = g["rec_choice"], g["ind_choice"], int(g["cfg"][4])
    cands = [[float(x) for x in g["cand"]], [float(x) for x in g["cand2"]]] + _cands(nl, 5, 3)
    shared = [get_shared(rc, ic, cand_to_block(rc, ic, c)[0]) for c in cands]
    stats = {}
    ranks = m.predict_rank_candidates(g["src"], g["items"], shared, stats=stats).cpu().numpy()
    # the two golden block choices: ranks from the reference's scores
    assert (ranks[0] == ranks_from_scores(g["predict"])).all() and (ranks[1] == ranks_from_scores(g["predict2"])).all()
    for p, c in enumerate(cands):       # every candidate: identical to selecting it alone
        m.set_choice(cand_to_block(rc, ic, c)[0])
        _, r1 = m.predict(None, g["src"], None, None, g["items"], want_rank=True)
        assert (ranks[p] == r1.cpu().numpy()).all(), p
    assert stats["layer_calls"] < 4 * nl * len(cands)       # layers were shared / stacked, not run per (candidate, layer)


@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_batched_candidates_superstosa(tag):
    from adt_amd.supersearch import cand_to_block, get_shared
    g = np.load(os.path.join(GOLD, "superstosa_%s.npz" % tag))
    m = build_stosa(g, "f32")
    rc, ic, nl = g["rec_choice"], g["ind_choice"], int(g["cfg"][4])
    cands = [[float(x) for x in g["cand"]], [float(x) for x in g["cand2"]]] + _cands(nl, 4, 5)
    shared = [get_shared(rc, ic, cand_to_block(rc, ic, c)[0]) for c in cands]
    stats = {}
    dist = m.predict_full_candidates(g["input_ids"], shared, stats=stats).cpu().numpy()
    B = g["input_ids"].shape[0]
    assert golden_err(dist[:B], g, "full_dist", K) < 5e-4 and golden_err(dist[B:2 * B], g, "full_dist2", K) < 5e-4
    for p, c in enumerate(cands):
        m.set_choice(cand_to_block(rc, ic, c)[0])
        one = m.predict_full(g["input_ids"]).cpu().numpy()
        assert np.abs(dist[p * B:(p + 1) * B] - one).max() <= 1e-5 * np.abs(one).max(), p
    assert stats["layer_calls"] < 4 * nl * len(cands)


@pytest.mark.parametrize("tag", ["c3", "l2"])
def test_batched_candidates_supersasrec(tag):
    from adt_amd.sasrec.supersasrec import SuperSASRecModel
    from adt_amd.supersearch import cand_to_block, get_shared
    from oracle import super_oracle as su
    g = np.load(os.path.join(GOLD, "super_%s.npz" % tag))
    V, L, d, H, nl = [int(x) for x in g["cfg"]]
    cfg = su.Cfg(V, L, d, H, nl, g["rec_choice"], g["ind_choice"], 0.0)
    a = Args()
    a.device, a.num_heads, a.maxlen, a.num_layers, a.hidden_units, a.dropout, a.precision = "cuda:0", H, L, nl, d, 0.0, "f32"
    m = SuperSASRecModel(1, V, cfg.rec_choice, cfg.ind_choice, a)
    m.load_numpy(su.init_params(cfg, int(g["seed"])))
    rc, ic = g["rec_choice"], g["ind_choice"]
    cands = [[float(x) for x in g["cand"]]] + _cands(nl, 6, 7)
    shared = [get_shared(rc, ic, cand_to_block(rc, ic, c)[0]) for c in cands]
    stats = {}
    ranks = m.predict_rank_candidates(g["seq"], g["items"], shared, stats=stats).cpu().numpy()
    assert (ranks[0] == ranks_from_scores(g["predict"])).all()
    for p, c in enumerate(cands):
        m.set_choice(cand_to_block(rc, ic, c)[0])
        _, r1 = m.predict_rank(g["seq"], g["items"])
        assert (ranks[p] == r1.cpu().numpy()).all(), p
    assert stats["layer_calls"] < 4 * nl * len(cands)
