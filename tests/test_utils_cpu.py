"""CPU: host-side mirror of the reference's config/data/metric surface (adt_amd/sasrec/utils.py) against
known answers recorded from the reference (tests/golden/{config,data,metrics}_kat.npz)."""
import os

import numpy as np

from adt_amd.sasrec import utils as U


def test_get_lambdas_and_candidate_interpolation(golden_dir):
    z = np.load(os.path.join(golden_dir, "config_kat.npz"))
    for ds in ("ml-1m", "beauty", "Beauty", "steam", "ml-20m"):
        l1, l2 = U.get_lambdas(ds)
        assert l1 == list(z["lam1." + ds]) and l2 == list(z["lam2." + ds])
    assert U.get_lambdas("nope") is None
    got = [U.candidate_to_lambda(list(z["c2l.choices"]), c) for c in z["c2l.cand"]]
    assert np.allclose(got, z["c2l.out"], rtol=0, atol=1e-15)


def test_set_template_overrides_cli():
    class A:
        dataset = "ml-1m"
        hidden_units = 50
        maxlen = 50
    a = U.set_template(A())
    assert a.hidden_units == 256 and a.maxlen == 200 and a.num_heads == 2 and a.weight_decay == 0.001
    a = U.set_template(type("B", (), {"dataset": "beauty"})())
    assert a.maxlen == 50 and a.weight_decay == 0.0001


def test_data_partition_and_samplers(golden_dir, tmp_path):
    z = np.load(os.path.join(golden_dir, "data_kat.npz"))
    os.makedirs(tmp_path / "data")
    (tmp_path / "data" / "toy.txt").write_text(str(z["text"]))
    tr, va, te, usernum, itemnum = U.data_partition("toy", str(tmp_path / "data"))
    assert usernum == int(z["usernum"]) and itemnum == int(z["itemnum"])
    users = sorted(tr)
    L = 5
    wd = U.WarpDataset(tr, usernum, itemnum, L)
    r = np.random.RandomState(0)
    for u in users:
        assert tr[u] == list(z["train.%d" % u]) and va[u] == list(z["valid.%d" % u]) and te[u] == list(z["test.%d" % u])
        _, seq, dec, pos, neg = wd.sample_data(u, r)
        assert (seq == z["warp.seq.%d" % u]).all() and (dec == z["warp.dec.%d" % u]).all() and (pos == z["warp.pos.%d" % u]).all()
        assert ((neg != 0) == z["warp.negmask.%d" % u]).all()
        assert not (set(neg[neg != 0].tolist()) & set(tr[u]))
    batch = wd.sample_batch(users, r)
    assert batch[1].shape == (len(users), L) and batch[1].dtype == np.int32
    assert sum(len(b[0]) for b in wd.epoch_batches(3, r)) == usernum
    ps = U.PopularSampler(tr, va, te, usernum, itemnum, 3)
    assert np.allclose(ps.popular_p, z["popular_p"])
    for mode in ("val", "test"):
        ed = U.EvalDataset(tr, va, te, usernum, itemnum, L, ps, mode=mode, frozen=True)
        assert ed.users == list(z["eval.%s.users" % mode])
        for u in ed.users:
            _, seq, cand, label = ed.sample_data(u)
            assert (seq == z["eval.%s.seq.%d" % (mode, u)]).all()
            assert cand[0] == int(z["eval.%s.first.%d" % (mode, u)]) and len(cand) == int(z["eval.%s.ncand.%d" % (mode, u)])
            seen = set(tr[u]) | set(va[u]) | (set(te[u]) if mode == "test" else set())
            assert not (set(cand[1:].tolist()) & seen) and len(set(cand[1:].tolist())) == 3
            assert (ed.sample_data(u)[2] == cand).all()     # frozen: same candidates on every access
            assert label[0] == 1 and label[1:].sum() == 0


def test_metrics_kat(golden_dir):
    z = np.load(os.path.join(golden_dir, "metrics_kat.npz"))
    s = z["scores"]
    ranks = np.concatenate([(x[:, 1:] > x[:, :1]).sum(1) for x in s])
    (ndcg, hr), auc = U.metrics_from_ranks(ranks, 101)
    assert abs(ndcg[5] - float(z["ndcg5"])) < 1e-6 and abs(ndcg[10] - float(z["ndcg10"])) < 1e-6
    assert abs(hr[5] - float(z["hr5"])) < 1e-9 and abs(hr[10] - float(z["hr10"])) < 1e-9 and abs(auc - float(z["auc"])) < 1e-9


def test_synthetic_generator_is_deterministic():
    from adt_amd.sasrec import synth
    h1, nu, ni = synth.generate("tiny", 23)
    h2, _, _ = synth.generate("tiny", 23)
    assert h1 == h2 and nu == 64 and ni == 120
    assert all(len(set(v)) == len(v) and min(v) >= 1 and max(v) <= ni for v in h1.values())


def test_native_batch_sampler_matches_numpy_sampler():
    """libadt_host.so (C++/OpenMP) vs the numpy sampler: identical seq/dec/pos, negatives obey the same contract."""
    from adt_amd.sasrec import synth
    h, nu, ni = synth.generate("tiny", 23)
    tr = {u: (v if len(v) < 3 else v[:-2]) for u, v in h.items()}
    wn = U.WarpDataset(tr, nu, ni, 20, native=True)
    wp = U.WarpDataset(tr, nu, ni, 20, native=False)
    if wn._native is None:
        import pytest
        pytest.skip("libadt_host.so not built")
    users = list(range(1, nu + 1))
    bn, bp = wn.sample_batch(users, np.random.RandomState(0)), wp.sample_batch(users, np.random.RandomState(0))
    for k in (1, 2, 3):
        assert (bn[k] == bp[k]).all()
    assert ((bn[4] != 0) == (bp[4] != 0)).all()
    for b, u in enumerate(users):
        ng = bn[4][b]
        assert not (set(ng[ng != 0].tolist()) & set(tr[u])) and ng.max() <= ni
    b2 = wn.sample_batch(users, np.random.RandomState(0))
    assert (b2[4] != bn[4]).any()      # a fresh stream every call


def test_parameter_registration_order_is_the_references(golden_dir):
    """torch.optim.Adam keys its state by position in model.parameters(): the mirrors must register their parameters in the
    reference's order (recorded by tools/gen_golden_param_order.py) or optimizer-state interop maps moments onto the wrong tensors."""
    import json
    from adt_amd.wide import ref_sorted
    from adt_amd.sasrec import model as sm, supersasrec as ss
    from adt_amd.bert4rec import model as bm
    from adt_amd.stosa import models as tm
    want = json.load(open(os.path.join(golden_dir, "param_order.json")))
    assert ref_sorted([n for n, _ in sm.param_table(20, 8, 16, 2, 2)], sm.REF_ORDER) == want["sasrec_nl2"]
    assert ref_sorted([n for n, _ in bm.param_table(20, 8, 16, 2, 2, 32, 2)], bm.REF_ORDER) == want["bert_nl2"]
    assert ref_sorted([n for n, _ in tm.param_table(22, 8, 16, 2, 2, 5)[0]], tm.REF_ORDER) == want["stosa_nl2"]
    sup = ["item_emb.weight", "pos_emb.weight"]
    for side, slots in (("encoder.encoder_layers", ss._ENC), ("decoder.decoder_layers", ss._DEC)):
        sup += ["%s.%d.%d.%s" % (side, i, c, n) for i in range(2) for c in range(36) for n in slots]
    assert ref_sorted(sup, sm.REF_ORDER) == want["supersasrec_nl2_c6"]
    # and the golden state_dict of a recorded reference model lists its weights in that same order
    z = np.load(os.path.join(golden_dir, "sasrec_small.npz"))
    assert [k[2:] for k in z.keys() if k.startswith("w.")] == want["sasrec_nl2"]


def test_synthetic_presets_have_the_baseline_configs_shapes(tmp_path):
    """ml20m / beauty / beauty-stosa presets (BASELINE configs[2..4]): user and item counts exact, mean history near the
    public figure, file in the reference's "<user> <item>" line format and readable by data_partition."""
    from adt_amd.sasrec import synth
    assert (synth.PRESETS["ml20m"]["users"], synth.PRESETS["ml20m"]["items"]) == (138493, 26744)
    h, U, V = synth.generate("beauty", 23)
    n = sum(len(v) for v in h.values())
    assert (U, V) == (40226, 54542) and 8.0 < n / U < 10.0
    assert max(max(v) for v in h.values()) <= V and min(min(v) for v in h.values()) >= 1
    assert all(len(set(v)) == len(v) for v in list(h.values())[:2000])          # no repeated item inside a history
    h2, _, _ = synth.generate("beauty", 23)
    assert h2[17] == h[17] and h2[40226] == h[40226]                             # seeded
    os.makedirs(tmp_path / "data")
    small = {u: h[u] for u in range(1, 301)}
    synth.write(str(tmp_path / "data" / "b.txt"), small)
    tr, va, te, usernum, itemnum = U_data_partition("b", str(tmp_path / "data"))
    assert usernum == 300 and all((len(small[u]) < 3) == (len(va[u]) == 0) for u in small)


def U_data_partition(name, d):
    return U.data_partition(name, d)
