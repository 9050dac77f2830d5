"""GPU: trained-metric parity of BERT4Rec-ADT and STOSA-ADT with the reference on the same data.  The reference models (PyTorch CPU,
fp32; imported read-only in the build container by tools/ref_train_wide.py) were trained for 20 epochs on the seeded synthetic
"ml1m-small" set (1,200 users, 800 items) with the batches of tools/wide_parity_common.py; tests/golden/ref_ndcg_{bert,stosa}_small.json
hold their metrics at epochs 10 and 20.  Here the HIP path (bf16 MFMA operands, hash dropout, fused trainer, HIP graph) trains on the
same batches with the same hyper-parameters.  The reference ran THREE model seeds (ref_ndcg_{bert,stosa}_small{,_s*}.json), the HIP
path runs five; the stated tolerance max(0.01, 2 sigma of the reference's own seed spread) applies to the seed means of NDCG@10 / HR@10 on
the final checkpoint's test split (one user of the 1,200 evaluation users is 0.0008 HR); see _seed_mean_check for the rest."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


BERT_REF = ("ref_ndcg_bert_small.json", "ref_ndcg_bert_small_s24.json", "ref_ndcg_bert_small_s25.json")          # model seeds 23, 24, 25
STOSA_REF = ("ref_ndcg_stosa_small.json", "ref_ndcg_stosa_small_s43.json", "ref_ndcg_stosa_small_s44.json")      # model seeds 42, 43, 44


def _seed_mean_check(ours, refs, keys, headline):
    """north_star / SURVEY 8(d): on the FINAL checkpoint's test split, NDCG@10 and HR@10 (`headline`) averaged over the HIP seeds lie within
    max(0.01, 2 sigma) of the mean over the three reference seeds, sigma = the reference's own sample standard deviation at that checkpoint
    (tools/ref_train_wide.py <model> <seed>).  Every other (checkpoint, split, metric) of the same runs is checked too, at
    max(0.015, 3 sigma): the HIP runs are not bit-reproducible (float atomics), the difference of two small-sample means has a standard
    deviation of ~0.7-0.8 sigma, and twelve-odd simultaneous 2 sigma checks would fail one run in five by chance alone."""
    for r in refs + ours:
        assert [e["epoch"] for e in r["evals"]] == [10, 20]
    for i, epoch in enumerate((10, 20)):
        for mode in ("val", "test"):
            for k in keys:
                rv = np.array([r["evals"][i][mode][k] for r in refs])
                ov = np.array([o["evals"][i][mode][k] for o in ours])
                sig = rv.std(ddof=1)
                tol = max(0.01, 2.0 * sig) if (epoch == 20 and mode == "test" and k in headline) else max(0.015, 3.0 * sig)
                assert abs(ov.mean() - rv.mean()) <= tol, (epoch, mode, k, ov, rv, tol)


def test_bert_ranking_matches_reference(golden_dir):
    from tools.gpu_wide_ndcg_run import run_bert
    from tools import wide_parity_common as C
    refs = [json.load(open(os.path.join(golden_dir, f))) for f in BERT_REF]
    data = C.bert_data()
    ours = [run_bert(seed=s, data=data) for s in (23, 24, 25, 26, 27)]
    _seed_mean_check(ours, refs, ("ndcg10", "hr10", "auc"), headline=("ndcg10", "hr10"))
    rl = np.mean([r["loss"][-1] for r in refs])
    assert abs(np.mean([o["loss"][-1] for o in ours]) - rl) <= 0.02 * rl
    assert min(o["evals"][-1]["test"]["ndcg10"] for o in ours) > 2 * 0.045          # far above the random ranker (NDCG@10 of 101 candidates ~ 0.045)


def test_stosa_ranking_matches_reference(golden_dir):
    from tools.gpu_wide_ndcg_run import run_stosa
    from tools import wide_parity_common as C
    refs = [json.load(open(os.path.join(golden_dir, f))) for f in STOSA_REF]
    data = C.stosa_data()
    ours = [run_stosa(seed=s, data=data) for s in (42, 43, 44, 45, 46)]
    _seed_mean_check(ours, refs, ("ndcg10", "hit10", "mrr"), headline=("ndcg10", "hit10"))
    rl = np.mean([r["loss"][-1] for r in refs])
    assert abs(np.mean([o["loss"][-1] for o in ours]) - rl) <= 0.05 * rl
    assert min(o["evals"][-1]["test"]["ndcg10"] for o in ours) > 0.1               # full-sort over 800 items: random is ~0.006


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_stosa_deterministic_training_matches_reference(golden_dir, precision):
    """The same 20 epochs with dropout 0 from the SAME initial weights (oracle.stosa_oracle.init_params, seed 42) as the reference run
    recorded in ref_ndcg_stosa_small_det.json: nothing random is left, so the comparison is as tight as north_star states -- NDCG@10
    and HIT@10 within +-0.01 absolute on validation and test at epochs 10 and 20, the epoch-mean loss within 1 % throughout.  (Round 1's
    stochastic runs sat ~0.009 below the mean of three reference seeds whose own spread is 0.0087: this run shows the two
    implementations train to the same point when the dropout streams and the initialisation are taken out.)"""
    from tools.gpu_wide_ndcg_run import run_stosa
    ref = json.load(open(os.path.join(golden_dir, "ref_ndcg_stosa_small_det.json")))
    ours = run_stosa(seed=42, precision=precision, deterministic=True)
    for lo, lr in zip(ours["loss"], ref["loss"]):
        assert abs(lo - lr) <= 0.01 * lr, (ours["loss"], ref["loss"])
    for eo, er in zip(ours["evals"], ref["evals"]):
        for mode in ("val", "test"):
            for k in ("ndcg10", "hit10", "mrr"):
                assert abs(eo[mode][k] - er[mode][k]) <= 0.01, (precision, eo["epoch"], mode, k, eo[mode], er[mode])
