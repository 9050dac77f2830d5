"""GPU: trained-metric parity of BERT4Rec-ADT and STOSA-ADT with the reference on the same data.  The reference models (PyTorch CPU,
fp32; imported read-only in the build container by tools/ref_train_wide.py) were trained for 20 epochs on the seeded synthetic
"ml1m-small" set (1,200 users, 800 items) with the batches of tools/wide_parity_common.py; tests/golden/ref_ndcg_{bert,stosa}_small.json
hold their metrics at epochs 10 and 20.  Here the HIP path (bf16 MFMA operands, hash dropout, fused trainer, HIP graph) trains on the
same batches with the same hyper-parameters.  Tolerances (abs, stated per metric below) cover the run-to-run spread that dropout
streams and float atomics cause on 1,200 evaluation users (one user = 0.0008 HR); the measured seed spread of the HIP path is in
profiles/r01_ndcg_wide_ours.json."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_bert_ranking_matches_reference(golden_dir):
    from tools.gpu_wide_ndcg_run import run_bert
    ref = json.load(open(os.path.join(golden_dir, "ref_ndcg_bert_small.json")))
    ours = run_bert(seed=23)
    assert [e["epoch"] for e in ours["evals"]] == [e["epoch"] for e in ref["evals"]] == [10, 20]
    for eo, er in zip(ours["evals"], ref["evals"]):
        for mode in ("val", "test"):
            assert abs(eo[mode]["ndcg10"] - er[mode]["ndcg10"]) <= 0.03, (eo["epoch"], mode, eo[mode], er[mode])
            assert abs(eo[mode]["hr10"] - er[mode]["hr10"]) <= 0.04, (eo["epoch"], mode, eo[mode], er[mode])
            assert abs(eo[mode]["auc"] - er[mode]["auc"]) <= 0.02, (eo["epoch"], mode, eo[mode], er[mode])
    assert abs(ours["loss"][-1] - ref["loss"][-1]) <= 0.05 * ref["loss"][-1]
    assert ours["evals"][-1]["test"]["ndcg10"] > 2 * 0.045          # far above the random ranker (NDCG@10 of 101 candidates ~ 0.045)


def test_stosa_ranking_matches_reference(golden_dir):
    from tools.gpu_wide_ndcg_run import run_stosa
    ref = json.load(open(os.path.join(golden_dir, "ref_ndcg_stosa_small.json")))
    ours = run_stosa(seed=42)
    assert [e["epoch"] for e in ours["evals"]] == [e["epoch"] for e in ref["evals"]] == [10, 20]
    for eo, er in zip(ours["evals"], ref["evals"]):
        for mode in ("val", "test"):
            assert abs(eo[mode]["ndcg10"] - er[mode]["ndcg10"]) <= 0.03, (eo["epoch"], mode, eo[mode], er[mode])
            assert abs(eo[mode]["hit10"] - er[mode]["hit10"]) <= 0.04, (eo["epoch"], mode, eo[mode], er[mode])
            assert abs(eo[mode]["mrr"] - er[mode]["mrr"]) <= 0.03, (eo["epoch"], mode, eo[mode], er[mode])
    assert abs(ours["loss"][-1] - ref["loss"][-1]) <= 0.05 * ref["loss"][-1]
    assert ours["evals"][-1]["test"]["ndcg10"] > 0.1               # full-sort over 800 items: random is ~0.006


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
