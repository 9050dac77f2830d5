"""GPU: ranking parity with the reference on the same split.  The reference (PyTorch CPU, fp32) was trained for 30
epochs on the seeded synthetic ml-1m-shaped file with THREE seeds (tools/ref_train_ndcg.py --seed 23 / 24 / 25; ~55 minutes each on the
build container's cores) and scored on candidate sets frozen with RandomState(23); tests/golden/ref_ndcg_ml1m{,_s24,_s25}.json hold
its NDCG@10 / HR@10 / AUC at epochs 10, 20 and 30.  Here the HIP path (bf16 MFMA operands, dropout 0.5, fused trainer, HIP graph) trains
three seeds on the same file with the same hyper-parameters and is scored on the same candidates.

Tolerance (north_star / SURVEY 8d): on the test split at epoch 30 the mean over the HIP seeds lies within max(0.01, 2 sigma) of the mean
over the reference seeds, sigma = the reference's own sample standard deviation over its three seeds (0.003 NDCG@10, 0.003-0.005 HR@10: the
bound is the stated +-0.01).  Epochs 10 / 20 (the steep part of the curve, reference sigma 0.009-0.014) and the validation split are
checked at max(0.015, 3 sigma)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REF_SEEDS = {23: "ref_ndcg_ml1m.json", 24: "ref_ndcg_ml1m_s24.json", 25: "ref_ndcg_ml1m_s25.json"}


def test_ndcg_hr_match_reference_on_same_split(golden_dir):
    from tools.gpu_ndcg_run import run
    from adt_amd.sasrec import synth
    refs = [json.load(open(os.path.join(golden_dir, f))) for f in REF_SEEDS.values()]
    data = synth.generate("ml1m", 23)
    ours = [run("ml1m", 30, 10, seed=s, precision="bf16", data=data) for s in REF_SEEDS]
    for r in refs + ours:
        assert [e["epoch"] for e in r["evals"]] == [10, 20, 30]
    for i, epoch in enumerate((10, 20, 30)):
        for mode in ("val", "test"):
            for k in ("ndcg10", "hr10"):
                rv = np.array([r["evals"][i][mode][k] for r in refs])
                ov = np.array([o["evals"][i][mode][k] for o in ours])
                sig = rv.std(ddof=1)
                # the stated bound on the final checkpoint's test split; the intermediate checkpoints and the validation split (the same
                # runs, twelve simultaneous checks of not bit-reproducible trainings) at max(0.015, 3 sigma): see test_ndcg_parity_wide.py
                tol = max(0.01, 2.0 * sig) if (epoch == 30 and mode == "test") else max(0.015, 3.0 * sig)
                assert abs(ov.mean() - rv.mean()) <= tol, (epoch, mode, k, ov, rv, tol)
    assert min(o["evals"][-1]["test"]["ndcg10"] for o in ours) > 0.25     # the model actually learned the sequential structure


def test_template_width_ndcg_matches_reference(golden_dir):
    """The same comparison at the SHIPPED template width (sasrec/templates/ml-1m.json: hidden_units 256, 2 heads => head size 128),
    which runs on the general kernels (adt_amd/sasrec/model_wide.py): reference CPU run of tools/ref_train_ndcg.py --preset
    ml1m-small --hidden 256 --maxlen 100 (30 epochs, 1,200 users) in tests/golden/ref_ndcg_small_d256.json; three HIP seeds were
    0.002..0.012 away in NDCG@10 at epoch 30 (profiles/r01_ndcg_small_d256_ours.json).  Early checkpoints sit on the steep part of
    the curve (NDCG doubles between epochs 10 and 20), so they get a wider band."""
    from tools.gpu_ndcg_run import run
    from adt_amd.sasrec import synth
    ref = json.load(open(os.path.join(golden_dir, "ref_ndcg_small_d256.json")))
    data = synth.generate("ml1m-small", 23)
    ours = run("ml1m-small", 30, 10, seed=23, precision="bf16", hidden=256, maxlen=100, data=data)
    assert [e["epoch"] for e in ours["evals"]] == [e["epoch"] for e in ref["evals"]] == [10, 20, 30]
    for eo, er in zip(ours["evals"], ref["evals"]):
        tol_n, tol_h = (0.03, 0.04) if eo["epoch"] == 30 else (0.06, 0.08)
        for mode in ("val", "test"):
            assert abs(eo[mode]["ndcg10"] - er[mode]["ndcg10"]) <= tol_n, (eo["epoch"], mode, eo[mode], er[mode])
            assert abs(eo[mode]["hr10"] - er[mode]["hr10"]) <= tol_h, (eo["epoch"], mode, eo[mode], er[mode])
    assert ours["evals"][-1]["test"]["ndcg10"] > 0.3


@pytest.mark.parametrize("precision", ["bf16"])
def test_deterministic_training_matches_reference(golden_dir, precision):
    """North star: "ranking outputs match the reference CPU path's NDCG@10 / HR@10 within a stated fp tolerance on the same split".
    With everything random taken out -- dropout 0, the SAME initial weights (oracle.sasrec_oracle.init_params(cfg, 23)) and the SAME
    seeded batches (WarpDataset.epoch_batches(256, RandomState(1000 + epoch))) on both sides -- 30 epochs of the reference
    (tools/ref_train_ndcg.py --deterministic, PyTorch CPU fp32: tests/golden/ref_ndcg_small_det.json) and of the HIP path (bf16 MFMA
    operands, fused trainer, HIP graph) differ by floating-point rounding only.  Stated tolerance: NDCG@10, HR@10 and AUC within
    +-0.01 absolute on validation and test at epochs 10, 20 and 30 (1,200 users: one user is 0.0008 HR); epoch-mean loss within 1 %."""
    from tools.gpu_ndcg_run import run
    from adt_amd.sasrec import synth
    ref = json.load(open(os.path.join(golden_dir, "ref_ndcg_small_det.json")))
    assert ref["deterministic"] and ref["hidden"] == 64 and ref["maxlen"] == 200
    data = synth.generate("ml1m-small", 23)
    ours = run("ml1m-small", 30, 10, seed=23, precision=precision, data=data, deterministic=True)
    for lo, lr in zip(ours["loss"], ref["loss"]):
        assert abs(lo - lr) <= 0.01 * lr, (ours["loss"], ref["loss"])
    assert [e["epoch"] for e in ours["evals"]] == [e["epoch"] for e in ref["evals"]] == [10, 20, 30]
    for eo, er in zip(ours["evals"], ref["evals"]):
        for mode in ("val", "test"):
            for k in ("ndcg10", "hr10", "auc"):
                assert abs(eo[mode][k] - er[mode][k]) <= 0.01, (eo["epoch"], mode, k, eo[mode], er[mode])
