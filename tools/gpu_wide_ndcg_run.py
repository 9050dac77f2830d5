#!/usr/bin/env python3
"""Train BERT4Rec-ADT / STOSA-ADT on the HIP path on the same seeded synthetic data, batches and hyper-parameters as the
reference runs recorded by tools/ref_train_wide.py (tests/golden/ref_ndcg_{bert,stosa}_small.json) and report the same metrics.
    python tools/gpu_wide_ndcg_run.py --model bert --seeds 23 24 25 --out gpurun_out/ndcg_bert_ours.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tools import wide_parity_common as C  # noqa: E402


class _A:
    pass


def run_bert(seed=23, precision="bf16", use_graph=True, data=None):
    import torch
    from adt_amd.bert4rec.model import BertModel
    from adt_amd.bert4rec.trainer import FusedBertTrainer
    cfg = C.BERT
    ds, evals, usernum, itemnum = data or C.bert_data()
    a = _A()
    for k in ("maxlen", "num_heads", "num_layers", "dropout", "hidden_units", "type_vocab_size", "inner_units", "attention_dropout"):
        setattr(a, k, cfg[k])
    a.device, a.precision = "cuda:0", precision
    torch.manual_seed(seed)
    m = BertModel(usernum, itemnum, a)
    tr = FusedBertTrainer(m, cfg["lambda1"], cfg["lambda2"], lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=cfg["weight_decay"], clip=cfg["clip"],
                          use_graph=use_graph, seed=seed)
    log = {"seed": seed, "evals": [], "loss": []}
    t0 = time.time()
    for epoch in range(cfg["epochs"]):
        m.train()
        tot = []
        for src, dec, lab in C.bert_batches(ds, epoch):
            tr.step(src, dec, lab)
            tot.append(tr.loss())
        log["loss"].append(float(torch.stack(tot).mean()))
        if (epoch + 1) % 10 == 0:
            rec = {"epoch": epoch + 1}
            for mode in ("val", "test"):
                (ndcg, hr), auc = tr.evaluate(evals[mode])
                rec[mode] = {"ndcg10": ndcg[10], "hr10": hr[10], "auc": auc}
            log["evals"].append(rec)
            print("bert", seed, rec, flush=True)
    log["train_seconds"] = time.time() - t0
    return log


def run_stosa(seed=42, precision="bf16", use_graph=True, data=None, deterministic=False):
    """deterministic: dropout 0 and the numpy initial weights oracle.stosa_oracle.init_params(cfg, seed) -- exactly what
    tools/ref_train_wide.py stosa_det gave the reference, so the two runs differ by floating-point rounding only."""
    import torch
    from adt_amd.stosa.main import _evaluate
    from adt_amd.stosa.models import DisenDistSAModel
    from adt_amd.stosa.trainer import FusedStosaTrainer
    cfg = dict(C.STOSA)
    if deterministic:
        cfg["dropout"] = cfg["attention_dropout"] = 0.0
    train, valid, test, vm, tm, max_item, nu = data or C.stosa_data()
    a = _A()
    a.item_size, a.hidden_units, a.maxlen, a.num_users, a.dropout, a.attention_dropout = max_item + 2, cfg["hidden_units"], cfg["maxlen"], nu, cfg["dropout"], cfg["attention_dropout"]
    a.num_heads, a.num_layers, a.hidden_act, a.initializer_range, a.distance_metric, a.kernel_param = cfg["num_heads"], cfg["num_layers"], "gelu", 0.02, "wasserstein", 1.0
    a.cuda_condition, a.pvn_weight, a.device, a.precision = True, cfg["pvn_weight"], "cuda:0", precision
    torch.manual_seed(seed)
    m = DisenDistSAModel(a)
    if deterministic:
        from oracle import stosa_oracle as so
        ocfg = so.Cfg(a.item_size, a.maxlen, a.hidden_units, a.num_heads, a.num_layers, num_users=nu, pvn_weight=cfg["pvn_weight"])
        m.load_numpy(so.init_params(ocfg, seed))
    tr = FusedStosaTrainer(m, cfg["lambda1"], cfg["lambda2"], lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=0.0, use_graph=use_graph, seed=seed)
    log = {"seed": seed, "evals": [], "loss": []}
    t0 = time.time()
    for epoch in range(cfg["epochs"]):
        m.train()
        tot = []
        for users, inp, dec, pos, neg, _ in C.stosa_batches(train, epoch):
            tr.step(inp, dec, pos, neg)
            tot.append(tr.loss())
        log["loss"].append(float(torch.stack(tot).mean()))
        if (epoch + 1) % 10 == 0:
            m.eval()
            rec = {"epoch": epoch + 1}
            for mode, ds, mat in (("val", valid, vm), ("test", test, tm)):
                sc = _evaluate(tr, ds, mat, 256)
                rec[mode] = {"hit10": float(sc[4]), "ndcg10": float(sc[5]), "hit20": float(sc[8]), "ndcg20": float(sc[9]), "mrr": float(sc[-1])}
            log["evals"].append(rec)
            print("stosa", seed, rec, flush=True)
    log["train_seconds"] = time.time() - t0
    return log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=("bert", "stosa"), required=True)
    ap.add_argument("--seeds", type=int, nargs="+", default=[23])
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    data = C.bert_data() if a.model == "bert" else C.stosa_data()
    fn = run_bert if a.model == "bert" else run_stosa
    runs = [fn(s, a.precision, data=data) for s in a.seeds]
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"model": a.model, "precision": a.precision, "runs": runs}, f, indent=1)


if __name__ == "__main__":
    main()
