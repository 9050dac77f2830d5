#!/usr/bin/env python3
"""Train the REFERENCE BERT4Rec-ADT / STOSA-ADT models (imported read-only from /root/reference, PyTorch CPU) on the seeded
"ml1m-small" synthetic dataset and record ranking metrics on frozen candidates / the full item set.  Build container only;
the recorded JSON (tests/golden/ref_ndcg_{bert,stosa}_small.json) is what travels.  The loop bodies are the reference's
(bert4rec/trainer.py:100-138; stosa/trainer.py:534-559 with bpr_optimization :358-391) driven with the same torch calls; the
batches come from tools/wide_parity_common.py so that the HIP run (tools/gpu_wide_ndcg_run.py) trains on the same rows.

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_train_wide.py bert   # -> tests/golden/ref_ndcg_bert_small.json
    PYTHONDONTWRITEBYTECODE=1 python tools/ref_train_wide.py stosa  # -> tests/golden/ref_ndcg_stosa_small.json
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tools import wide_parity_common as C  # noqa: E402
from tools.gen_golden_wide import _import_from  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


class Args:
    pass


def run_bert(seed=23, suffix=""):
    cfg = C.BERT
    bert = _import_from("/root/reference/bert4rec", "model.bert")
    ds, evals, usernum, itemnum = C.bert_data()
    a = Args()
    for k in ("maxlen", "num_heads", "num_layers", "dropout", "hidden_units", "type_vocab_size", "inner_units", "attention_dropout"):
        setattr(a, k, cfg[k])
    a.device = "cpu"
    torch.manual_seed(seed)
    m = bert.BertModel(usernum, itemnum, a)
    for name, module in m.named_modules():        # bert4rec/trainer.py:29-37
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.01, std=0.02)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=cfg["weight_decay"])
    ce = nn.CrossEntropyLoss(ignore_index=0)
    L, H = cfg["maxlen"], cfg["num_heads"]
    pos = torch.from_numpy(np.tile(np.arange(L), (cfg["batch_size"], 1)))
    sent = torch.zeros(cfg["batch_size"], L, dtype=torch.long)
    log = {"seed": seed, "users": usernum, "items": itemnum, "cfg": {k: v for k, v in cfg.items()}, "evals": [], "loss": []}
    t0 = time.time()
    for epoch in range(cfg["epochs"]):
        m.train()
        tot, nb = 0.0, 0
        for src, dec, lab in C.bert_batches(ds, epoch):
            opt.zero_grad()
            logits, enc_in, dec_out, rec = m(torch.from_numpy(src).long(), torch.from_numpy(dec).long(), pos, sent, pos, sent)
            loss = ce(logits.view(-1, logits.size(-1)), torch.from_numpy(lab).long().view(-1))
            for i in range(len(enc_in)):
                if cfg["lambda1"][i] != 0:
                    loss = loss + cfg["lambda1"][i] * F.mse_loss(enc_in[i], dec_out[i])
            label = torch.tile(torch.arange(H), [logits.shape[0] * L, 1])
            for l in range(len(rec)):
                if cfg["lambda2"][l] != 0:
                    loss = loss + cfg["lambda2"][l] * F.nll_loss(rec[l].view(logits.shape[0] * L, H, H), label)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), cfg["clip"])
            opt.step()
            tot += float(loss); nb += 1
        log["loss"].append(tot / nb)
        print("epoch %d loss %.4f (%.0fs)" % (epoch + 1, tot / nb, time.time() - t0), flush=True)
        if (epoch + 1) % 10 == 0:
            m.eval()
            rec_e = {"epoch": epoch + 1}
            with torch.no_grad():
                for mode in ("val", "test"):
                    ranks = []
                    for seq, cand in evals[mode]:
                        B = len(seq)
                        p = torch.from_numpy(np.tile(np.arange(L), (B, 1)))
                        s = torch.zeros(B, L, dtype=torch.long)
                        pred = -m.predict(None, torch.from_numpy(seq).long(), p, s, torch.from_numpy(cand).long())
                        ranks.append(pred.argsort(dim=1).argsort(dim=1)[:, 0].numpy())     # bert4rec/trainer.py:68
                    rec_e[mode] = C.rank_metrics(np.concatenate(ranks), 101)
            log["evals"].append(rec_e)
            print(rec_e, flush=True)
    log["train_seconds"] = time.time() - t0
    json.dump(log, open(os.path.join(OUT, "ref_ndcg_bert_small%s.json" % suffix), "w"), indent=1)


def run_stosa(seed=42, suffix="", deterministic=False):
    """deterministic: dropout 0 and the numpy initial weights of oracle.stosa_oracle.init_params(cfg, seed) -- the HIP path starts from
    the same weights (tests/test_ndcg_parity_wide.py), so the two training runs differ by floating-point rounding only."""
    cfg = dict(C.STOSA)
    if deterministic:
        cfg["dropout"] = cfg["attention_dropout"] = 0.0
    models = _import_from("/root/reference/stosa", "models")
    modules = sys.modules["modules"]
    train, valid, test, vm, tm, max_item, nu = C.stosa_data()
    a = Args()
    a.item_size, a.hidden_units, a.maxlen, a.num_users, a.dropout, a.attention_dropout = max_item + 2, cfg["hidden_units"], cfg["maxlen"], nu, cfg["dropout"], cfg["attention_dropout"]
    a.num_heads, a.num_layers, a.hidden_act, a.initializer_range, a.distance_metric, a.kernel_param = cfg["num_heads"], cfg["num_layers"], "gelu", 0.02, "wasserstein", 1.0
    a.cuda_condition, a.pvn_weight = False, cfg["pvn_weight"]
    torch.manual_seed(seed)
    m = models.DisenDistSAModel(a)
    if deterministic:
        from oracle import stosa_oracle as so
        ocfg = so.Cfg(a.item_size, a.maxlen, a.hidden_units, a.num_heads, a.num_layers, num_users=nu, pvn_weight=cfg["pvn_weight"])
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in so.init_params(ocfg, seed).items()}, strict=True)
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], betas=(0.9, 0.999), weight_decay=0.0)
    wd = modules.wasserstein_distance
    d, L, H = cfg["hidden_units"], cfg["maxlen"], cfg["num_heads"]
    log = {"seed": seed, "users": nu, "items": max_item, "cfg": {k: v for k, v in cfg.items()}, "evals": [], "loss": []}
    t0 = time.time()

    def full_sort(ds, matrix):
        from adt_amd.stosa.trainer import get_full_sort_score
        preds, answers = [], []
        with torch.no_grad():
            for s in range(0, len(ds), 256):
                users, inp, dec, pos, neg, ans = ds.batch(np.arange(s, min(s + 256, len(ds))))
                mo, co, _, _, _, _, _ = m.finetune(torch.from_numpy(inp).long(), torch.from_numpy(dec).long(), torch.from_numpy(users))
                dist = modules.wasserstein_distance_matmul(mo[:, -1, :], co[:, -1, :], m.item_mean_embeddings.weight,
                                                           nn.ELU()(m.item_cov_embeddings.weight) + 1).numpy().copy()
                dist[matrix[users].toarray() > 0] = 1e24
                ind = np.argpartition(dist, 40)[:, :40]
                arr = dist[np.arange(len(dist))[:, None], ind]
                preds.append(ind[np.arange(len(dist))[:, None], np.argsort(arr)])
                answers.append(ans)
        sc = get_full_sort_score(np.concatenate(answers), np.concatenate(preds))
        return {"hit10": sc[4], "ndcg10": sc[5], "hit20": sc[8], "ndcg20": sc[9], "mrr": sc[-1]}

    for epoch in range(cfg["epochs"]):
        m.train()
        tot, nb = 0.0, 0
        for users, inp, dec, pos, neg, _ in C.stosa_batches(train, epoch):
            t = [torch.from_numpy(x).long() for x in (inp, dec, pos, neg)]
            mo, co, att, margins, enc_in, enc_rec, dec_out = m.finetune(t[0], t[1], torch.from_numpy(users))
            act = nn.ELU()
            pm, nm = m.item_mean_embeddings(t[2]).view(-1, d), m.item_mean_embeddings(t[3]).view(-1, d)
            pc, nc = (act(m.item_cov_embeddings(t[2])) + 1).view(-1, d), (act(m.item_cov_embeddings(t[3])) + 1).view(-1, d)
            sm, sc = mo.view(-1, d), co.view(-1, d)
            pl, nl, pvn = wd(sm, sc, pm, pc), wd(sm, sc, nm, nc), wd(pm, pc, nm, nc)
            ist = (t[2] > 0).view(-1).float()
            loss = torch.sum(-torch.log(torch.sigmoid(nl - pl + 1e-24)) * ist) / torch.sum(ist)
            pvn_loss = cfg["pvn_weight"] * torch.sum(torch.clamp(pl - pvn, 0) * ist) / torch.sum(ist)
            dec_out.reverse()
            for l in range(cfg["num_layers"]):
                loss = loss + cfg["lambda1"][l] * F.mse_loss(enc_in[l][0], dec_out[l][0]) + cfg["lambda1"][l] * F.mse_loss(enc_in[l][1], dec_out[l][1])
            label = torch.tile(torch.arange(H), [len(users) * L, 1])
            for l in range(cfg["num_layers"]):
                loss = loss + cfg["lambda2"][l] * F.nll_loss(enc_rec[l][0].view(len(users) * L, H, H), label)
                loss = loss + cfg["lambda2"][l] * F.nll_loss(enc_rec[l][1].view(len(users) * L, H, H), label)
            loss = loss + pvn_loss
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot += float(loss); nb += 1
        log["loss"].append(tot / nb)
        print("epoch %d loss %.4f (%.0fs)" % (epoch + 1, tot / nb, time.time() - t0), flush=True)
        if (epoch + 1) % 10 == 0:
            m.eval()
            rec_e = {"epoch": epoch + 1, "val": full_sort(valid, vm), "test": full_sort(test, tm)}
            log["evals"].append(rec_e)
            print(rec_e, flush=True)
    log["train_seconds"] = time.time() - t0
    json.dump(log, open(os.path.join(OUT, "ref_ndcg_stosa_small%s.json" % suffix), "w"), indent=1)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if sys.argv[1] == "stosa_det":      # deterministic run: dropout 0, shared numpy init -> ref_ndcg_stosa_small_det.json
        run_stosa(int(sys.argv[2]) if len(sys.argv) > 2 else 42, "_det", deterministic=True)
        sys.exit(0)
    fn = {"bert": run_bert, "stosa": run_stosa}[sys.argv[1]]
    if len(sys.argv) > 2:          # extra model-init seeds (reference seed spread): ... <mode> <seed> -> *_small_s<seed>.json
        fn(int(sys.argv[2]), "_s%s" % sys.argv[2])
    else:
        fn()
