#!/usr/bin/env python3
"""Generate tests/golden/stosa_*.npz by IMPORTING the reference (read-only, /root/reference/stosa) in the build container
(run through tools/gen_golden_wide.py stosa).  Fixtures are data only: seeded inputs, the seed that regenerates the numpy
weights, and the tensors the reference produced.

The loss assembly below calls bpr_optimization's arithmetic (stosa/trainer.py:358-391, restated with the reference's own
wasserstein_distance from stosa/modules.py because the method needs a constructed Trainer with dataloaders) and the loop
body of DistSAModelTrainer.iteration (:534-559) with the same torch functions in the same order; dropout is 0.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")


class Args:
    pass


def stosa_batch(r, B, L, V):
    """Batch in the layout of stosa/datasets.py (left-padded input_ids, dec_ids shifted by one, target_pos = next item,
    target_neg sampled, 0 where padded)."""
    inp = np.zeros((B, L), np.int64)
    dec = np.zeros((B, L), np.int64)
    pos = np.zeros((B, L), np.int64)
    neg = np.zeros((B, L), np.int64)
    for b in range(B):
        n = L if b == 0 else int(r.randint(2, L + 1))
        items = r.randint(1, V, size=n + 1)
        inp[b, L - n:] = items[:-1]
        pos[b, L - n:] = items[1:]
        neg[b, L - n:] = r.randint(1, V, size=n)
        dec[b, 1:] = inp[b, :-1]
    return inp, dec, pos, neg


def gen_stosa(tag, cfg_kw, B, seed, lam1, lam2, lr=1e-3, keep_w3=True, compact=False, steps=3):
    from tools.gen_golden_wide import _import_from
    from oracle import stosa_oracle as so
    models = _import_from("/root/reference/stosa", "models")
    modules = sys.modules["modules"]
    cfg = so.Cfg(**cfg_kw)
    P = so.init_params(cfg, seed)
    r = np.random.RandomState(seed + 1)
    for k in P:   # biases away from zero so that their gradients are exercised
        if k.endswith(".bias") and "LayerNorm" not in k:
            P[k] = (0.02 * r.standard_normal(P[k].shape)).astype(np.float32)
    inp, dec, pos, neg = stosa_batch(r, B, cfg.maxlen, cfg.item_size)
    a = Args()
    a.item_size, a.hidden_units, a.maxlen, a.num_users, a.dropout, a.attention_dropout = cfg.item_size, cfg.hidden_units, cfg.maxlen, cfg.num_users, 0.0, 0.0
    a.num_heads, a.num_layers, a.hidden_act, a.initializer_range, a.distance_metric, a.kernel_param = cfg.num_heads, cfg.num_layers, "gelu", 0.02, "wasserstein", 1.0
    a.cuda_condition, a.pvn_weight = False, cfg.pvn_weight
    m = models.DisenDistSAModel(a)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    t = [torch.from_numpy(x) for x in (inp, dec, pos, neg)]
    uid = torch.zeros(B, dtype=torch.long)
    out = {"seed": seed, "input_ids": inp, "dec_ids": dec, "pos_ids": pos, "neg_ids": neg, "lambda1": np.array(lam1), "lambda2": np.array(lam2),
           "lr": lr, "cfg": np.array([cfg.item_size, cfg.maxlen, cfg.hidden_units, cfg.num_heads, cfg.num_layers, cfg.num_users]),
           "pvn_weight": cfg.pvn_weight}
    m.eval()
    with torch.no_grad():
        mo, co, att, margins, enc_in, enc_rec, dec_out = m.finetune(t[0], t[1], uid)
        out["mean_out"], out["cov_out"] = mo.numpy(), co.numpy()
        for i in range(cfg.num_layers):
            out["enc_in_mean_%d" % i], out["enc_in_cov_%d" % i] = enc_in[i][0].numpy(), enc_in[i][1].numpy()
            out["rec_mean_%d" % i], out["rec_cov_%d" % i] = enc_rec[i][0].numpy(), enc_rec[i][1].numpy()
            out["dec_out_mean_%d" % i], out["dec_out_cov_%d" % i] = dec_out[i][0].numpy(), dec_out[i][1].numpy()
        # dist_predict_full (trainer.py:464-479) on the last position
        elu = nn.ELU()
        out["full_dist"] = modules.wasserstein_distance_matmul(mo[:, -1, :], co[:, -1, :], m.item_mean_embeddings.weight,
                                                               elu(m.item_cov_embeddings.weight) + 1).numpy()
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=0.0)
    wd = modules.wasserstein_distance
    for step in range(steps):
        mo, co, att, margins, enc_in, enc_rec, dec_out = m.finetune(t[0], t[1], uid)
        # bpr_optimization (trainer.py:358-391)
        act = nn.ELU()
        pos_mean, neg_mean = m.item_mean_embeddings(t[2]), m.item_mean_embeddings(t[3])
        pos_cov, neg_cov = act(m.item_cov_embeddings(t[2])) + 1, act(m.item_cov_embeddings(t[3])) + 1
        d = cfg.hidden_units
        pos_mean, pos_cov, neg_mean, neg_cov = (x.view(-1, d) for x in (pos_mean, pos_cov, neg_mean, neg_cov))
        sm, sc = mo.view(-1, d), co.view(-1, d)
        pos_logits, neg_logits, pos_vs_neg = wd(sm, sc, pos_mean, pos_cov), wd(sm, sc, neg_mean, neg_cov), wd(pos_mean, pos_cov, neg_mean, neg_cov)
        istarget = (t[2] > 0).view(-1).float()
        loss = torch.sum(-torch.log(torch.sigmoid(neg_logits - pos_logits + 1e-24)) * istarget) / torch.sum(istarget)
        pvn_loss = a.pvn_weight * torch.sum(torch.clamp(pos_logits - pos_vs_neg, 0) * istarget) / torch.sum(istarget)
        auc = torch.sum(((torch.sign(neg_logits - pos_logits) + 1) / 2) * istarget) / torch.sum(istarget)
        if step == 0:
            out["bpr"], out["pvn"], out["auc"] = float(loss), float(pvn_loss), float(auc)
        # iteration (trainer.py:540-559)
        dec_out.reverse()
        for l in range(cfg.num_layers):
            loss = loss + lam1[l] * F.mse_loss(enc_in[l][0], dec_out[l][0])
            loss = loss + lam1[l] * F.mse_loss(enc_in[l][1], dec_out[l][1])
        bs = enc_rec[0][0].shape[0]
        label = torch.tile(torch.arange(a.num_heads), [bs * a.maxlen, 1])
        for l in range(cfg.num_layers):
            loss = loss + lam2[l] * F.nll_loss(enc_rec[l][0].view(bs * a.maxlen, a.num_heads, a.num_heads), label)
            loss = loss + lam2[l] * F.nll_loss(enc_rec[l][1].view(bs * a.maxlen, a.num_heads, a.num_heads), label)
        loss = loss + pvn_loss
        opt.zero_grad()
        loss.backward()
        if step == 0:
            out["loss"] = float(loss.item())
            none = []
            for k, p in m.named_parameters():
                if p.grad is None:
                    none.append(k)
                else:
                    out["grad." + k] = p.grad.numpy().copy()
            out["grad_none"] = np.array(none)
        opt.step()
        if step == 0 or (step == 2 and keep_w3):
            for k, p in m.named_parameters():
                out["w%d." % (step + 1) + k] = p.detach().numpy().copy()
    path = os.path.join(OUT, "stosa_%s.npz" % tag)
    if compact:
        from tools.gen_golden_inputs import compact as _compact
        out = _compact(out)
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", out["loss"], "none-grads", len(out["grad_none"]), "%.1f KB" % (os.path.getsize(path) / 1024))


def main():
    gen_stosa("small", dict(item_size=42, maxlen=12, hidden_units=64, num_heads=4, num_layers=1, num_users=4, pvn_weight=0.005), B=4, seed=21,
              lam1=[0.3], lam2=[0.2])
    gen_stosa("l2h2", dict(item_size=33, maxlen=20, hidden_units=64, num_heads=2, num_layers=2, num_users=3, pvn_weight=0.1), B=3, seed=22,
              lam1=[0.25, 0.1], lam2=[0.15, 0.05], keep_w3=False)
    gen_stosa("h1", dict(item_size=28, maxlen=9, hidden_units=64, num_heads=1, num_layers=1, num_users=2, pvn_weight=0.05), B=2, seed=23,
              lam1=[0.1], lam2=[0.05], keep_w3=False)


def main_cfg5():
    """BASELINE configs[4]: STOSA-ADT at the Amazon-Beauty template shape (stosa/templates/Beauty.json: d=64, H=4, 1 layer, L=100,
    pvn_weight 0.005; stosa/data/Beauty.txt has 12,101 items => item_size 12,103; 22,363 users), B=8; get_lambdas("Beauty")[:1]."""
    gen_stosa("cfg5_beauty", dict(item_size=12103, maxlen=100, hidden_units=64, num_heads=4, num_layers=1, num_users=22363, pvn_weight=0.005), B=8,
              seed=33, lam1=[0.0021], lam2=[0.0009], keep_w3=False, compact=True, steps=1)


if __name__ == "__main__":
    sys.path.insert(0, REPO)
    main()
