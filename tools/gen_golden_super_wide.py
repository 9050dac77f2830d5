#!/usr/bin/env python3
"""Generate tests/golden/superbert_*.npz and superstosa_*.npz by IMPORTING the reference supernets (read-only:
/root/reference/bert4rec/model/superbert.py + modules.py SuperEncoder / SuperDecoder, /root/reference/stosa/supernet.py +
super_modules.py) in the build container.  Fixtures are data only: seeded inputs, the seed of the weight fill
(tools/gen_golden_inputs.seeded_params) and what the reference produced.  The losses are the loop bodies of the warm-up
(bert4rec/evolution.py:266-296; stosa/super_trainer.py:205-235 with bpr_optimization :30-62) driven with the same torch calls;
dropout is 0.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_super_wide.py [bert] [stosa]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from adt_amd.supersearch import cand_to_block  # noqa: E402  (host arithmetic only; checked against the reference's shared_idx below)
from tools.gen_golden_inputs import compact, seeded_params  # noqa: E402
from tools.gen_golden_stosa import stosa_batch  # noqa: E402
from tools.gen_golden_wide import _import_from, bert_batch  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


class Args:
    pass


def _grads(m, out):
    none = []
    for k, p in m.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            out["grad." + k] = p.grad.numpy().copy()
    out["grad_none"] = np.array(none)


def gen_bert(tag, item_num, L, d, H, nl, rec_choice, ind_choice, B, seed, cand, wd=1e-2, lr=1e-3, clip=5.0):
    sb = _import_from("/root/reference/bert4rec", "model.superbert")
    a = Args()
    a.maxlen, a.num_heads, a.num_layers, a.device, a.dropout, a.hidden_units, a.type_vocab_size, a.attention_dropout = L, H, nl, "cpu", 0.0, d, 2, 0.0
    m = sb.SuperBertModel(1, item_num, np.array(rec_choice), np.array(ind_choice), a)
    P = seeded_params({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    block, rec_w, ind_w = cand_to_block(rec_choice, ind_choice, cand)
    m.set_choice(block)
    r = np.random.RandomState(seed + 1)
    src, dec, lab = bert_batch(r, B, L, item_num)
    tsrc, tdec = torch.from_numpy(src), torch.from_numpy(dec)
    pos = torch.from_numpy(np.tile(np.arange(L), (B, 1)))
    sent = torch.zeros(B, L, dtype=torch.long)
    out = {"seed": seed, "cfg": np.array([item_num, L, d, H, nl]), "rec_choice": np.array(rec_choice), "ind_choice": np.array(ind_choice),
           "cand": np.array(cand), "src": src, "dec": dec, "labels": lab, "wd": wd, "lr": lr, "clip": clip,
           "shared_idx": np.array(m.encoder.shared_idx), "shared_weights": np.array(m.encoder.shared_weights)}
    m.eval()
    with torch.no_grad():
        logits, enc_in, dec_out, rec = m(tsrc, tdec, pos, sent, pos, sent)
        out["logits"] = logits.numpy()
        for i in range(nl):
            out["enc_in_%d" % i], out["dec_out_%d" % i], out["rec_%d" % i] = enc_in[i].numpy(), dec_out[i].numpy(), rec[i].numpy()
        items = r.randint(1, item_num + 1, size=(B, 9)).astype(np.int64)
        out["items"] = items
        out["predict"] = m.predict(None, tsrc, pos, sent, torch.from_numpy(items)).numpy()
        # the same scores under a second block choice: what a batched two-candidate evaluation must reproduce
        cand2 = [1.0 - c for c in cand]
        m.set_choice(cand_to_block(rec_choice, ind_choice, cand2)[0])
        out["cand2"], out["predict2"] = np.array(cand2), m.predict(None, tsrc, pos, sent, torch.from_numpy(items)).numpy()
        m.set_choice(block)
    m.train()
    ce = nn.CrossEntropyLoss(ignore_index=0)
    opt = torch.optim.AdamW(m.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=wd)
    labels = torch.from_numpy(lab)
    for step in range(2):
        opt.zero_grad()
        logits, ei, do, rc = m(tsrc, tdec, pos, sent, pos, sent)
        loss = ce(logits.view(-1, logits.size(-1)), labels.view(-1))
        if len(ei) != 0 and len(ei) == len(do):
            for i in range(len(ei)):
                loss += rec_w[i] * F.mse_loss(ei[i], do[i])
        if H > 1:
            bs = rc[0].shape[0]
            label = torch.tile(torch.arange(H), [bs * L, 1])
            for l in range(len(rc)):
                loss += ind_w[i] * F.nll_loss(rc[l].view(bs * L, H, H), label)
        loss.backward()
        if step == 0:
            out["loss"] = float(loss.item())
            _grads(m, out)
        tn = torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
        if step == 0:
            out["grad_norm"] = float(tn)
        opt.step()
        if step == 0:
            used = [k for k, p in m.named_parameters() if p.grad is not None]
            for k in used[:30] + ["item_emb.word_emb.weight", "mask_bias", "mask_trans_feat.weight"]:
                out["w1." + k] = dict(m.named_parameters())[k].detach().numpy().copy()
    path = os.path.join(OUT, "superbert_%s.npz" % tag)
    np.savez_compressed(path, **compact(out, keep=("predict", "predict2"), k=96, thresh=600))
    print("wrote", path, "loss", out["loss"], "grad_norm", out["grad_norm"], "none", len(out["grad_none"]), "%.1f KB" % (os.path.getsize(path) / 1024))


_STOSA = []


def gen_stosa(tag, item_size, L, d, H, nl, rec_choice, ind_choice, B, seed, cand, pvn_weight=0.05, lr=1e-3):
    if not _STOSA:
        _STOSA.append(_import_from("/root/reference/stosa", "supernet"))
        _STOSA.append(sys.modules["modules"])
    sn, modules = _STOSA
    a = Args()
    a.item_size, a.hidden_units, a.maxlen, a.num_users, a.dropout, a.attention_dropout = item_size, d, L, 3, 0.0, 0.0
    a.num_heads, a.num_layers, a.hidden_act, a.initializer_range, a.distance_metric, a.kernel_param = H, nl, "gelu", 0.02, "wasserstein", 1.0
    a.cuda_condition, a.pvn_weight = False, pvn_weight
    m = sn.DisenDistSASupernet(a, np.array(rec_choice), np.array(ind_choice))
    P = seeded_params({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    block, rec_w, ind_w = cand_to_block(rec_choice, ind_choice, cand)
    m.set_choice(block)
    r = np.random.RandomState(seed + 1)
    inp, dec, pos, neg = stosa_batch(r, B, L, item_size)
    t = [torch.from_numpy(x) for x in (inp, dec, pos, neg)]
    uid = torch.zeros(B, dtype=torch.long)
    out = {"seed": seed, "cfg": np.array([item_size, L, d, H, nl, 3]), "rec_choice": np.array(rec_choice), "ind_choice": np.array(ind_choice),
           "cand": np.array(cand), "input_ids": inp, "dec_ids": dec, "pos_ids": pos, "neg_ids": neg, "lr": lr, "pvn_weight": pvn_weight,
           "shared_idx": np.array(m.item_encoder.shared_idx), "shared_weights": np.array(m.item_encoder.shared_weights)}
    elu = nn.ELU()
    m.eval()
    with torch.no_grad():
        mo, co, att, margins, enc_in, enc_rec, dec_out = m.finetune(t[0], t[1], uid)
        out["mean_out"], out["cov_out"] = mo.numpy(), co.numpy()
        for i in range(nl):
            out["enc_in_mean_%d" % i], out["enc_in_cov_%d" % i] = enc_in[i][0].numpy(), enc_in[i][1].numpy()
            out["rec_mean_%d" % i], out["rec_cov_%d" % i] = enc_rec[i][0].numpy(), enc_rec[i][1].numpy()
            out["dec_out_mean_%d" % i], out["dec_out_cov_%d" % i] = dec_out[i][0].numpy(), dec_out[i][1].numpy()
        full = lambda mo_, co_: modules.wasserstein_distance_matmul(mo_[:, -1, :], co_[:, -1, :], m.item_mean_embeddings.weight,
                                                                    elu(m.item_cov_embeddings.weight) + 1).numpy()
        out["full_dist"] = full(mo, co)
        cand2 = [1.0 - c for c in cand]
        m.set_choice(cand_to_block(rec_choice, ind_choice, cand2)[0])
        mo2, co2 = m.finetune(t[0], t[1], uid)[:2]
        out["cand2"], out["full_dist2"] = np.array(cand2), full(mo2, co2)
        m.set_choice(block)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=0.0)
    wdist = modules.wasserstein_distance
    for step in range(2):
        mo, co, att, margins, enc_in, enc_rec, dec_out = m.finetune(t[0], t[1], uid)
        pos_mean, neg_mean = m.item_mean_embeddings(t[2]), m.item_mean_embeddings(t[3])
        pos_cov, neg_cov = elu(m.item_cov_embeddings(t[2])) + 1, elu(m.item_cov_embeddings(t[3])) + 1
        pos_mean, pos_cov, neg_mean, neg_cov = (x.view(-1, d) for x in (pos_mean, pos_cov, neg_mean, neg_cov))
        sm, sc = mo.view(-1, d), co.view(-1, d)
        pos_logits, neg_logits, pos_vs_neg = wdist(sm, sc, pos_mean, pos_cov), wdist(sm, sc, neg_mean, neg_cov), wdist(pos_mean, pos_cov, neg_mean, neg_cov)
        istarget = (t[2] > 0).view(-1).float()
        loss = torch.sum(-torch.log(torch.sigmoid(neg_logits - pos_logits + 1e-24)) * istarget) / torch.sum(istarget)
        pvn_loss = pvn_weight * torch.sum(torch.clamp(pos_logits - pos_vs_neg, 0) * istarget) / torch.sum(istarget)
        dec_out.reverse()
        for l in range(nl):
            loss += rec_w[l] * F.mse_loss(enc_in[l][0], dec_out[l][0])
            loss += rec_w[l] * F.mse_loss(enc_in[l][1], dec_out[l][1])
        bs = enc_rec[0][0].shape[0]
        label = torch.tile(torch.arange(H), [bs * L, 1])
        for l in range(nl):
            loss += ind_w[l] * F.nll_loss(enc_rec[l][0].view(bs * L, H, H), label)
            loss += ind_w[l] * F.nll_loss(enc_rec[l][1].view(bs * L, H, H), label)
        loss += pvn_loss
        opt.zero_grad()
        loss.backward()
        if step == 0:
            out["loss"] = float(loss.item())
            _grads(m, out)
            out["grad_norm"] = float(torch.sqrt(sum((p.grad ** 2).sum() for p in m.parameters() if p.grad is not None)))
        opt.step()
        if step == 0:
            used = [k for k, p in m.named_parameters() if p.grad is not None]
            for k in used[:30]:
                out["w1." + k] = dict(m.named_parameters())[k].detach().numpy().copy()
    path = os.path.join(OUT, "superstosa_%s.npz" % tag)
    np.savez_compressed(path, **compact(out, keep=("full_dist", "full_dist2"), k=96, thresh=600))
    print("wrote", path, "loss", out["loss"], "none", len(out["grad_none"]), "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    which = sys.argv[1:] or ["bert", "stosa"]
    if "bert" in which:
        gen_bert("c3", 30, 12, 64, 2, 1, [0, 0.001, 0.01], [0, 0.0005, 0.002], B=3, seed=41, cand=[0.7, 0.2])
        gen_bert("l2", 25, 10, 64, 2, 2, [0, 0.01], [0, 0.002], B=2, seed=42, cand=[0.3, 0.9, 0.6, 0.4])
    if "stosa" in which:
        gen_stosa("c3", 42, 12, 64, 4, 1, [0, 0.001, 0.01], [0, 0.001, 0.01], B=3, seed=51, cand=[0.7, 0.2])
        gen_stosa("l2", 33, 10, 64, 2, 2, [0, 0.01], [0, 0.01], B=2, seed=52, cand=[0.3, 0.9, 0.6, 0.4])
