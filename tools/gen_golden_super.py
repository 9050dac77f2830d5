#!/usr/bin/env python3
"""Generate tests/golden/super_*.npz by IMPORTING the reference supernet (read-only, /root/reference/sasrec: supersasrec.py,
super_modules.py, base_super_modules.py) in the build container.  Fixtures are data only.  The loss is the loop body of
SearcherEvolution._train_warmup (sasrec/evolution.py:286-316) driven with the same torch calls (evolution.py itself needs
`jsonlines` and the dataset files, so the class cannot be constructed here); dropout is 0.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_super.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/sasrec")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import super_oracle as su  # noqa: E402
from tools.gen_golden_inputs import make_batch  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


class Args:
    pass


def gen(tag, item_num, L, d, H, nl, rec_choice, ind_choice, B, seed, cand, wd=1e-4, lr=1e-3, clip=5.0):
    import supersasrec
    cfg = su.Cfg(item_num, L, d, H, nl, rec_choice, ind_choice)
    P = su.init_params(cfg, seed)
    a = Args()
    a.device, a.num_heads, a.maxlen, a.num_layers, a.hidden_units, a.dropout = "cpu", H, L, nl, d, 0.0
    m = supersasrec.SuperSASRecModel(1, item_num, np.array(rec_choice), np.array(ind_choice), a)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    block, rec_w, ind_w = su.cand_to_block(cfg, cand)
    m.set_choice(block)
    seq, dec, pos, neg = make_batch(np.random.RandomState(seed + 1), B, L, item_num)
    out = {"seed": seed, "cfg": np.array([item_num, L, d, H, nl]), "rec_choice": np.array(rec_choice), "ind_choice": np.array(ind_choice),
           "cand": np.array(cand), "seq": seq, "dec": dec, "pos": pos, "neg": neg, "wd": wd, "lr": lr, "clip": clip,
           "shared_idx": np.array(m.encoder.shared_idx), "shared_weights": np.array(m.encoder.shared_weights)}
    m.eval()
    with torch.no_grad():
        pl, nl_, ei, do, rc = m(np.zeros(B), seq, dec, pos, neg)
        out["pos_logits"], out["neg_logits"] = pl.numpy(), nl_.numpy()
        for i in range(nl):
            out["enc_in_%d" % i], out["dec_out_%d" % i], out["rec_%d" % i] = ei[i].numpy(), do[i].numpy(), rc[i].numpy()
        items = np.random.RandomState(seed + 2).randint(1, item_num + 1, size=(B, 7))
        out["items"] = items
        out["predict"] = m.predict(np.zeros(B), seq, items).numpy()
    m.train()
    bce = torch.nn.BCEWithLogitsLoss()
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=wd)
    for step in range(2):
        pl, nl_, ei, do, rc = m(np.zeros(B), seq, dec, pos, neg)
        pos_labels, neg_labels = torch.ones(pl.shape), torch.zeros(nl_.shape)
        opt.zero_grad()
        indices = np.where(pos != 0)
        loss = bce(pl[indices], pos_labels[indices])
        loss += bce(nl_[indices], neg_labels[indices])
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
            none = []
            for k, p in m.named_parameters():
                if p.grad is None:
                    none.append(k)
                else:
                    out["grad." + k] = p.grad.numpy().copy()
            out["grad_none"] = np.array(none)
        tn = torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
        if step == 0:
            out["grad_norm"] = float(tn)
        opt.step()
        if step == 0:
            used = [k for k, p in m.named_parameters() if p.grad is not None]
            for k in used[:40] + ["item_emb.weight"]:
                out["w1." + k] = dict(m.named_parameters())[k].detach().numpy().copy()
    path = os.path.join(OUT, "super_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", out["loss"], "grad_norm", out["grad_norm"], "none", len(out["grad_none"]), "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    gen("c3", 30, 12, 64, 2, 1, [0, 0.001, 0.01], [0, 0.0005, 0.002], B=3, seed=31, cand=[0.7, 0.2])
    gen("l2", 25, 10, 64, 2, 2, [0, 0.01], [0, 0.002], B=2, seed=32, cand=[0.3, 0.9, 0.6, 0.4])
