#!/usr/bin/env python3
"""Train the REFERENCE SASRec-ADT (imported read-only from /root/reference/sasrec, PyTorch CPU) on the seeded
synthetic ml-1m-shaped dataset and record NDCG@10 / HR@10 / AUC on frozen candidate sets.  Build container only;
the recorded JSON (tests/golden/ref_ndcg_*.json) is what travels.  The loop body is the reference's
sasrec/main.py:141-173 driven with the same torch calls; data comes from the reference's own data_partition /
WarpDataset (DataLoader, shuffle), evaluation from the reference's evaluate_loader on candidate sets frozen with
adt_amd.sasrec.utils.EvalDataset(frozen=True, seed=23) so that both implementations rank the same items.

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_train_ndcg.py --preset ml1m --epochs 30 --out tests/golden/ref_ndcg_ml1m.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/sasrec")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from adt_amd.sasrec import synth  # noqa: E402
from adt_amd.sasrec import utils as our_utils  # noqa: E402


class Args:
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="ml1m")
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--eval_every", type=int, default=10)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--maxlen", type=int, default=200)
    ap.add_argument("--deterministic", action="store_true",
                    help="dropout 0, numpy initial weights oracle.sasrec_oracle.init_params(cfg, 23) and seeded batches from "
                         "adt_amd.sasrec.utils.WarpDataset.epoch_batches: the HIP run (tools/gpu_ndcg_run.py --deterministic) sees the same")
    ap.add_argument("--seed", type=int, default=23, help="the reference's set_rng_seed value (init, batch order, dropout); the data file and "
                                                         "the frozen evaluation candidates stay at seed 23, so every seed ranks the same items")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    import model as ref_model
    import utils as ref_utils

    hist, _, _ = synth.generate(a.preset, 23)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data"))
    synth.write(os.path.join(tmp, "data", "synth.txt"), hist)
    cwd = os.getcwd()
    os.chdir(tmp)
    user_train, user_valid, user_test, usernum, itemnum = ref_utils.data_partition("synth")
    os.chdir(cwd)
    args = Args()
    args.device, args.num_heads, args.maxlen, args.num_layers, args.hidden_units = "cpu", 2, a.maxlen, 2, a.hidden
    args.dropout, args.weight_decay, args.lr, args.clip, args.batch_size = (0.0 if a.deterministic else 0.5), 1e-3, 1e-3, 5.0, 256
    lambdas1, lambdas2 = ref_utils.get_lambdas("ml-1m")

    # reference's seeding (sasrec/main.py:60-66, :71)
    import random
    random.seed(a.seed); np.random.seed(a.seed); torch.manual_seed(a.seed)
    model = ref_model.SASRecADT(usernum, itemnum, args)
    for _, p in model.named_parameters():
        try:
            torch.nn.init.xavier_normal_(p.data)
        except Exception:
            pass
    if a.deterministic:
        from oracle import sasrec_oracle as so
        ocfg = so.Cfg(itemnum, args.maxlen, args.hidden_units, args.num_heads, args.num_layers, dropout=0.0)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in so.init_params(ocfg, seed=23).items()}, strict=True)
    model.train()
    bce = torch.nn.BCEWithLogitsLoss()
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.98))
    ds = ref_utils.WarpDataset(user_train, usernum, itemnum, args.maxlen)
    loader = DataLoader(ds, batch_size=args.batch_size, num_workers=4, shuffle=True)
    our_warp = our_utils.WarpDataset(user_train, usernum, itemnum, args.maxlen)

    def det_batches(epoch):
        for u, seq, dec, pos, neg in our_warp.epoch_batches(args.batch_size, np.random.RandomState(1000 + epoch)):
            yield [u, seq, dec, pos, neg], None

    sampler = our_utils.PopularSampler(user_train, user_valid, user_test, usernum, itemnum, 100)
    evals = {}
    for mode in ("val", "test"):
        ed = our_utils.EvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, mode=mode, frozen=True, seed=23)
        evals[mode] = [((torch.from_numpy(u), torch.from_numpy(s), torch.from_numpy(c.astype(np.int64))), torch.from_numpy(l))
                       for (u, s, c), l in ed.batches(512)]

    log = {"seed": a.seed, "deterministic": bool(a.deterministic), "preset": a.preset, "users": usernum, "items": itemnum, "hidden": a.hidden, "maxlen": a.maxlen, "evals": [], "loss": []}
    t0 = time.time()
    for epoch in range(a.epochs):
        tot, nb = 0.0, 0
        for batch, _ in (det_batches(epoch) if a.deterministic else loader):
            u, seq, dec, pos, neg = [np.array(x) for x in batch]
            pos_logits, neg_logits, enc_in, dec_out, rec_ind = model(u, seq, dec, pos, neg)
            pos_labels, neg_labels = torch.ones(pos_logits.shape), torch.zeros(neg_logits.shape)
            opt.zero_grad()
            indices = np.where(pos != 0)
            loss = bce(pos_logits[indices], pos_labels[indices])
            loss += bce(neg_logits[indices], neg_labels[indices])
            for i in range(len(enc_in)):
                loss += lambdas1[i] * F.mse_loss(enc_in[i], dec_out[i])
            if args.num_heads > 1:
                bs = rec_ind[0].shape[0]
                label = torch.tile(torch.arange(args.num_heads), [bs * args.maxlen, 1])
                for l in range(len(rec_ind)):
                    loss += lambdas2[i] * F.nll_loss(rec_ind[l].view(bs * args.maxlen, args.num_heads, args.num_heads), label)
            for prm in model.item_emb.parameters():
                loss += args.weight_decay * torch.norm(prm)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip)
            opt.step()
            tot += float(loss.item()); nb += 1
        log["loss"].append(tot / nb)
        print("epoch %d loss %.4f (%.0fs)" % (epoch + 1, tot / nb, time.time() - t0), flush=True)
        if (epoch + 1) % a.eval_every == 0 or epoch + 1 == a.epochs:
            model.eval()
            rec = {"epoch": epoch + 1}
            for mode in ("val", "test"):
                (ndcg, hr), auc = ref_utils.evaluate_loader(model, evals[mode], args, mode, [5, 10])
                rec[mode] = {"ndcg10": ndcg[10], "hr10": hr[10], "ndcg5": ndcg[5], "hr5": hr[5], "auc": float(auc)}
            model.train()
            log["evals"].append(rec)
            print(rec, flush=True)
            with open(a.out, "w") as f:
                json.dump(log, f, indent=1)
    log["train_seconds"] = time.time() - t0
    with open(a.out, "w") as f:
        json.dump(log, f, indent=1)


if __name__ == "__main__":
    main()
