#!/usr/bin/env python3
"""Entry point of STOSA-ADT on the MI355X path -- the counterpart of the reference's stosa/main.py: flags (:20-55), template
override, get_lambdas (the tables hold up to 3 layers; the first num_layers entries are used, as the reference's loops do),
leave-two-out datasets, fused device-side training step, full-sort evaluation (distance of the last state to every item,
seen items masked, top 40), early stopping on validation MRR, final test with the test rating matrix.

    python -m adt_amd.stosa.main --dataset Beauty --data_dir data/ --synthetic 1 --epochs 2
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import utils as U
from .datasets import DisenDataset, get_user_seqs
from .models import DisenDistSAModel
from .trainer import FusedStosaTrainer, get_full_sort_score


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", default="./data/")
    p.add_argument("--output_dir", default="./experiment/")
    p.add_argument("--dataset", default="Beauty")
    p.add_argument("--synthetic", type=int, default=0, help="write a seeded Beauty-shaped sequence file when the data file is missing")
    p.add_argument("--hidden_units", type=int, default=64)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("--num_heads", type=int, default=2)
    p.add_argument("--attention_dropout", type=float, default=0.5)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--initializer_range", type=float, default=0.02)
    p.add_argument("--maxlen", type=int, default=50)
    p.add_argument("--distance_metric", default="wasserstein", choices=["wasserstein"],
                   help="the reference's default (stosa/main.py); its 'kl' variant (stosa/modules.py:52-70) is not built")
    p.add_argument("--pvn_weight", type=float, default=0.1)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--eval_batch_size", type=int, default=512)
    p.add_argument("--eval_set", type=int, default=-1)
    p.add_argument("--epochs", type=int, default=400)
    p.add_argument("--patience", type=int, default=100)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--topk", type=int, default=-1)
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--use_graph", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True)
    p.add_argument("--override", default=None)
    return p.parse_args(argv)


def _write_synthetic(path, users=22363, items=12101, seed=42):
    """Seeded sequence file with the public shape of Amazon Beauty 5-core: `user item item ...` per line."""
    r = np.random.RandomState(seed)
    pop = 1.0 / np.arange(1, items + 1) ** 0.8
    pop = pop[r.permutation(items)]
    pop /= pop.sum()
    with open(path, "w") as f:
        for u in range(1, users + 1):
            n = int(np.clip(r.geometric(0.18) + 4, 5, 200))
            seq = r.choice(items, size=n, p=pop) + 1
            f.write("%d %s\n" % (u, " ".join(str(int(x)) for x in seq)))


def _evaluate(trainer, ds, matrix, batch_size):
    """Full-sort scores of the whole user set on every rank: under data parallelism rank r sorts batches r, r+W, ... on its GPU
    and the (N, 40) id lists are gathered (a few hundred KB)."""
    def gen():
        for i, (users, inp, dec, pos, neg, ans) in enumerate(ds.epoch_batches(batch_size, shuffle=False)):
            if i % trainer.world == trainer.rank:
                yield inp, matrix[users], ans
    pred, answers = trainer.full_sort(gen())
    if trainer.world > 1:
        parts = [None] * trainer.world
        torch.distributed.all_gather_object(parts, (pred, answers), group=trainer.pg)
        pred, answers = np.concatenate([p for p, _ in parts]), np.concatenate([a for _, a in parts])
    return get_full_sort_score(answers, pred)


def main(argv=None):
    args = parse_args(argv)
    from ..dp import init_from_env, shard_bounds, skip_batch
    pg, rank, world, local = init_from_env("nccl")     # torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE
    args = U.set_template(args)
    if args.override:
        for k, v in json.loads(args.override).items():
            setattr(args, k, v)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    os.makedirs(args.output_dir, exist_ok=True)
    data_file = os.path.join(args.data_dir, args.dataset + ".txt")
    if pg is not None:
        args.device = "cuda:%d" % local
    if not os.path.exists(data_file) and args.synthetic:
        if rank == 0:
            os.makedirs(args.data_dir, exist_ok=True)
            _write_synthetic(data_file)
        if world > 1:
            torch.distributed.barrier()
    user_seq, max_item, valid_matrix, test_matrix, num_users = get_user_seqs(data_file)
    args.item_size, args.num_users, args.mask_id = max_item + 2, num_users, max_item + 1
    lambda1, lambda2 = U.get_lambdas(args.dataset, args.topk)
    lambda1, lambda2 = lambda1[:args.num_layers], lambda2[:args.num_layers]
    train_ds = DisenDataset(args, user_seq, "train", seed=args.seed)
    valid_ds = DisenDataset(args, user_seq, "valid", args.eval_set, seed=args.seed + 1)
    test_ds = DisenDataset(args, user_seq, "test", args.eval_set, seed=args.seed + 2)
    model = DisenDistSAModel(args)
    if world > 1:
        torch.distributed.broadcast(model.flat, 0)
    trainer = FusedStosaTrainer(model, lambda1, lambda2, lr=args.lr, betas=(args.adam_beta1, args.adam_beta2), weight_decay=args.weight_decay,
                                process_group=pg, use_graph=args.use_graph, seed=args.seed)
    ckpt = os.path.join(args.output_dir, "adt-%s-%d-%d-%d.pt" % (args.dataset, args.hidden_units, args.num_layers, args.num_heads))
    best, wait, T, nseq = None, 0, 0.0, 0
    for epoch in range(args.epochs):
        t0 = time.time()
        for users, inp, dec, pos, neg, _ in train_ds.epoch_batches(args.batch_size):
            if len(users) != args.batch_size or skip_batch(len(users), world):
                continue          # one captured graph shape
            if pg is None:
                trainer.step(inp, dec, pos, neg)
            else:                 # same global batch on every rank (same seed); each trains on its rows with the GLOBAL normalisers
                lo, hi = shard_bounds(len(users), rank, world)
                trainer.step(inp[lo:hi], dec[lo:hi], pos[lo:hi], neg[lo:hi], n_target_global=int((np.asarray(pos) > 0).sum()), b_offset=lo,
                             norms_scale=len(users) / float(hi - lo))
            nseq += len(users)
        torch.cuda.synchronize()
        T += time.time() - t0
        scores = _evaluate(trainer, valid_ds, valid_matrix, args.eval_batch_size)
        parts = trainer.loss_parts().cpu().numpy()      # a collective under data parallelism: every rank calls it
        if rank == 0:
            print(json.dumps({"epoch": epoch, "train_seconds": T, "sequences_per_sec": nseq / max(T, 1e-9), "n_gpus": world,
                              "rec_cur_loss": float((trainer._loss_w.cpu().numpy() * parts).sum()), "auc": float(parts[2]), "pvn_loss": float(parts[1]),
                              "valid_HIT@10": scores[4], "valid_NDCG@10": scores[5], "valid_MRR": scores[-1]}), flush=True)
        if best is None or scores[-1] > best:       # EarlyStopping on MRR (stosa/main.py:122-126, utils.py:38-86)
            best, wait = scores[-1], 0
            if rank == 0:
                torch.save(model.state_dict(), ckpt)
        else:
            wait += 1
            if wait >= args.patience:
                break
    if world > 1:
        torch.distributed.barrier()
    model.load_state_dict(torch.load(ckpt))
    valid_scores = _evaluate(trainer, valid_ds, valid_matrix, args.eval_batch_size)
    scores = _evaluate(trainer, test_ds, test_matrix, args.eval_batch_size)
    if rank == 0:
        print("(%s, %s, %s, %s, %s, %s, %s, %s)" % (valid_scores[0], valid_scores[2], valid_scores[3], valid_scores[-1], scores[0], scores[2], scores[3], scores[-1]))
    if pg is not None:
        torch.distributed.destroy_process_group()
    return valid_scores, scores


if __name__ == "__main__":
    main()
