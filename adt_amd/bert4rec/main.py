#!/usr/bin/env python3
"""Entry point of BERT4Rec-ADT on the MI355X path -- the counterpart of the reference's bert4rec/main.py + options.py +
BertTrainer.train (trainer.py:88-160): template override, get_lambda, masked-sequence training set, fused device-side
training step, evaluation of NDCG/HR@{5,10} and AUC on popularity-sampled candidates, model selection by valid AUC.

    python -m adt_amd.bert4rec.main --dataset ml-1m --synthetic ml1m --num_epochs 2
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import datasets as D
from . import utils as U
from .model import BertModel
from .trainer import FusedBertTrainer


def parse_args(argv=None):
    p = argparse.ArgumentParser()       # bert4rec/options.py:7-64
    p.add_argument("--dataset", default="ml-1m")
    p.add_argument("--data_dir", default="data")
    p.add_argument("--synthetic", default=None)
    p.add_argument("--dataset_random_seed", type=int, default=23)
    p.add_argument("--eval_set_size", type=int, default=-1)
    p.add_argument("--topk", type=int, default=-1)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--eval_batch_size", type=int, default=512)
    p.add_argument("--eval_negative_sample_size", type=int, default=100)
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--weight_decay", type=float, default=0.001)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("--clip", type=int, default=5)
    p.add_argument("--eval_interval", type=int, default=1)
    p.add_argument("--dupe_factor", type=int, default=10)
    p.add_argument("--prop_sliding_window", type=float, default=0.1)
    p.add_argument("--type_vocab_size", type=int, default=2)
    p.add_argument("--initializer_range", type=float, default=0.02)
    p.add_argument("--maxlen", type=int, default=200)
    p.add_argument("--hidden_units", type=int, default=64)
    p.add_argument("--inner_units", type=int, default=128)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("--num_heads", type=int, default=2)
    p.add_argument("--dropout", type=float, default=0.2)
    p.add_argument("--attention_dropout", type=float, default=0.2)
    p.add_argument("--mask_prob", type=float, default=0.2)
    p.add_argument("--template", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True)
    p.add_argument("--override", default=None, help="JSON applied after the template")
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--use_graph", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=True)
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    from ..dp import init_from_env, shard_bounds, skip_batch
    pg, rank, world, local = init_from_env("nccl")     # torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE
    if pg is not None:
        args.device = "cuda:%d" % local
    if args.template:
        args = U.set_template(args)
    if args.override:
        for k, v in json.loads(args.override).items():
            setattr(args, k, v)
    lam = U.get_lambda(args.dataset, args.topk)
    if lam is None:
        raise SystemExit("no lambdas for dataset %r" % args.dataset)
    lambda1, lambda2 = lam
    path = os.path.join(args.data_dir, "%s.txt" % args.dataset)
    if not os.path.exists(path) and args.synthetic:
        from ..sasrec import synth
        if rank == 0:
            os.makedirs(args.data_dir, exist_ok=True)
            h, _, _ = synth.generate(args.synthetic, 23)
            synth.write(path, h)
        if world > 1:
            torch.distributed.barrier()
    user_train, user_valid, user_test, usernum, itemnum = D.data_partition(args.dataset, args.data_dir)
    for u in user_train:          # the training sequences include the validation item (bert4rec/trainer.py:165-167)
        user_train[u] = list(user_train[u]) + list(user_valid.get(u, []))
    train_ds = D.BertTrainDataset(user_train, usernum, itemnum, args.maxlen, args.mask_prob, args.dataset_random_seed, args.dupe_factor,
                                  args.prop_sliding_window)
    sampler = D.PopularSampler(user_train, user_valid, user_test, usernum, itemnum, args.eval_negative_sample_size)
    val_ds = D.BertEvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "val", args.eval_set_size)
    test_ds = D.BertEvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "test", args.eval_set_size)
    torch.manual_seed(23)
    model = BertModel(usernum, itemnum, args)
    if world > 1:
        torch.distributed.broadcast(model.flat, 0)
    trainer = FusedBertTrainer(model, lambda1, lambda2, lr=args.lr, betas=(0.9, 0.999), weight_decay=args.weight_decay, clip=args.clip,
                               process_group=pg, use_graph=args.use_graph, seed=23)
    rng = np.random.RandomState(23)
    best = dict(score=0.0, epoch=0, valid=None, test=None, auc_valid=0.0, auc_test=0.0)
    T, nseq = 0.0, 0
    for epoch in range(args.num_epochs):
        t0 = time.time()
        for src, dec, lab in train_ds.epoch_batches(args.batch_size, rng):
            if len(src) != args.batch_size or skip_batch(len(src), world):
                continue      # keep one captured graph shape (the reference's last partial batch is < 0.5 % of an epoch)
            if pg is None:
                trainer.step(src, dec, lab)
            else:             # every rank draws the same global batch (same seed) and trains on its contiguous rows, with the
                lo, hi = shard_bounds(len(src), rank, world)     # GLOBAL normalisers: label count, B*L*d, B*L*H of the whole batch
                trainer.step(src[lo:hi], dec[lo:hi], lab[lo:hi], n_valid_global=int(np.count_nonzero(lab)), b_offset=lo,
                             norms_scale=len(src) / float(hi - lo))
            nseq += len(src)
        torch.cuda.synchronize()
        T += time.time() - t0
        if (epoch + 1) % args.eval_interval == 0 or epoch + 1 == args.num_epochs:
            t_test, auc_test = trainer.evaluate(test_ds.batches(args.eval_batch_size))
            t_valid, auc_valid = trainer.evaluate(val_ds.batches(args.eval_batch_size))
            for k in (5, 10) if rank == 0 else ():
                print("epoch: %d, time: %f, valid (NDCG@%d: %.4f, HR@%d: %.4f, AUC: %s), test (NDCG@%d: %.4f, HR@%d: %.4f, AUC: %s)"
                      % (epoch + 1, T, k, t_valid[0][k], k, t_valid[1][k], auc_valid, k, t_test[0][k], k, t_test[1][k], auc_test))
            loss = float(trainer.loss())      # a collective under data parallelism: every rank calls it
            if rank == 0:
                print(json.dumps({"epoch": epoch + 1, "train_seconds": T, "sequences_per_sec": nseq / max(T, 1e-9), "loss": loss,
                                  "n_gpus": world, "valid": {"ndcg10": t_valid[0][10], "hr10": t_valid[1][10], "auc": auc_valid},
                                  "test": {"ndcg10": t_test[0][10], "hr10": t_test[1][10], "auc": auc_test}}), flush=True)
            if auc_valid >= best["score"]:
                best.update(score=auc_valid, epoch=epoch, valid=t_valid, test=t_test, auc_valid=auc_valid, auc_test=auc_test)
    if pg is not None:
        torch.distributed.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
