#!/usr/bin/env python3
"""Retrain entry point -- the counterpart of the reference's sasrec/main.py (same CLI flags, template override,
get_lambdas() surface, best-by-valid-AUC selection, log/args files), with the training step on the HIP hot path.

    python -m adt_amd.sasrec.main --dataset ml-1m --train_dir run1                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 -m adt_amd.sasrec.main --dataset ml-1m --train_dir run1

Extra flags (not in the reference): --data_dir, --precision {bf16,f32}, --loop {fused,reference} (reference =
the reference's own torch loss/backward/clip/Adam loop on the autograd-wired model), --synthetic PRESET
(generate the seeded ml-1m-shaped file when the dataset is not shipped), --no_template, --frozen_eval.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

from .model import SASRecADT
from .trainer import FusedTrainer
from . import utils as U


def str2bool(s):
    if s.lower() not in {"false", "true"}:
        raise ValueError("Not a valid boolean string")
    return s.lower() == "true"


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", required=True)
    p.add_argument("--train_dir", help="output dir", required=True)
    p.add_argument("--batch_size", default=256, type=int)
    p.add_argument("--lr", default=0.001, type=float)
    p.add_argument("--maxlen", help="max length of sequence", default=50, type=int)
    p.add_argument("--hidden_units", default=50, type=int)
    p.add_argument("--num_layers", default=4, type=int)
    p.add_argument("--num_epochs", default=200, type=int)
    p.add_argument("--num_heads", default=1, type=int)
    p.add_argument("--dropout", default=0.5, type=float)
    p.add_argument("--clip", default=5.0, type=float)
    p.add_argument("--sample_size", help="sample size of negative candidates", default=100, type=int)
    p.add_argument("--device", default="cuda", type=str)
    p.add_argument("--inference_only", default=False, type=str2bool)
    p.add_argument("--state_dict_path", default=None, type=str)
    p.add_argument("--is_save", default=False, type=str2bool)
    p.add_argument("--weight_decay", default=0, type=float)
    p.add_argument("--eval_interval", help="interval to evaluate the model when training", default=20, type=int)
    p.add_argument("--eval_batch_size", default=512, type=int)
    p.add_argument("--eval_set", help="number of the test set, negative value means all users should be evaluated", default=-1, type=int)
    p.add_argument("--topk", help="use which lambda set", default=-1, type=int)
    # additions
    p.add_argument("--data_dir", default="data")
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--loop", default="fused", choices=["fused", "reference"])
    p.add_argument("--synthetic", default=None, help="generate this adt_amd.sasrec.synth preset as data/<dataset>.txt if missing")
    p.add_argument("--no_template", action="store_true", help="keep the CLI hyper-parameters (do not apply templates/<dataset>.json)")
    p.add_argument("--override", default=None, help="JSON dict applied AFTER the template (e.g. '{\"hidden_units\": 64}')")
    p.add_argument("--frozen_eval", default=True, type=str2bool, help="draw evaluation negatives once (seed 23)")
    p.add_argument("--use_graph", default=True, type=str2bool)
    args = p.parse_args(argv)
    if not args.no_template:
        args = U.set_template(args)          # sasrec/main.py:50: the template silently overrides the CLI
    if args.override:
        for k, v in json.loads(args.override).items():
            setattr(args, k, v)
    return args


def set_rng_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def main(argv=None):
    args = parse_args(argv)
    from ..dp import global_norms, init_from_env, shard_bounds, skip_batch
    # torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE; ADT_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
    pg, rank, world, local = init_from_env(os.environ.get("ADT_DIST_BACKEND", "nccl"))
    if args.device == "cuda":
        args.device = "cuda:%d" % local
    out_dir = args.dataset + "_" + args.train_dir
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "args.txt"), "w") as f:
            f.write("\n".join("%s,%s" % (k, v) for k, v in sorted(vars(args).items())))
    set_rng_seed(23)
    lam = U.get_lambdas(args.dataset, args.topk)
    if lam is None:
        raise SystemExit("no lambdas for dataset %r (get_lambdas knows: ml-1m, beauty, steam, ml-20m)" % args.dataset)
    lambdas1, lambdas2 = lam
    args.num_layers = len(lambdas1)        # sasrec/main.py:75
    path = os.path.join(args.data_dir, "%s.txt" % args.dataset)
    if not os.path.exists(path) and args.synthetic:
        from . import synth
        if rank == 0:
            os.makedirs(args.data_dir, exist_ok=True)
            h, _, _ = synth.generate(args.synthetic, 23)
            synth.write(path, h)
        if world > 1:
            torch.distributed.barrier()
    user_train, user_valid, user_test, usernum, itemnum = U.data_partition(args.dataset, args.data_dir)
    sampler = U.PopularSampler(user_train, user_valid, user_test, usernum, itemnum, args.sample_size)
    warp = U.WarpDataset(user_train, usernum, itemnum, args.maxlen)
    val_ds = U.EvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "val", args.eval_set, args.frozen_eval)
    test_ds = U.EvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "test", args.eval_set, args.frozen_eval)

    wide = args.hidden_units != 64     # the fused executor is 64-wide; other widths (the d = 256 template) take the general kernels
    if wide:
        from .model_wide import SASRecADTWide, WideSasrecTrainer
        model = SASRecADTWide(usernum, itemnum, args)
    else:
        model = SASRecADT(usernum, itemnum, args).to(args.device)
    for _, prm in model.named_parameters():       # sasrec/main.py:95-99
        try:
            torch.nn.init.xavier_normal_(prm.data)
        except Exception:
            pass
    if world > 1:
        torch.distributed.broadcast(model.flat, 0)
    epoch_start = 1
    if args.state_dict_path is not None:          # sasrec/main.py:104-114
        model.load_state_dict(torch.load(args.state_dict_path, map_location=torch.device(args.device)))
        tail = args.state_dict_path[args.state_dict_path.find("epoch=") + 6:]
        epoch_start = int(tail[:tail.find(".")]) + 1
    model.train()
    ks = [5, 10]
    logf = open(os.path.join(out_dir, "log.txt"), "w") if rank == 0 else None
    trainer = None
    if wide and args.loop == "fused":
        trainer = WideSasrecTrainer(model, lambdas1, lambdas2, lr=args.lr, betas=(0.9, 0.98), weight_decay=args.weight_decay, clip=args.clip,
                                    process_group=pg, use_graph=args.use_graph, seed=23)
    elif args.loop == "fused":
        trainer = FusedTrainer(model, lambdas1, lambdas2, lr=args.lr, betas=(0.9, 0.98), weight_decay=args.weight_decay, clip=args.clip,
                               process_group=pg, use_graph=args.use_graph, seed=23)
    else:
        if pg is not None:
            raise SystemExit("--loop reference is the single-process loop of sasrec/main.py; data-parallel runs use --loop fused")
        bce = torch.nn.BCEWithLogitsLoss()
        opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.98))
    feeder = None
    if isinstance(trainer, FusedTrainer) and warp._native is not None and os.environ.get("ADT_FEEDER", "1") != "0":
        from .trainer import RingFeeder
        feeder = RingFeeder(trainer, warp, rank, world)
    best = dict(score=0.0, epoch=0, valid=None, test=None, auc_valid=0.0, auc_test=0.0)
    T, t0, nseq = 0.0, time.time(), 0
    rng = np.random.RandomState(23)   # every rank draws the same global batches and takes its own shard
    for epoch in range(epoch_start - 1, args.num_epochs):
        if args.inference_only:
            break
        if feeder is not None:       # native sampler -> pinned ring slots two steps ahead; one graph launch per step on this thread
            for n in feeder.epoch(args.batch_size, rng):
                nseq += n
        for u, seq, dec, pos, neg in (() if feeder is not None else warp.epoch_batches(args.batch_size, rng)):
            if pg is not None:   # contiguous shard of the global batch (every rank draws the same batch: same seed)
                if skip_batch(len(u), world):
                    continue        # a trailing batch with fewer sequences than ranks: dropped on every rank alike
                lo, hi = shard_bounds(len(u), rank, world)
                norms = global_norms(pos, args.hidden_units, args.num_heads)
                trainer.step(seq[lo:hi], dec[lo:hi], pos[lo:hi], neg[lo:hi], norms=norms, b_offset=lo)
            elif trainer is not None:
                trainer.step(seq, dec, pos, neg)
            else:      # the reference's loop body, verbatim in structure (sasrec/main.py:146-173)
                import torch.nn.functional as F
                pos_logits, neg_logits, enc_in, dec_out, rec_ind = model(u, seq, dec, pos, neg)
                pos_labels, neg_labels = torch.ones_like(pos_logits), torch.zeros_like(neg_logits)
                opt.zero_grad()
                indices = np.where(pos != 0)
                loss = bce(pos_logits[indices], pos_labels[indices])
                loss += bce(neg_logits[indices], neg_labels[indices])
                for i in range(len(enc_in)):
                    loss += lambdas1[i] * F.mse_loss(enc_in[i], dec_out[i])
                if args.num_heads > 1:
                    bs = rec_ind[0].shape[0]
                    label = torch.tile(torch.arange(args.num_heads), [bs * args.maxlen, 1]).to(args.device)
                    for l in range(len(rec_ind)):
                        loss += lambdas2[i] * F.nll_loss(rec_ind[l].view(bs * args.maxlen, args.num_heads, args.num_heads), label)
                for prm in model.item_emb.parameters():
                    loss += args.weight_decay * torch.norm(prm)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip)
                opt.step()
            nseq += len(u)
        if (epoch + 1) % args.eval_interval == 0 or epoch + 1 == args.num_epochs:
            torch.cuda.synchronize()
            T += time.time() - t0
            model.eval()
            t_test, auc_test = U.evaluate_loader(model, test_ds.batches(args.eval_batch_size), args, "test", ks, process_group=pg)
            t_valid, auc_valid = U.evaluate_loader(model, val_ds.batches(args.eval_batch_size), args, "val", ks, process_group=pg)
            model.train()
            # trainer.loss() sum-reduces the per-rank loss slots when world > 1: EVERY rank calls it (a collective issued by rank 0
            # alone would pair with the other ranks' next gradient all-reduce)
            last_loss = float(trainer.loss()) if trainer is not None else None
            if rank == 0:
                for k in ks:
                    print("epoch: %d, time: %f, valid (NDCG@%d: %.4f, HR@%d: %.4f, AUC: %s), test (NDCG@%d: %.4f, HR@%d: %.4f, AUC: %s)"
                          % (epoch + 1, T, k, t_valid[0][k], k, t_valid[1][k], auc_valid, k, t_test[0][k], k, t_test[1][k], auc_test))
                print(json.dumps({"epoch": epoch + 1, "train_seconds": T, "sequences_per_sec": nseq / max(T, 1e-9),
                                  "valid": {"ndcg10": t_valid[0][10], "hr10": t_valid[1][10], "auc": auc_valid},
                                  "test": {"ndcg10": t_test[0][10], "hr10": t_test[1][10], "auc": auc_test},
                                  "loss": last_loss}), flush=True)
                logf.write(str(t_valid) + " " + str(t_test) + "\n")
                logf.flush()
            if auc_valid >= best["score"]:     # model selection by valid AUC (sasrec/main.py:194-200)
                best.update(score=auc_valid, epoch=epoch, valid=t_valid, test=t_test, auc_valid=auc_valid, auc_test=auc_test)
            if args.is_save and rank == 0:
                fname = "SASRec.epoch=%d.lr=%s.layer=%d.head=%d.hidden=%d.maxlen=%d.pth" % (
                    epoch, args.lr, args.num_layers, args.num_heads, args.hidden_units, args.maxlen)
                torch.save(model.state_dict(), os.path.join(out_dir, fname))
            t0 = time.time()
    if rank == 0 and best["valid"] is not None:
        for k in ks:
            print("epoch: %d, time: %f, valid (NDCG@%d: %.4f, HR@%d: %.4f, AUC: %s), test (NDCG@%d: %.4f, HR@%d: %.4f, AUC: %s)"
                  % (best["epoch"], T, k, best["valid"][0][k], k, best["valid"][1][k], best["auc_valid"], k, best["test"][0][k], k,
                     best["test"][1][k], best["auc_test"]))
    if logf:
        logf.close()
    if pg is not None:
        torch.distributed.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
