#!/usr/bin/env python3
"""Evolutionary search of the per-layer (reconstruction, independence) loss weights on the MI355X path -- the entry point
mirroring the reference's sasrec/evolution.py: warm up the weight-sharing supernet with a random candidate per epoch
(SearcherEvolution._train_warmup, :286-316), then evolve a population of candidates scored by validation AUC of the supernet
under that candidate (get_cand_auc :173-180; random init, top-k selection, mutation, crossover :190-283, search :318-360).

    python -m adt_amd.sasrec.evolution --dataset ml-1m --synthetic ml1m --warmup_epochs 2 --search_epochs 2 ...

Same flags and same output file (one JSON object per surviving candidate, written by hand: `jsonlines` is not needed).
The supernet forward/backward/optimizer run in libadt_hip.so (adt_amd/sasrec/supersasrec.py); the population bookkeeping and the
batched candidate evaluation are adt_amd/supersearch.py, shared with the BERT4Rec-ADT and STOSA-ADT searches.
"""
import argparse
import os
import random

import numpy as np
import torch

from ..supersearch import EvolutionSearch, cand_to_block, get_shared, result_name
from . import utils as U
from .supersasrec import SuperSASRecModel, SuperTrainer


def parse_args(argv=None):
    p = argparse.ArgumentParser()      # sasrec/evolution.py:29-58
    p.add_argument("--dataset", required=True)
    p.add_argument("--data_dir", default="data")
    p.add_argument("--synthetic", default=None, help="generate data/<dataset>.txt with this preset when it is missing (ml1m, beauty)")
    p.add_argument("--batch_size", default=256, type=int)
    p.add_argument("--eval_batch_size", default=512, type=int)
    p.add_argument("--lr", default=0.001, type=float)
    p.add_argument("--maxlen", default=200, type=int)
    p.add_argument("--hidden_units", default=64, type=int)
    p.add_argument("--num_layers", default=2, type=int)
    p.add_argument("--num_heads", default=2, type=int)
    p.add_argument("--dropout", default=0.5, type=float)
    p.add_argument("--weight_decay", default=0.001, type=float)
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--clip", default=5, type=float)
    p.add_argument("--sample_size", default=100, type=int)
    p.add_argument("--eval_set", default=-1, type=int)
    p.add_argument("--warmup_epochs", default=200, type=int)
    p.add_argument("--search_epochs", default=500, type=int)
    p.add_argument("--population_num", default=100, type=int)
    p.add_argument("--select_num", default=50, type=int)
    p.add_argument("--m_prob", default=0.1, type=float)
    p.add_argument("--crossover_num", default=25, type=int)
    p.add_argument("--mutation_num", default=25, type=int)
    p.add_argument("--seed", default=2022, type=int)
    p.add_argument("--scale_factor", default=0.5, type=float)
    p.add_argument("--scale_decay_rate", default=0.5, type=float)
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--out_dir", default="res")
    return p.parse_args(argv)


class SearcherEvolution:
    """sasrec/evolution.py:60-360 on adt_amd.supersearch: the supernet and its warm-up step run in libadt_hip.so, candidates are scored
    a chunk at a time (validation AUC of the supernet under each candidate's block choice, one batched pass per chunk)."""

    def __init__(self, args):
        self.args = args
        path = os.path.join(args.data_dir, "%s.txt" % args.dataset)
        if not os.path.exists(path) and args.synthetic:
            from . import synth
            os.makedirs(args.data_dir, exist_ok=True)
            h, _, _ = synth.generate(args.synthetic, 23)
            synth.write(path, h)
        user_train, user_valid, user_test, usernum, itemnum = self.dataset = U.data_partition(args.dataset, args.data_dir)
        sampler = U.PopularSampler(user_train, user_valid, user_test, usernum, itemnum, args.sample_size)
        self.warp = U.WarpDataset(user_train, usernum, itemnum, args.maxlen)
        self.val_ds = U.EvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "val", args.eval_set, True)
        self.test_ds = U.EvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "test", args.eval_set, True)
        # search space (sasrec/evolution.py:93-98)
        self.rec_choice = [0, 0.0001, 0.0005, 0.001, 0.005, 0.01]
        self.ind_choice = [0, 0.0001, 0.0005, 0.001, 0.0015, 0.002]
        torch.manual_seed(args.seed)
        self.model = SuperSASRecModel(usernum, itemnum, self.rec_choice, self.ind_choice, args)
        self.trainer = SuperTrainer(self.model, lr=args.lr, betas=(0.9, 0.999), weight_decay=args.weight_decay, clip=args.clip, seed=args.seed)
        self.search_state = EvolutionSearch(args.num_layers, self.evaluate_candidates, "auc", args.select_num, args.population_num, args.m_prob,
                                            args.crossover_num, args.mutation_num, args.scale_factor)
        self.rng = np.random.RandomState(args.seed)
        self.eval_stats = {}

    @property
    def vis_dict(self):
        return self.search_state.vis_dict

    def evaluate_candidates(self, cands, dataset=None, group=32):
        """Validation metrics of the supernet under every candidate of `cands` (get_cand_auc :173-180, for a whole chunk): each
        validation batch is scored for `group` candidates per pass."""
        ds = self.val_ds if dataset is None else dataset
        shared = [get_shared(self.rec_choice, self.ind_choice, cand_to_block(self.rec_choice, self.ind_choice, c)[0]) for c in cands]
        ranks = [[] for _ in cands]
        ncand = None
        for (u, seq, item_idx), _ in ds.batches(self.args.eval_batch_size):
            item_idx = np.asarray(item_idx)
            ncand = item_idx.shape[1]
            for g0 in range(0, len(cands), group):
                r = self.model.predict_rank_candidates(np.asarray(seq), item_idx, shared[g0:g0 + group], stats=self.eval_stats).cpu().numpy()
                for k in range(r.shape[0]):
                    ranks[g0 + k].append(r[k])
        out = []
        for rk in ranks:
            (ndcg, hr), auc = U.metrics_from_ranks(np.concatenate(rk), ncand, [10])
            out.append({"V_NDCG": float(ndcg[10]), "V_HR": float(hr[10]), "V_AUC": float(auc), "auc": float(auc)})
        return out

    def _train_warmup(self):
        for epoch in range(self.args.warmup_epochs):
            self.trainer.set_choice(self.search_state.sample_random())
            for u, seq, dec, pos, neg in self.warp.epoch_batches(self.args.batch_size, self.rng):
                self.trainer.step(seq, dec, pos, neg)
            print("warmup epoch %d / %d loss %.4f" % (epoch + 1, self.args.warmup_epochs, float(self.trainer.loss())), flush=True)

    def search(self):
        self._train_warmup()
        os.makedirs("./checkpoint", exist_ok=True)
        torch.save(self.model.state_dict(), "./checkpoint/super.pth")
        self.search_state.run(self.args.search_epochs, log=lambda m: print(m, flush=True))
        return self.search_state.write(result_name(self.args.out_dir, self.args), self.rec_choice, self.ind_choice)


def set_rng_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def main(argv=None):
    args = parse_args(argv)
    set_rng_seed(args.seed)
    s = SearcherEvolution(args)
    print("results:", s.search())


if __name__ == "__main__":
    main()
