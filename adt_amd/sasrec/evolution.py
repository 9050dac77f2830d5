#!/usr/bin/env python3
"""Evolutionary search of the per-layer (reconstruction, independence) loss weights on the MI355X path -- the entry point
mirroring the reference's sasrec/evolution.py: warm up the weight-sharing supernet with a random candidate per epoch
(SearcherEvolution._train_warmup, :286-316), then evolve a population of candidates scored by validation AUC of the supernet
under that candidate (get_cand_auc :173-180; random init, top-k selection, mutation, crossover :190-283, search :318-360).

    python -m adt_amd.sasrec.evolution --dataset ml-1m --synthetic ml1m --warmup_epochs 2 --search_epochs 2 ...

Same flags and same output file (one JSON object per surviving candidate, written by hand: `jsonlines` is not needed).
The supernet forward/backward/optimizer run in libadt_hip.so (adt_amd/sasrec/supersasrec.py); only the population
bookkeeping below is host Python, as in the reference.
"""
import argparse
import json
import os
import random
import sys
from random import choice

import numpy as np
import torch

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
    def __init__(self, args):
        self.args = args
        self.select_num, self.population_num, self.m_prob = args.select_num, args.population_num, args.m_prob
        self.crossover_num, self.mutation_num, self.num_layers = args.crossover_num, args.mutation_num, args.num_layers
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
        self.memory, self.epoch, self.candidates = [], 0, []
        self.keep_top_k = {self.select_num: []}
        self.vis_dict = {}
        self.scale_factor = args.scale_factor
        self.rng = np.random.RandomState(args.seed)

    # ---- candidates ------------------------------------------------------------------------------------------------------
    @property
    def rec_weights(self):
        return self.trainer.rec_weights

    @property
    def ind_weights(self):
        return self.trainer.ind_weights

    def _set_choice(self, cand):
        self.trainer.set_choice(cand)

    def sample_random(self):
        return [random.random() for _ in range(2 * self.args.num_layers)]

    def stack_random_cand(self, random_func, *, batch_size=10):
        while True:
            cands = [random_func() for _ in range(batch_size)]
            for cand in cands:
                self.vis_dict.setdefault(str(cand), {})
            for cand in cands:
                yield cand

    def get_cand_auc(self, cand):
        self._set_choice(cand)
        self.model.eval()
        t_valid, auc = U.evaluate_loader(self.model, self.val_ds.batches(self.args.eval_batch_size), self.args, "val", ks=[10])
        info = self.vis_dict[str(cand)]
        info["V_NDCG"], info["V_HR"], info["V_AUC"] = float(t_valid[0][10]), float(t_valid[1][10]), float(auc)
        return float(auc)

    def check_cand(self, cand):
        info = self.vis_dict.setdefault(str(cand), {})
        if "visited" in info:
            return False
        info["visited"] = True
        info["auc"] = float(self.get_cand_auc(cand))
        return True

    def get_random(self, population_num):
        cand_iter = self.stack_random_cand(self.sample_random)
        max_iter = (population_num - len(self.candidates) + 1) * 50
        while len(self.candidates) < population_num and max_iter > 0:
            max_iter -= 1
            cand = next(cand_iter)
            if self.check_cand(cand):
                self.candidates.append(cand)

    def update_top_k(self, candidates, *, k, key, reverse=True):
        t = self.keep_top_k[k]
        t += candidates
        t.sort(key=key, reverse=reverse)
        self.keep_top_k[k] = t[:k]

    def get_crossover(self, k, crossover_num):
        res, max_iter = [], crossover_num * 10

        def random_func():
            c1, c2 = choice(self.keep_top_k[k]), choice(self.keep_top_k[k])
            return [choice([i, j]) for i, j in zip(c1, c2)]
        cand_iter = self.stack_random_cand(random_func)
        while len(res) < crossover_num and max_iter > 0:
            max_iter -= 1
            cand = next(cand_iter)
            if self.check_cand(cand):
                res.append(cand)
        return res

    def get_mutation(self, k, mutation_num, m_prob):
        res, max_iter = [], mutation_num * 10

        def random_func():     # differential mutation (sasrec/evolution.py:262-271)
            cand = list(choice(self.keep_top_k[k]))
            for i in range(self.num_layers * 2):
                if np.random.random_sample() < m_prob:
                    cand2, cand3 = list(choice(self.keep_top_k[k])), list(choice(self.keep_top_k[k]))
                    cand[i] = min(1 - 1e-10, max(1e-10, cand[i] + self.scale_factor * (cand2[i] - cand3[i])))
            return cand
        cand_iter = self.stack_random_cand(random_func)
        while len(res) < mutation_num and max_iter > 0:
            max_iter -= 1
            cand = next(cand_iter)
            if self.check_cand(cand):
                res.append(cand)
        return res

    # ---- training / search ----------------------------------------------------------------------------------------------------
    def _train_warmup(self):
        for epoch in range(self.args.warmup_epochs):
            self._set_choice(self.sample_random())
            for u, seq, dec, pos, neg in self.warp.epoch_batches(self.args.batch_size, self.rng):
                self.trainer.step(seq, dec, pos, neg)
            print("warmup epoch %d / %d loss %.4f" % (epoch + 1, self.args.warmup_epochs, float(self.trainer.loss())), flush=True)

    def search(self):
        self._train_warmup()
        os.makedirs("./checkpoint", exist_ok=True)
        torch.save(self.model.state_dict(), "./checkpoint/super.pth")
        self.get_random(self.population_num)
        for _ in range(self.args.search_epochs):
            self.epoch += 1
            self.memory.append(list(self.candidates))
            self.update_top_k(self.candidates, k=self.select_num, key=lambda x: self.vis_dict[str(x)]["auc"])
            mutation = self.get_mutation(self.select_num, self.mutation_num, self.m_prob)
            crossover = self.get_crossover(self.select_num, self.crossover_num)
            self.candidates = mutation + crossover
            self.get_random(self.population_num)
        os.makedirs(self.args.out_dir, exist_ok=True)
        a = self.args
        fname = os.path.join(a.out_dir, "res_%s_lr_%s_reg_%s_warm_%d_search_%d_layers_%d_select_%d_population_%d_cross_%d_mutation_%d.jsonl" % (
            a.dataset, a.lr, a.weight_decay, a.warmup_epochs, a.search_epochs, a.num_layers, a.select_num, a.population_num, a.crossover_num,
            a.mutation_num))
        with open(fname, "w") as f:
            for cand in self.keep_top_k[self.select_num]:
                info = dict(self.vis_dict[str(cand)])
                self._set_choice(cand)
                info["cand"], info["rec"], info["ind"] = str(cand), str([float(x) for x in self.rec_weights]), str([float(x) for x in self.ind_weights])
                f.write(json.dumps(info) + "\n")
        return fname


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
