#!/usr/bin/env python3
"""Evolutionary search of the per-layer (reconstruction, independence) loss weights of BERT4Rec-ADT on the MI355X path -- the entry
point mirroring the reference's bert4rec/evolution.py: warm up the weight-sharing supernet with one random candidate per epoch
(SearcherEvolution._train_warmup, :253-296), then evolve a population of candidates scored by the validation AUC of the supernet under
each candidate's block choice (get_cand_auc :150-157; random init, top-k, mutation, crossover :170-251; search :298-347).

    python -m adt_amd.bert4rec.evolution --dataset ml-1m --synthetic ml1m-small --warmup_epochs 2 --search_epochs 2 ...

The supernet forward / backward / AdamW run in libadt_hip.so (adt_amd/bert4rec/superbert.py); the population bookkeeping and the batched
candidate evaluation are adt_amd/supersearch.py.  Same flags (bert4rec/options.py) and the same result file.
"""
import argparse
import json
import os
import random

import numpy as np
import torch

from ..sasrec.utils import metrics_from_ranks
from ..supersearch import EvolutionSearch, cand_to_block, get_shared, result_name
from . import datasets as D
from . import utils as U
from .superbert import SuperBertModel, SuperBertTrainer


def parse_args(argv=None):
    p = argparse.ArgumentParser()       # bert4rec/options.py:7-65
    p.add_argument("--dataset", default="ml-1m")
    p.add_argument("--data_dir", default="data")
    p.add_argument("--synthetic", default=None)
    p.add_argument("--dataset_random_seed", type=int, default=23)
    p.add_argument("--eval_set_size", type=int, default=-1)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--eval_batch_size", type=int, default=512)
    p.add_argument("--eval_negative_sample_size", type=int, default=100)
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--weight_decay", type=float, default=0.001)
    p.add_argument("--clip", type=int, default=5)
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
    p.add_argument("--template", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=False)
    p.add_argument("--override", default=None, help="JSON applied after the template")
    p.add_argument("--warmup_epochs", default=200, type=int)
    p.add_argument("--search_epochs", default=500, type=int)
    p.add_argument("--population_num", type=int, default=100)
    p.add_argument("--select_num", type=int, default=50)
    p.add_argument("--m_prob", type=float, default=0.1)
    p.add_argument("--crossover_num", type=int, default=25)
    p.add_argument("--mutation_num", type=int, default=25)
    p.add_argument("--seed", type=int, default=2022)
    p.add_argument("--scale_factor", type=float, default=0.5)
    p.add_argument("--scale_decay_rate", type=float, default=0.5)
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--out_dir", default="res")
    return p.parse_args(argv)


class SearcherEvolution:
    def __init__(self, args):
        self.args = args
        path = os.path.join(args.data_dir, "%s.txt" % args.dataset)
        if not os.path.exists(path) and args.synthetic:
            from ..sasrec import synth
            os.makedirs(args.data_dir, exist_ok=True)
            h, _, _ = synth.generate(args.synthetic, 23)
            synth.write(path, h)
        user_train, user_valid, user_test, usernum, itemnum = self.dataset = D.data_partition(args.dataset, args.data_dir)
        self.train_ds = D.BertTrainDataset(user_train, usernum, itemnum, args.maxlen, args.mask_prob, args.dataset_random_seed, args.dupe_factor,
                                           args.prop_sliding_window)
        sampler = D.PopularSampler(user_train, user_valid, user_test, usernum, itemnum, args.eval_negative_sample_size)
        self.val_ds = D.BertEvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "val", args.eval_set_size)
        self.test_ds = D.BertEvalDataset(user_train, user_valid, user_test, usernum, itemnum, args.maxlen, sampler, "test", args.eval_set_size)
        # search space (bert4rec/evolution.py:71-77)
        self.rec_choice = [0, 0.0001, 0.0005, 0.001, 0.005, 0.01]
        self.ind_choice = [0, 0.0001, 0.0005, 0.001, 0.0015, 0.002]
        torch.manual_seed(args.seed)
        self.model = SuperBertModel(usernum, itemnum, self.rec_choice, self.ind_choice, args)
        self.trainer = SuperBertTrainer(self.model, lr=args.lr, betas=(0.9, 0.999), weight_decay=args.weight_decay, clip=args.clip, seed=args.seed)
        self.search_state = EvolutionSearch(args.num_layers, self.evaluate_candidates, "auc", args.select_num, args.population_num, args.m_prob,
                                            args.crossover_num, args.mutation_num, args.scale_factor)
        self.rng = np.random.RandomState(args.seed)
        self.eval_stats = {}

    @property
    def vis_dict(self):
        return self.search_state.vis_dict

    def evaluate_candidates(self, cands, dataset=None, group=16):
        """Validation metrics (evaluate_loader, bert4rec/utils.py; ks = [10]) of the supernet under every candidate of `cands`: each
        validation batch is scored for `group` candidates per pass."""
        ds = self.val_ds if dataset is None else dataset
        shared = [get_shared(self.rec_choice, self.ind_choice, cand_to_block(self.rec_choice, self.ind_choice, c)[0]) for c in cands]
        ranks = [[] for _ in cands]
        ncand = None
        for seq, cand_items in ds.batches(self.args.eval_batch_size):
            ncand = cand_items.shape[1]
            for g0 in range(0, len(cands), group):
                r = self.model.predict_rank_candidates(seq, cand_items, shared[g0:g0 + group], stats=self.eval_stats).cpu().numpy()
                for k in range(r.shape[0]):
                    ranks[g0 + k].append(r[k])
        out = []
        for rk in ranks:
            (ndcg, hr), auc = metrics_from_ranks(np.concatenate(rk), ncand, [10])
            out.append({"V_NDCG": float(ndcg[10]), "V_HR": float(hr[10]), "V_AUC": float(auc), "auc": float(auc)})
        return out

    def _train_warmup(self):
        for epoch in range(self.args.warmup_epochs):
            self.trainer.set_choice(self.search_state.sample_random())
            for src, dec, lab in self.train_ds.epoch_batches(self.args.batch_size, self.rng):
                self.trainer.step(src, dec, lab)
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
    if args.template:
        args = U.set_template(args)
    if args.override:
        for k, v in json.loads(args.override).items():
            setattr(args, k, v)
    set_rng_seed(args.seed)
    s = SearcherEvolution(args)
    print("results:", s.search())


if __name__ == "__main__":
    main()
