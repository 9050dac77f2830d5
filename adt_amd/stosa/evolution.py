#!/usr/bin/env python3
"""Entry point of the STOSA-ADT lambda search on the MI355X path -- the counterpart of the reference's stosa/evolution.py:22-78 (flags)
+ searcher.py.

    python -m adt_amd.stosa.evolution --dataset Beauty --data_dir data/ --synthetic 1 --warmup_epochs 2 --search_epochs 2 ...
"""
import argparse
import os
import random

import numpy as np
import torch

from .main import _write_synthetic
from .searcher import SearcherEvolution


def parse_args(argv=None):
    p = argparse.ArgumentParser()          # stosa/evolution.py:22-66
    p.add_argument("--data_dir", default="./data/")
    p.add_argument("--output_dir", default="./experiment_ev/")
    p.add_argument("--dataset", default="Beauty")
    p.add_argument("--synthetic", type=int, default=0, help="write a seeded Beauty-shaped sequence file when the data file is missing")
    p.add_argument("--maxlen", type=int, default=100)
    p.add_argument("--pvn_weight", type=float, default=0.005)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--eval_batch_size", type=int, default=512)
    p.add_argument("--eval_set", type=int, default=-1)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--hidden_units", type=int, default=64)
    p.add_argument("--num_layers", type=int, default=1)
    p.add_argument("--num_heads", type=int, default=4)
    p.add_argument("--hidden_act", default="gelu")
    p.add_argument("--attention_dropout", type=float, default=0.5)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--initializer_range", type=float, default=0.02)
    p.add_argument("--distance_metric", default="wasserstein", choices=["wasserstein"],
                   help="the reference's default (stosa/main.py); its 'kl' variant (stosa/modules.py:52-70) is not built")
    p.add_argument("--kernel_param", type=float, default=1.0)
    p.add_argument("--warmup_epochs", default=200, type=int)
    p.add_argument("--search_epochs", default=50, type=int)
    p.add_argument("--population_num", type=int, default=20)
    p.add_argument("--select_num", type=int, default=10)
    p.add_argument("--m_prob", type=float, default=0.1)
    p.add_argument("--crossover_num", type=int, default=5)
    p.add_argument("--mutation_num", type=int, default=5)
    p.add_argument("--scale_factor", type=float, default=0.5)
    p.add_argument("--scale_decay_rate", type=float, default=0.5)
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--out_dir", default="res")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    random.seed(args.seed)           # utils.set_seed
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    args.data_file = os.path.join(args.data_dir, args.dataset + ".txt")
    if not os.path.exists(args.data_file) and args.synthetic:
        os.makedirs(args.data_dir, exist_ok=True)
        _write_synthetic(args.data_file)
    s = SearcherEvolution(args)
    print("results:", s.search())


if __name__ == "__main__":
    main()
