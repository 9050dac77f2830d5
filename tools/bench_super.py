#!/usr/bin/env python3
"""Supernet throughput on the ml-1m shape (SASRec-ADT search space: d=64, H=2, 2 depths x 36 candidate layers, L=200): warm-up training
steps/s (SuperTrainer.step: four candidate layers per depth, mixed) and candidate evaluations/s of the evolutionary search, one
candidate at a time (the reference's check_cand granularity) against the batched evaluation of adt_amd/supersearch.py.

    python tools/bench_super.py [--batch 256] [--eval-batch 512] [--cands 32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from adt_amd.sasrec.supersasrec import SuperSASRecModel, SuperTrainer  # noqa: E402
from adt_amd.supersearch import cand_to_block, get_shared  # noqa: E402


class Args:
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--eval-batch", type=int, default=512)
    ap.add_argument("--cands", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    C = bench.CFG
    args = Args()
    args.device, args.num_heads, args.maxlen, args.num_layers, args.hidden_units, args.dropout, args.precision = "cuda:0", C["num_heads"], C["maxlen"], C["num_layers"], C["hidden_units"], C["dropout"], a.precision
    rec_choice = [0, 0.0001, 0.0005, 0.001, 0.005, 0.01]
    ind_choice = [0, 0.0001, 0.0005, 0.001, 0.0015, 0.002]
    torch.manual_seed(1)
    m = SuperSASRecModel(6040, C["item_num"], rec_choice, ind_choice, args)
    tr = SuperTrainer(m, lr=1e-3, weight_decay=1e-3, clip=5.0)
    r = np.random.RandomState(3)
    batches = bench.synth_batches(2, a.batch, C["maxlen"], C["item_num"], 7)
    tr.set_choice([float(x) for x in r.rand(2 * C["num_layers"])])
    for i in range(3):
        tr.step(*batches[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.step(*batches[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res = {"supernet_warmup": {"ms_per_step": round(dt * 1e3, 3), "sequences_per_s": round(a.batch / dt, 1), "batch": a.batch, "precision": a.precision}}
    # candidate evaluation: one validation batch (eval-batch users x 101 candidates items)
    seq = batches[0][0][:a.eval_batch] if a.eval_batch <= a.batch else np.tile(batches[0][0], ((a.eval_batch + a.batch - 1) // a.batch, 1))[:a.eval_batch]
    items = r.randint(1, C["item_num"] + 1, size=(a.eval_batch, 101)).astype(np.int32)
    cands = [[float(x) for x in r.rand(2 * C["num_layers"])] for _ in range(a.cands)]
    shared = [get_shared(rec_choice, ind_choice, cand_to_block(rec_choice, ind_choice, c)[0]) for c in cands]

    def one_at_a_time():
        out = []
        for c in cands:
            m.set_choice(cand_to_block(rec_choice, ind_choice, c)[0])
            out.append(m.predict_rank(seq, items)[1])
        return torch.stack(out)

    stats = {}

    def batched(group=16):
        return torch.cat([m.predict_rank_candidates(seq, items, shared[g:g + group], stats=stats) for g in range(0, len(shared), group)])
    ra, rb = one_at_a_time(), batched()
    torch.cuda.synchronize()
    same = bool((ra == rb).all())
    t0 = time.perf_counter()
    one_at_a_time()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stats.clear()
    batched()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res["candidate_evaluation"] = {"candidates": a.cands, "eval_batch": a.eval_batch, "one_at_a_time_ms": round((t1 - t0) * 1e3, 2),
                                   "batched_ms": round((t2 - t1) * 1e3, 2), "speedup": round((t1 - t0) / (t2 - t1), 2), "identical_ranks": same,
                                   "layer_calls_batched": stats.get("layer_calls"), "layer_calls_one_at_a_time": 4 * C["num_layers"] * a.cands}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
