#!/usr/bin/env python3
"""Step-time measurement of the BERT4Rec-ADT and STOSA-ADT training steps (BASELINE.json configs[2] and configs[4] shapes,
one GPU, synthetic ids, HIP graph).  Not the driver's bench (bench.py measures configs[1]); results go to profiles/.

    python tools/bench_wide.py bert  [--steps 20] [--batch 256]
    python tools/bench_wide.py stosa [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class Args:
    pass


def bert_batch(r, B, L, V, mask_prob):
    src = np.zeros((B, L), np.int32)
    dec = np.zeros((B, L), np.int32)
    lab = np.zeros((B, L), np.int32)
    lens = np.clip(np.exp(r.normal(4.6, 0.95, size=B)), 20, L).astype(int)
    for b in range(B):
        n = int(lens[b])
        items = r.randint(1, V + 1, size=n)
        m = r.rand(n) < mask_prob
        m[-1] = True
        dec[b, L - n:] = items
        src[b, L - n:] = np.where(m, V + 1, items)
        lab[b, L - n:] = np.where(m, items, 0)
    return src, dec, lab


def run_bert(args):
    import torch
    from adt_amd.bert4rec.model import BertModel
    from adt_amd.bert4rec.trainer import FusedBertTrainer
    a = Args()
    a.device, a.maxlen, a.num_heads, a.num_layers, a.hidden_units, a.inner_units = "cuda:0", 200, 4, 2, 256, 1024
    a.dropout, a.attention_dropout, a.type_vocab_size, a.precision = 0.5, 0.5, 2, "bf16"
    V = args.items
    torch.manual_seed(23)
    m = BertModel(1, V, a)
    tr = FusedBertTrainer(m, [0.1, 0.05], [0.1, 0.05], weight_decay=1e-4, use_graph=not args.no_graph, seed=23, mcap_frac=args.mcap)
    r = np.random.RandomState(1)
    staged = [tr.stage(*bert_batch(r, args.batch, 200, V, 0.2)) for _ in range(4)]
    for i in range(args.warmup):
        tr.step_staged(staged[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr.step_staged(staged[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    M = float(np.mean([s["M_host"] for s in staged]))
    flops_logits = 3 * 2 * M * 256 * (V + 100)
    print(json.dumps({"workload": "BERT4Rec-ADT ml-20m shape: d=256 H=4 inner=1024 L=200 2+2 layers V+100=%d, batch %d, dropout 0.5, full train step" % (V + 100, args.batch),
                      "ms_per_step": round(dt / args.steps * 1e3, 3), "sequences_per_s": round(args.batch * args.steps / dt, 1), "masked_rows": M,
                      "logits_gemm_tflops_per_step": round(flops_logits / 1e12, 3), "loss": round(float(tr.loss()), 4), "dtype": "bf16"}))


def run_stosa(args):
    import torch
    from adt_amd.stosa.models import DisenDistSAModel
    from adt_amd.stosa.trainer import FusedStosaTrainer
    a = Args()
    a.device, a.item_size, a.maxlen, a.hidden_units, a.num_heads, a.num_layers, a.num_users = "cuda:0", 12103, 100, 64, 4, 1, 22364
    a.dropout, a.attention_dropout, a.pvn_weight, a.precision, a.distance_metric = 0.3, 0.3, 0.005, "bf16", "wasserstein"
    torch.manual_seed(42)
    m = DisenDistSAModel(a)
    tr = FusedStosaTrainer(m, [0.1], [0.1], use_graph=not args.no_graph, seed=42)
    r = np.random.RandomState(2)
    B, L, V = args.batch, 100, a.item_size

    def batch():
        inp = np.zeros((B, L), np.int32)
        pos = np.zeros((B, L), np.int32)
        neg = np.zeros((B, L), np.int32)
        dec = np.zeros((B, L), np.int32)
        lens = np.clip(r.geometric(0.12, size=B) + 3, 4, L)
        for b in range(B):
            n = int(lens[b])
            it = r.randint(1, V, size=n + 1)
            inp[b, L - n:], pos[b, L - n:], neg[b, L - n:] = it[:-1], it[1:], r.randint(1, V, size=n)
            dec[b, 1:] = inp[b, :-1]
        return inp, dec, pos, neg
    staged = [tr.stage(*batch()) for _ in range(4)]
    for i in range(args.warmup):
        tr.step_staged(staged[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr.step_staged(staged[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "STOSA-ADT Beauty shape: d=64 H=4 L=100 1+1 layers item_size=12103, batch %d, dropout 0.3, full train step" % B,
                      "ms_per_step": round(dt / args.steps * 1e3, 3), "sequences_per_s": round(B * args.steps / dt, 1), "loss": round(float(tr.loss()), 4),
                      "dtype": "bf16 MFMA operands (dense layers and Wasserstein attention), fp32 accumulation, statistics and losses"}))


def run_sasrec256(args):
    """SASRec-ADT at the shipped ml-1m template width (sasrec/templates/ml-1m.json: hidden_units 256, 2 heads, maxlen 200)."""
    import torch
    import bench
    from adt_amd.sasrec.model_wide import SASRecADTWide, WideSasrecTrainer
    a = Args()
    a.device, a.num_heads, a.maxlen, a.num_layers, a.hidden_units, a.dropout, a.precision = "cuda:0", 2, 200, 2, 256, 0.5, "bf16"
    torch.manual_seed(23)
    m = SASRecADTWide(6040, 3416, a)
    for _, p in m.named_parameters():
        try:
            torch.nn.init.xavier_normal_(p.data)
        except Exception:
            pass
    tr = WideSasrecTrainer(m, bench.CFG["lambdas1"], bench.CFG["lambdas2"], weight_decay=1e-3, use_graph=not args.no_graph, seed=23)
    batches = bench.synth_batches(4, args.batch, 200, 3416, seed=100)
    for i in range(args.warmup):
        tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "SASRec-ADT ml-1m TEMPLATE width: d=256 H=2 (head size 128) L=200 2+2 layers V=3416, batch %d, dropout 0.5, full train step (H2D of ids included)" % args.batch,
                      "ms_per_step": round(dt / args.steps * 1e3, 3), "sequences_per_s": round(args.batch * args.steps / dt, 1), "loss": round(float(tr.loss()), 4),
                      "dtype": "bf16"}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["bert", "stosa", "sasrec256"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--items", type=int, default=26744)
    ap.add_argument("--mcap", type=float, default=0.3)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    {"bert": run_bert, "stosa": run_stosa, "sasrec256": run_sasrec256}[args.which](args)
