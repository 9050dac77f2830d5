#!/usr/bin/env python3
"""bench.py -- train sequences/sec of SASRec-ADT (ml-1m shape: seq_len 200, d=64, 2 heads, 2 blocks, batch 256 per
GPU, dropout 0.5, bf16 MFMA operands / fp32 accumulate) on N MI355X.

A "step" is the whole loop body of the reference's sasrec/main.py:143-173 on one batch of 256 sequences per GPU:
forward (encoder + reconstruction decoder + head classifiers), loss assembly, backward, weight-decay term,
gradient all-reduce (N > 1), clip_grad_norm_ and Adam -- nothing skipped, dropout on.  `value`: the synthetic ml-1m-shaped id batches
are resident in HBM before the timed region (a ring of staged batches; per step the host issues one graph launch).  `value_incl_h2d`: the
same steps fed from HOST int arrays through the trainer's pinned ring (SURVEY 8d counts the id hand-over inside the step): the step's
first kernel fetches the batch over PCIe.  `ndcg`: a short deterministic training run scored against the reference's recorded NDCG@10 /
HR@10.  One JSON line on stdout (rank 0).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus 8 ...          # launches its own 8 ranks (torch.distributed.run, 127.0.0.1) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
    python bench.py --force-dp            # one rank, but through the data-parallel path (1-rank RCCL group, bucketed all-reduce)
    python bench.py --gpus 8 --scaling strong     # global batch 256 split over the ranks (what adt_amd/sasrec/main.py trains)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CFG = dict(item_num=3416, maxlen=200, hidden_units=64, num_heads=2, num_layers=2, dropout=0.5, batch=256,
           lambdas1=[0.104292, 0.065892], lambdas2=[0.100833, 0.000607], weight_decay=1e-3, lr=1e-3, clip=5.0)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


class Args:
    pass


def synth_batches(n, B, L, V, seed):
    """ml-1m-shaped id batches: history length ~ lognormal clipped to [20, 2314] (mean ~165) so most
    rows are full, items Zipf(1.0)-popular; layout of WarpDataset.sample_data (sasrec/utils.py:288-307)."""
    r = np.random.RandomState(seed)
    pop = 1.0 / np.arange(1, V + 1)
    pop = pop[r.permutation(V)]
    pop /= pop.sum()
    out = []
    for _ in range(n):
        seq = np.zeros((B, L), np.int32)
        dec = np.zeros((B, L), np.int32)
        pos = np.zeros((B, L), np.int32)
        neg = np.zeros((B, L), np.int32)
        lens = np.clip(np.exp(r.normal(4.6, 0.95, size=B)), 20, 2314).astype(int)
        for b in range(B):
            n_hist = min(int(lens[b]) - 3, L)   # train part minus the item used as last target
            items = r.choice(V, size=n_hist + 1, p=pop) + 1
            seq[b, L - n_hist:] = items[:-1]
            pos[b, L - n_hist:] = items[1:]
            neg[b, L - n_hist:] = r.randint(1, V + 1, size=n_hist)
            dec[b, 1:] = seq[b, :-1]
        out.append((seq, dec, pos, neg))
    return out


def build_model(device, prec):
    import torch
    from adt_amd.sasrec.model import SASRecADT
    a = Args()
    a.device, a.num_heads, a.maxlen, a.num_layers = device, CFG["num_heads"], CFG["maxlen"], CFG["num_layers"]
    a.hidden_units, a.dropout, a.precision = CFG["hidden_units"], CFG["dropout"], prec
    torch.manual_seed(23)
    m = SASRecADT(6040, CFG["item_num"], a)
    for _, p in m.named_parameters():   # sasrec/main.py:95-99
        try:
            torch.nn.init.xavier_normal_(p.data)
        except Exception:
            pass
    m.train()
    return m


def _cpu_share():
    """Threads for the CPU baseline: the cores this process may use -- its affinity mask, capped by a cgroup CPU quota when one is set
    (a one-GPU lease of a shared host gets a share of its cores) -- or ADT_CPU_THREADS."""
    if os.environ.get("ADT_CPU_THREADS"):
        return int(os.environ["ADT_CPU_THREADS"])
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline():
    """The CPU restatement of the identical train step (oracle/csrc/adt_cpu.cpp -> libadt_cpu.so: C++17 / OpenMP, fp32, one task per user
    sequence; pinned to the reference's golden vectors by tests/test_cpu_restatement.py) timed on this host's cores on a bounded sample of the
    same workload: the bench's own batch of 256 sequences, 1 warm-up step, then full train steps (forward + loss + backward + weight-decay term
    + clip + Adam, dropout 0.5) until about 15 s of CPU work have been timed (at least 3 steps).  `cores` = the OpenMP thread count used
    (all host cores).  A reported baseline, not a target."""
    from oracle import cpu_restatement as cr
    from oracle import sasrec_oracle as so
    cfg = so.Cfg(CFG["item_num"], CFG["maxlen"], CFG["hidden_units"], CFG["num_heads"], CFG["num_layers"], CFG["dropout"])
    m = cr.CpuSasrec(CFG["item_num"], CFG["maxlen"], CFG["hidden_units"], CFG["num_heads"], CFG["num_layers"], CFG["dropout"], threads=_cpu_share())
    m.load_params(so.init_params(cfg, 0))
    B = CFG["batch"]
    batch = synth_batches(1, B, CFG["maxlen"], CFG["item_num"], 5)[0]
    m.train_step(batch, CFG["lambdas1"], CFG["lambdas2"], CFG["weight_decay"], seed=1)
    t0 = time.time()
    nst = 0
    while nst < 3 or (time.time() - t0 < 15.0 and nst < 400):
        m.train_step(batch, CFG["lambdas1"], CFG["lambdas2"], CFG["weight_decay"], seed=2 + nst)
        nst += 1
    dt = time.time() - t0
    return {"value": round(B * nst / dt, 2), "unit": "sequences/s", "cores": int(cr.load().adt_cpu_threads()), "host_cpus": os.cpu_count(), "kind": "port", "kind_detail": "restatement",
            "sample": "libadt_cpu.so (C++/OpenMP fp32 restatement of the reference step), %d full train steps at batch %d (same L=200, d=64, 2 blocks, "
                      "dropout 0.5) after 1 warm-up, %.1f s" % (nst, B, dt)}


MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


def _time_us(fn, reps=20, warm=3):
    """Average duration of fn() in microseconds, HIP events on the launch stream (torch's current stream = the C ABI's stream)."""
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def _time_in_step(model, trainer, batch, which, layer, reps=12):
    """Duration (us, median) of ONE launch inside a real training step, in its own context: the C library records two HIP events on the launch
    stream right around that launch (adt_debug_time_launch) while complete eager steps run."""
    import ctypes
    import torch
    lib = model.lib
    lib.adt_debug_time_launch.restype = ctypes.c_int
    lib.adt_debug_time_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    graph = trainer.use_graph
    trainer.use_graph = False
    ts = []
    try:
        for _ in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()          # creates the underlying hipEvents
            torch.cuda.synchronize()
            lib.adt_debug_time_launch(which, layer, ctypes.c_void_p(e0.cuda_event), ctypes.c_void_p(e1.cuda_event))
            trainer.step(*batch)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
    finally:
        lib.adt_debug_time_launch(0, 0, None, None)
        trainer.use_graph = graph
    return sorted(ts[2:])[len(ts[2:]) // 2]


def roofline_probe(model, trainer, B, seq_per_s, batch):
    """`roofline` of the bench line.  The object itself describes the kernel with the LARGEST share of the step -- the fused attention-block
    backward k_seqtt_attn_pre_bwd (one workgroup per sequence; four launches per step, 24 % of it in profiles/r04_kernel_stats.csv).
    Algorithmic bytes per launch = tokens x (x 256 + dO 256 + residual-path gradient 256 + gx read 256 + gx write 256 fp32; O 128 bf16;
    H log-sum-exp floats; H x 32 B of dropout keep bits) + 3 x 8 KB of private bf16 weight-gradient partials per sequence, divided by its launch
    duration measured live with HIP events recorded by the library right around that launch inside complete training steps
    (adt_debug_time_launch), net of what the event pair itself adds (the same two events recorded with nothing between them, at the same
    place in the same steps: `event_pair_us`); the committed rocprofv3 average is used instead when it is the larger, i.e. the conservative,
    figure.
    `kernels` adds the fused decoder-layer forward (the round-2 dominant kernel), and what north_star asks for by name: achieved HBM GB/s of
    the embedding gather and MFMA utilisation of the attention kernels (causal-aware FLOPs, SURVEY 8d: 4*L(L+1)/2*hd per (b,h) forward,
    2.5x that backward; counter-based matrix-pipe busy fractions: profiles/r04_mfma.json) on the standalone C-ABI kernels; `step` is
    SURVEY 8d's step-level figure, sequences/s x 2.4 MB of ideal fused-step traffic against the HBM peak."""
    import torch
    from adt_amd import ops
    L, d, H = CFG["maxlen"], CFG["hidden_units"], CFG["num_heads"]
    hd = d // H
    T = B * L
    dev = model.dev
    prec, p, sd = model.cfg.prec, CFG["dropout"], model._seed
    ids4 = [model._ids(a) for a in batch]
    # dominant kernel: in-step duration of the encoder instantiation, layer 1 (the first encoder block of the backward)
    us_bwd_raw = _time_in_step(model, trainer, batch, 1, CFG["num_layers"] - 1)
    us_pair = _time_in_step(model, trainer, batch, 3, CFG["num_layers"] - 1)      # the same two events with no launch between them
    us_bwd_blk = us_bwd_raw - us_pair
    us_prof = _profile_avg_us("k_seqtt_attn_pre_bwd<32, 1, false>")
    us_used = max(us_bwd_blk, us_prof or 0.0)
    blk_bytes = T * (5 * 256 + 128 + H * 4 + H * 32) + B * 3 * 8192       # (the weight-gradient partials are bf16 since round 4)
    blk_flops = T * 9 * 2 * d * d + B * H * 10 * (L * (L + 1) // 2) * hd      # 3 recomputed + 3 weight-gradient + 3 input-gradient 64x64 products per token; five causal score-sized products per head
    achieved = blk_bytes / (us_used * 1e-6) / 1e9
    # the fused decoder-layer forward: back-to-back relaunches, and in context (a whole forward before every timed launch)
    model.run_forward(*ids4, B, True)
    us_dec_iso = _time_us(lambda: model.probe_dec_layer_forward(ids4[1], B, 1))
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e0, e1 in pairs:
        model.run_forward(*ids4, B, True)
        e0.record()
        model.probe_dec_layer_forward(ids4[1], B, 1)
        e1.record()
    torch.cuda.synchronize()
    us_dec = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in pairs)[len(pairs) // 2]
    us_dec_prof = _profile_avg_us("k_seqtt_dec_fwd")
    us_dec_used = max(us_dec, us_dec_prof or 0.0)
    dec_bytes = T * (3 * 256 + 6 * 128 + 256 + 2 * H * 4 + 2 * H * 32 + 4)
    dec_flops = T * 10 * 2 * d * d + 2 * B * H * 4 * (L * (L + 1) // 2) * hd      # ten 64x64 products per token + two causal attentions
    qkv = torch.randn(T, 3 * d, device=dev)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    mask = torch.zeros(B * H * L * 8, device=dev, dtype=torch.int32)   # dropout keep bits, as inside the step
    O, LSE = ops.attn_fwd(prec, q, k, v, B, H, L, True, p, sd, 16, 0, mask)
    dO = torch.randn(T, d, device=dev)
    us_bwd = _time_us(lambda: ops.attn_bwd(prec, q, k, v, O, LSE, dO, B, H, L, True, p, sd, 16, 0, mask))
    us_fwd = _time_us(lambda: ops.attn_fwd(prec, q, k, v, B, H, L, True, p, sd, 16, 0, mask))
    ids = torch.randint(1, CFG["item_num"] + 1, (T,), device=dev, dtype=torch.int32)
    E, Pt = model.flat[:(CFG["item_num"] + 1) * d].view(-1, d), model.flat[model.offsets[1]:model.offsets[1] + L * d].view(L, d)
    us_emb = _time_us(lambda: ops.embed_fwd(ids, E, Pt, L, p, sd, 1, 0))
    fl_fwd = B * H * 4 * (L * (L + 1) // 2) * hd           # QK^T and PV, causal half
    emb_bytes = T * (4 + d * 4 + d * 4)                     # id + fp32 table row + fp32 output row (positional table stays in cache)
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes of this same command,
    # corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py): a profiler measurement, committed under profiles/
    traffic = dec_traffic = None
    for name, key in (("r04_attn_pre_bwd_pmc.json", "blk"), ("r04_dec_fwd_pmc.json", "dec"), ("r03_attn_pre_bwd_pmc.json", "blk"), ("r03_dec_fwd_pmc.json", "dec")):
        pmc = os.path.join(REPO, "profiles", name)
        if os.path.exists(pmc):
            if key == "blk" and traffic is None:
                traffic = round(json.load(open(pmc))["traffic_bytes"])
            if key == "dec" and dec_traffic is None:
                dec_traffic = round(json.load(open(pmc))["traffic_bytes"])
    step_bytes = 2.4e6
    logits_entry = lce_probe(dev)
    mfma_pmc = {}
    for name in ("r04_mfma.json", "r03_mfma.json"):
        try:
            mfma_pmc = json.load(open(os.path.join(REPO, "profiles", name)))["runs"]["flagship"]
            break
        except Exception:
            pass
    # whole-step HBM traffic (every kernel between two k_step_begin launches; tools/pmc_step_traffic.py on the same two PMC passes)
    step_traffic = None
    try:
        step_traffic = round(json.load(open(os.path.join(REPO, "profiles", "r04_step_traffic.json")))["traffic_bytes"])
    except Exception:
        pass

    def busy(name):
        for k, e in mfma_pmc.items():
            if k.startswith(name):
                return round(e.get("mfma_busy_frac", 0.0), 4)
        return None
    return {"bound": "hbm", "kernel": "k_seqtt_attn_pre_bwd (fused attention-block backward, encoder instantiation; the largest share of the step)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_us": round(us_used, 2),
            "avg_launch_us_in_step": round(us_bwd_blk, 2), "avg_launch_us_in_step_raw": round(us_bwd_raw, 2), "event_pair_us": round(us_pair, 2),
            "avg_launch_us_rocprof": us_prof,
            "traffic_source": "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_traffic.py): a committed measurement, not taken in this run",
            "algorithmic_bytes_per_launch": blk_bytes,
            "mfma": {"flops_per_launch": blk_flops, "achieved_TFLOPs": round(blk_flops / (us_used * 1e-6) / 1e12, 2),
                     "mfma_util": round(blk_flops / (us_used * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "mfma_busy_frac_pmc": busy("k_seqtt_attn_pre_bwd<32, 1, false>")},
            "kernels": [
                {"kernel": "k_seqtt_dec_fwd (fused decoder-layer forward)", "bound": "hbm", "avg_launch_us": round(us_dec_used, 2),
                 "avg_launch_us_in_context": round(us_dec, 2), "avg_launch_us_back_to_back": round(us_dec_iso, 2), "avg_launch_us_rocprof": us_dec_prof,
                 "algorithmic_bytes_per_launch": dec_bytes, "achieved_GBps": round(dec_bytes / (us_dec_used * 1e-6) / 1e9, 1),
                 "frac": round(dec_bytes / (us_dec_used * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": dec_traffic, "flops_per_launch": dec_flops,
                 "mfma_util": round(dec_flops / (us_dec_used * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "mfma_busy_frac_pmc": busy("k_seqtt_dec_fwd")},
                {"kernel": "k_embed_fwd (item + positional gather, dropout, pad mask)", "bound": "hbm", "avg_launch_us": round(us_emb, 2),
                 "algorithmic_bytes_per_launch": emb_bytes, "achieved_GBps": round(emb_bytes / (us_emb * 1e-6) / 1e9, 1),
                 "frac": round(emb_bytes / (us_emb * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                {"kernel": "k_attn_fwd (standalone attention, C ABI)", "bound": "mfma", "avg_launch_us": round(us_fwd, 2), "flops_per_launch": fl_fwd,
                 "achieved_TFLOPs": round(fl_fwd / (us_fwd * 1e-6) / 1e12, 2), "mfma_util": round(fl_fwd / (us_fwd * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                 "mfma_busy_frac_pmc": busy("k_attn_fwd_bf16")},
                {"kernel": "k_seq_attn_bwd (standalone attention backward, C ABI)", "bound": "mfma", "avg_launch_us": round(us_bwd, 2),
                 "flops_per_launch": int(2.5 * fl_fwd), "achieved_TFLOPs": round(2.5 * fl_fwd / (us_bwd * 1e-6) / 1e12, 2),
                 "mfma_util": round(2.5 * fl_fwd / (us_bwd * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "mfma_busy_frac_pmc": busy("k_seq_attn_bwd")}] + ([logits_entry] if logits_entry else []),
            "step": {"bytes_per_sequence_ideal": step_bytes, "achieved_GBps": round(seq_per_s * step_bytes / 1e9, 1),
                     "frac": round(seq_per_s * step_bytes / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": step_traffic, "ideal_bytes_per_step": round(step_bytes * B),
                     "traffic_over_ideal": round(step_traffic / (step_bytes * B), 3) if step_traffic else None,
                     "traffic_source": "profiles/r04_step_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE of every kernel of one graph-replayed step, separate --pmc passes)",
                     "mfma_util": round(seq_per_s * 250e6 / 1e12 / MFMA_PEAK_TFLOPS, 4)}}


def _profile_avg_us(kernel):
    """Average duration (us) of `kernel` in the newest committed rocprofv3 kernel-stats summary of the bench step (profiles/rNN_kernel_stats.csv)."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_kernel_stats.csv")), reverse=True):
        if os.path.basename(path).count("_") != 2:      # rNN_kernel_stats.csv only (not the per-backbone summaries)
            continue
        try:
            for row in csv.DictReader(open(path)):
                if kernel in row.get("Name", ""):
                    return round(float(row["AverageNs"]) / 1e3, 2)
        except Exception:
            pass
    return None


def ndcg_check(device):
    """NDCG@10 / HR@10 half of BASELINE.json's metric, in the same run: a seeded DETERMINISTIC training run (dropout 0, the numpy initial
    weights and the seeded batches the reference run used) of the bench's own kernels (bf16, fused trainer, HIP graph) on the 1,200-user
    ml-1m-shaped synthetic set, 30 epochs, scored on the frozen 101-candidate sets -- beside the REFERENCE's figures for exactly that run
    (tests/golden/ref_ndcg_small_det.json, recorded from the imported reference by tools/ref_train_ndcg.py --deterministic).  Tolerance of
    north_star: +-0.01 absolute.  (The stochastic ml-1m-shaped run, dropout 0.5, is tests/test_ndcg_parity.py.)"""
    try:
        from tools.gpu_ndcg_run import run
        from adt_amd.sasrec import synth
        ref = json.load(open(os.path.join(REPO, "tests", "golden", "ref_ndcg_small_det.json")))
        t0 = time.time()
        ours = run("ml1m-small", 30, 30, seed=23, precision="bf16", data=synth.generate("ml1m-small", 23), deterministic=True, device=device)
        eo, er = ours["evals"][-1], ref["evals"][-1]
        return {"workload": "SASRec-ADT d=64 L=200 2 blocks, ml1m-small synthetic (1,200 users, 800 items), 30 epochs, deterministic (dropout 0, shared init and batches), 101 frozen candidates",
                "ndcg10": round(eo["test"]["ndcg10"], 4), "hr10": round(eo["test"]["hr10"], 4), "ref_ndcg10": round(er["test"]["ndcg10"], 4),
                "ref_hr10": round(er["test"]["hr10"], 4), "val_ndcg10": round(eo["val"]["ndcg10"], 4), "ref_val_ndcg10": round(er["val"]["ndcg10"], 4),
                "abs_diff_ndcg10": round(abs(eo["test"]["ndcg10"] - er["test"]["ndcg10"]), 4), "tolerance": 0.01,
                "within_tolerance": bool(abs(eo["test"]["ndcg10"] - er["test"]["ndcg10"]) <= 0.01 and abs(eo["test"]["hr10"] - er["test"]["hr10"]) <= 0.01),
                "seconds": round(time.time() - t0, 1)}
    except Exception as e:       # a measurement extra: never fails the bench line
        print("bench.py: ndcg check skipped (%s)" % e, file=sys.stderr)
        return None


def lce_probe(dev):
    """The MFMA-bound product north_star names besides attention: BERT4Rec-ADT's all-item logits + cross-entropy at BASELINE config 3's shape
    (V + 100 = 26,844 items, d = 256, 5,921 masked rows of a 256 x 200 batch: the mean of the synthetic ml-20m batches), the fused forward +
    backward of adt_lce_fwd_bwd timed with HIP events on the launch stream.  FLOPs: the five 2 M V d products it executes (forward scores,
    score recompute + gradient product for dh and for dE); `useful` counts the three a materialising implementation needs."""
    try:
        import torch
        from adt_amd import ops
        T, V, K, M = 51200, 26844, 256, 5921
        g = torch.Generator(device="cpu").manual_seed(5)
        h = torch.randn(T, K, generator=g).to(dev)
        E = (torch.randn(V, K, generator=g) / K ** 0.5).to(dev)
        b = torch.zeros(V, device=dev)
        rows = torch.zeros(T, dtype=torch.int32)
        rows[:M] = torch.sort(torch.randperm(T, generator=g)[:M]).values.to(torch.int32)
        lab = torch.zeros(T, dtype=torch.int32)
        lab[:M] = torch.randint(1, V, (M,), generator=g).to(torch.int32)
        rows, lab = rows.to(dev), lab.to(dev)
        m_dev = torch.tensor([M], device=dev, dtype=torch.int32)
        inv = torch.tensor([1.0 / M], device=dev)
        loss64 = torch.zeros(64, device=dev)
        dh, dE, db = torch.zeros(T, K, device=dev), torch.zeros(V, K, device=dev), torch.zeros(V, device=dev)
        us = _time_us(lambda: ops.lce_fwd_bwd(h, rows, lab, T, m_dev, E, b, inv, loss64, dh, dE, db), reps=10)
        one = 2.0 * M * V * K
        return {"kernel": "adt_lce_fwd_bwd (BERT4Rec-ADT all-item logits + CE, fused forward + backward, config-3 shape: 5,921 rows x 26,844 items x 256)",
                "bound": "mfma", "avg_launch_us": round(us, 1), "flops_per_launch": int(5 * one), "achieved_TFLOPs": round(5 * one / (us * 1e-6) / 1e12, 1),
                "mfma_util": round(5 * one / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "useful_TFLOPs": round(3 * one / (us * 1e-6) / 1e12, 1)}
    except Exception as e:       # a measurement extra: never fails the bench line
        print("bench.py: lce probe skipped (%s)" % e, file=sys.stderr)
        return None


def deterministic_leg(args):
    """The same step with ADT_ITEM_SORT=1 (item / positional table gradients as sorted segmented sums: no float atomics anywhere, two runs of
    a step give the same bits -- tests/test_hip_model.py::test_step_is_bit_deterministic_with_sorted_item_gradient), in a child process (the
    library reads the switch once).  Reported beside `value`, never as it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--precision", args.precision,
           "--no-cpu-baseline", "--no-ndcg", "--no-roofline"] + (["--no-graph"] if args.no_graph else [])
    try:
        p = subprocess.run(cmd, env=dict(os.environ, ADT_ITEM_SORT="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        return {"env": "ADT_ITEM_SORT=1", "value": d["value"], "ms_per_step": d["ms_per_step"], "value_incl_h2d": d["value_incl_h2d"],
                "ms_per_step_incl_h2d": d["ms_per_step_incl_h2d"], "blocks": d["timing"]["blocks"]}
    except Exception as e:      # a reported extra: never fails the bench line
        return {"env": "ADT_ITEM_SORT=1", "error": repr(e)[:200]}


def visible_gpus():
    """Number of GPUs this process may use, WITHOUT initialising any GPU runtime: the device-visibility variables if set, else the KFD
    topology in sysfs (nodes with a non-zero simd_count are GPUs).  None when neither is readable."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            props = open(os.path.join(root, node, "properties")).read()
            for line in props.splitlines():
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        return n
    except Exception:
        return None


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run on 127.0.0.1 and
    pass their output through.  Runs BEFORE anything in this process touches the GPU; the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel code path even on one rank (1-rank RCCL group)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: 256 sequences per GPU (global batch 256 N); strong: global batch 256 split over the ranks")
    ap.add_argument("--no-ndcg", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        have = visible_gpus()                 # from sysfs / the environment: this parent process never touches the HIP runtime
        if have is not None and have < args.gpus and os.environ.get("ADT_DIST_BACKEND", "nccl") == "nccl":
            sys.exit("bench.py: --gpus %d but this node shows %d GPU(s)" % (args.gpus, have))
        sys.exit(self_launch(args.gpus))
    if args.force_dp:
        os.environ["ADT_FORCE_DP"] = "1"
    # stdout carries exactly ONE line, the JSON result: whatever a library prints to fd 1 (RCCL's version banner when the communicator
    # is created) is routed to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from adt_amd.dp import init_from_env, shard_bounds
    backend = os.environ.get("ADT_DIST_BACKEND", "nccl")      # "gloo": the multi-rank branch on one GPU / on CPU-side collectives (tests/test_bench_dp_gpu.py)
    pg, rank, world, local = init_from_env(backend)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    device = "cuda:%d" % local
    torch.cuda.set_device(local)
    from adt_amd.sasrec.trainer import FusedTrainer
    L, d, H = CFG["maxlen"], CFG["hidden_units"], CFG["num_heads"]

    def timed(fn, k):
        """EXACTLY k steps between barrier + synchronize on both sides; MAX over the ranks."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            fn(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    def blocks(fn, k, min_blocks=5, min_seconds=0.5, max_blocks=200):
        """Blocks of exactly k steps each (every block bracketed like `timed`), at least min_blocks of them and until min_seconds of timed
        steps have run: returns the per-block seconds.  The figure of the line is the MEDIAN block."""
        fixed = int(os.environ.get("ADT_BENCH_BLOCKS", "0"))      # a fixed block count (tests: the step count must not depend on the clock)
        out = []
        while (len(out) < fixed) if fixed > 0 else (len(out) < min_blocks or (sum(out) < min_seconds and len(out) < max_blocks)):
            out.append(timed(fn, k))
        return out

    def run(scaling):
        """One complete measurement at `scaling`: every rank draws the same global batches (same seed) and keeps rows [lo, hi); loss
        normalisers are those of the GLOBAL batch, dropout indices are global (b_offset = lo): N ranks compute the step one process would
        compute on the global batch."""
        GB = CFG["batch"] * (world if scaling == "weak" else 1)      # global batch
        lo, hi = shard_bounds(GB, rank, world)
        B = hi - lo                                                   # this rank's rows of every global batch
        model = build_model(device, args.precision)
        if world > 1:
            if backend == "nccl":
                dist.broadcast(model.flat, 0)
            else:
                t = model.flat.detach().cpu()
                dist.broadcast(t, 0)
                model.flat.copy_(t.to(device))
        tr = FusedTrainer(model, CFG["lambdas1"], CFG["lambdas2"], lr=CFG["lr"], weight_decay=CFG["weight_decay"], clip=CFG["clip"],
                          process_group=pg, use_graph=not args.no_graph and (backend == "nccl" or world == 1), seed=23)
        nb = 4
        gbatches = synth_batches(nb, GB, L, CFG["item_num"], seed=100)
        norms = [(float(np.count_nonzero(b[2])), float(GB * L * d), float(GB * L * H)) for b in gbatches]
        batches = [tuple(a[lo:hi] for a in b) for b in gbatches]
        ring = tr.stage_ring(batches, norms)
        torch.cuda.synchronize()
        for i in range(args.warmup):
            tr.step_staged(ring, b_offset=lo)
        bl = blocks(lambda i: tr.step_staged(ring, b_offset=lo), args.steps)
        loss = float(tr.loss())
        # the same steps from HOST int arrays: packed into a slot of the pinned ring, fetched by the step's first kernel / prefetched
        for i in range(max(3, min(args.warmup, 10))):
            tr.step(*batches[i % nb], norms=norms[i % nb], b_offset=lo)
        bl_h = blocks(lambda i: tr.step(*batches[i % nb], norms=norms[i % nb], b_offset=lo), args.steps)
        return dict(GB=GB, B=B, lo=lo, model=model, tr=tr, batches=batches, bl=bl, bl_h=bl_h, loss=loss)

    def stats(bl, GB):
        med = sorted(bl)[len(bl) // 2]
        return {"ms_per_step": round(med / args.steps * 1e3, 4), "ms_per_step_min": round(min(bl) / args.steps * 1e3, 4),
                "ms_per_step_max": round(max(bl) / args.steps * 1e3, 4), "blocks": len(bl), "timed_s": round(sum(bl), 3),
                "sequences_per_s": round(GB * args.steps / med, 1)}

    main_scaling = args.scaling
    r = run(main_scaling)
    other = None
    if world > 1:      # both scalings in one line: weak (256 sequences per GPU) and strong (the reference's global batch of 256 split over the ranks)
        del r["tr"], r["model"]
        torch.cuda.empty_cache()
        other = run("strong" if main_scaling == "weak" else "weak")
    if rank == 0:
        st, st_h = stats(r["bl"], r["GB"]), stats(r["bl_h"], r["GB"])
        res = {"metric": "train sequences/sec, SASRec-ADT ml-1m (seq_len=200, d=64, 2 blocks)", "value": st["sequences_per_s"],
               "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": st["ms_per_step"], "higher_is_better": True, "scaling": main_scaling, "vs_baseline": None,
               "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": "SASRec-ADT ml-1m shape, 2 blocks d=64 H=2 seq_len=200, global batch %d (%d sequences on this GPU), dropout 0.5, "
                                      "full train step (fwd+loss+bwd+clip+Adam), ids resident in HBM" % (r["GB"], r["B"]),
                          "global_batch": r["GB"], "seq_len": CFG["maxlen"],
                          "parallelism": "dp%d%s" % (world, ("-rccl" if backend == "nccl" else "-" + backend) if pg is not None else ""),
                          "hip_graph": not args.no_graph and (backend == "nccl" or world == 1),
                          "world_size": dist.get_world_size() if pg is not None else 1,
                          "backend": dist.get_backend() if pg is not None else None},
               "value_definition": "median of `blocks` blocks of exactly `steps` steps each (barrier + synchronize around every block, MAX over ranks), "
                                   "id batches resident in HBM; value_incl_h2d = the same from host int arrays through the pinned ring (SURVEY 8d's step)",
               "timing": st, "loss_last_step": round(r["loss"], 5),
               "value_incl_h2d": st_h["sequences_per_s"], "ms_per_step_incl_h2d": st_h["ms_per_step"], "timing_incl_h2d": st_h}
        if other is not None:
            so, so_h = stats(other["bl"], other["GB"]), stats(other["bl_h"], other["GB"])
            res["strong" if main_scaling == "weak" else "weak"] = {
                "global_batch": other["GB"], "sequences_per_gpu": other["B"], "value": so["sequences_per_s"], "ms_per_step": so["ms_per_step"],
                "timing": so, "value_incl_h2d": so_h["sequences_per_s"], "ms_per_step_incl_h2d": so_h["ms_per_step"], "loss_last_step": round(other["loss"], 5)}
        if world == 1 and not args.no_ndcg:
            res["ndcg"] = ndcg_check(device)
        if r["B"] == CFG["batch"] and world == 1 and not args.no_roofline:
            res["roofline"] = roofline_probe(r["model"], r["tr"], r["B"], st["sequences_per_s"], r["batches"][0])
            res["roofline"]["step"]["incl_h2d"] = {"achieved_GBps": round(st_h["sequences_per_s"] * 2.4e6 / 1e9, 1),
                                                   "frac": round(st_h["sequences_per_s"] * 2.4e6 / 1e9 / HBM_PEAK_GBS, 4)}
        if world == 1 and not args.no_roofline and not args.force_dp and os.environ.get("ADT_ITEM_SORT", "0") == "0":
            res["deterministic_mode"] = deterministic_leg(args)
        if world == 1 and not args.no_cpu_baseline and not args.force_dp:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), file=json_out, flush=True)
    if pg is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
