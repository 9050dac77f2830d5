#!/usr/bin/env python3
"""GPU: time the fused all-item logits + cross-entropy path (adt_lce_fwd_bwd) at the BASELINE config-3 shape of BERT4Rec-ADT's output
layer (V + 100 = 26,844 items, d = 256, ~5.9k masked rows of a 256 x 200 batch) against the unfused sequence it replaces
(gather_rows -> dense_fwd -> ce_rows -> dense_bwd -> scatter_rows).  Prints one JSON line; run under rocprofv3 --kernel-trace --stats
for the per-kernel split.  Usage: python tools/bench_lce.py [--rows 5921] [--iters 20] [--no-unfused]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5921)
    ap.add_argument("--T", type=int, default=51200)
    ap.add_argument("--V", type=int, default=26844)
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-unfused", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    r = np.random.RandomState(0)
    T, V, K, M = a.T, a.V, a.K, a.rows
    h = torch.randn(T, K, device=dev)
    E = torch.randn(V, K, device=dev) / K ** 0.5
    b = 0.1 * torch.randn(V, device=dev)
    rows = torch.from_numpy(np.sort(r.choice(T, M, replace=False)).astype(np.int32))
    rows_p = torch.zeros(T, dtype=torch.int32)
    rows_p[:M] = rows
    lab_p = torch.zeros(T, dtype=torch.int32)
    lab_p[:M] = torch.from_numpy(r.randint(1, V, M).astype(np.int32))
    rows_p, lab_p = rows_p.to(dev), lab_p.to(dev)
    m_dev = torch.tensor([M], device=dev, dtype=torch.int32)
    inv = torch.tensor([1.0 / M], device=dev)
    loss64 = torch.zeros(64, device=dev)
    dh, dE, db = torch.zeros(T, K, device=dev), torch.zeros(V, K, device=dev), torch.zeros(V, device=dev)
    mcap = T

    def fused():
        ops.lce_fwd_bwd(h, rows_p, lab_p, mcap, m_dev, E, b, inv, loss64, dh, dE, db)

    ldv = (V + 3) // 4 * 4

    def unfused():
        hm = ops.gather_rows(h, rows_p, mcap, m_dev)
        logits, _ = ops.dense_fwd(1, hm, E, b, t_dev=m_dev, ldy=ldv)
        ops.ce_rows(logits, lab_p, V, inv, loss64, mcap, m_dev)
        dhm = torch.empty_like(hm)
        ops.dense_bwd(1, logits, hm, E, dE, db, dhm, False, t_dev=m_dev)
        ops.scatter_rows(dhm, rows_p, dh, False, mcap, m_dev)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    out = {"rows": M, "V": V, "K": K, "fused_us": round(timeit(fused), 1)}
    flops = 2.0 * M * V * K
    out["gemm_flops_one_pass"] = flops
    out["fused_equiv_TFLOPs_5_passes"] = round(5 * flops / out["fused_us"] / 1e6, 1)
    if not a.no_unfused:
        out["unfused_us"] = round(timeit(unfused), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
