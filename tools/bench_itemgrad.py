#!/usr/bin/env python3
"""Timing of the deterministic item-table gradient kernels at the flagship shape (B 256, L 200, V 3,416, Zipf ids): sort, the two segmented-sum
passes, the positional sum -- against the float-atomic scatters they replace (run under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from adt_amd import ops  # noqa: E402

DEV = "cuda:0"
B, L, V = 256, 200, 3416
T, V1 = B * L, V + 1
seq, dec, pos, neg = bench.synth_batches(1, B, L, V, 7)[0]
d_ids = [torch.from_numpy(a.reshape(-1).copy()).to(DEV) for a in (seq, dec, pos, neg)]
rows = [torch.randn(T, 64, device=DEV) for _ in range(3)]
rows.append(rows[2])
coef = [None, None, torch.randn(T, device=DEV), torch.randn(T, device=DEV)]
sd = torch.from_numpy(np.array([5], dtype=np.uint32).view(np.int32)).to(DEV)
dE = torch.zeros(V1, 64, device=DEV)
dE2 = torch.zeros(V1, 64, device=DEV)
dP = torch.zeros(L, 64, device=DEV)


def run():
    work = ops.item_sort(d_ids, V1, rows, coef, [0, 0, 1, 1], 0)
    ops.item_segsum(work, 4, T, V1, 0b1111, [1, 2, 0, 0], 0.5, sd, 8.0, dE)
    ops.posemb_sum(d_ids[:2], rows[:2], [1, 2], B, L, 0.5, sd, 0, dP)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("sort + 2 segsum passes + posemb: %.1f us per iteration" % (e0.elapsed_time(e1) / 20 * 1e3))
