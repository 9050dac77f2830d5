#!/usr/bin/env python3
"""Stand-alone timing of the masked / general attention kernels (adt_attn_masked_fwd/bwd) at the BERT ml-20m and SASRec d=256 shapes,
with and without dropout (the backward regenerates the dropout hash in both of its passes)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from adt_amd import ops  # noqa: E402
from tools.bench_dense import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    seed = torch.tensor([7], device=dev, dtype=torch.int32)
    for name, B, H, L, hd, causal in (("bert ml-20m", 256, 4, 200, 64, False), ("sasrec d=256", 256, 2, 200, 128, True)):
        d = H * hd
        T = B * L
        qkv = torch.randn(T, 3 * d, device=dev)
        Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        ids = torch.randint(1, 100, (T,), device=dev, dtype=torch.int32)
        dO = torch.randn(T, d, device=dev)
        for p in (0.0, 0.2):
            kid = None if causal else ids
            O, LSE = ops.attn_masked_fwd(ops.PREC_BF16, Q, K, V, B, H, L, causal, kid, -1e9, p, seed, 3, 0)
            t_f = timeit(lambda: ops.attn_masked_fwd(ops.PREC_BF16, Q, K, V, B, H, L, causal, kid, -1e9, p, seed, 3, 0))
            t_b = timeit(lambda: ops.attn_masked_bwd(ops.PREC_BF16, Q, K, V, O, LSE, dO, B, H, L, causal, kid, -1e9, p, seed, 3, 0))
            print("%-13s p=%.1f  fwd %7.1f us  bwd %7.1f us" % (name, p, t_f, t_b), flush=True)


if __name__ == "__main__":
    main()
