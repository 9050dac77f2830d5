#!/usr/bin/env python3
"""Stand-alone timing of the dense-layer kernels (bf16 operands) in both implementations (tiled / row-streaming):
    python tools/bench_dense.py            # the layer shapes of the d = 256 and BERT ml-20m steps
Prints one line per shape and kernel: microseconds, TFLOP/s, algorithmic GB/s (fp32 activations in and out)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from adt_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    T = int(os.environ.get("BENCH_T", "51200"))
    shapes = [(256, 256, 0, 0.0, False), (256, 768, 0, 0.0, False), (256, 256, 1, 0.5, True), (256, 1024, 2, 0.2, False), (64, 64, 0, 0.3, True),
              (64, 256, 2, 0.3, False), (1024, 256, 0, 0.2, True)]
    if len(sys.argv) > 1:
        shapes = [shapes[int(i)] for i in sys.argv[1:]]
    seed = torch.tensor([123], device=dev, dtype=torch.int32)
    for K, N, act, p, res in shapes:
        X = torch.randn(T, K, device=dev)
        W = torch.randn(N, K, device=dev) / np.sqrt(K)
        b = torch.randn(N, device=dev)
        R = torch.randn(T, N, device=dev) if res else None
        dY = torch.randn(T, N, device=dev)
        dX = torch.empty(T, K, device=dev)
        Y = torch.empty(T, N, device=dev)
        U = None
        if act:
            _, U = ops.dense_fwd(ops.PREC_BF16, X, W, b, act, True)
        flops = 2.0 * T * K * N
        for rows in (False, True):
            ops.dense_rows_enable(rows)
            t_f = timeit(lambda: ops.dense_fwd(ops.PREC_BF16, X, W, b, act, False, p, seed, 3, 0, R, None, Y))
            t_x = timeit(lambda: ops.dense_bwd(ops.PREC_BF16, dY, X, W, None, None, dX, False, act, U, p, seed, 3))
            dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
            t_w = timeit(lambda: ops.dense_bwd(ops.PREC_BF16, dY, X, W, dW, db, None, False, act, U, p, seed, 3))
            by_f = 4.0 * T * (K + N * (2 if res else 1))
            by_x = 4.0 * T * (K + N * (2 if act else 1))
            print("K=%4d N=%4d act=%d p=%.1f res=%d %-6s fwd %7.1f us %6.1f TF/s %6.0f GB/s | dx %7.1f us %6.1f TF/s %6.0f GB/s | dw %7.1f us %6.1f TF/s"
                  % (K, N, act, p, int(res), "rows" if rows else "tiled", t_f, flops / t_f * 1e-6, by_f / t_f * 1e-3, t_x, flops / t_x * 1e-6, by_x / t_x * 1e-3,
                     t_w, flops / t_w * 1e-6), flush=True)
        ops.dense_rows_enable(True)


if __name__ == "__main__":
    main()
