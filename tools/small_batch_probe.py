#!/usr/bin/env python3
"""Step time of the fused trainer at small local batches (the strong-scaling regime: global batch 256 over N ranks), with the per-sequence
kernels (one workgroup per sequence) and with the token-parallel staged kernels (ADT_SEQ=0).  One process per setting (the library reads
the switch once).    python tools/small_batch_probe.py            # prints one JSON line per (batch, path)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def arm(B):
    sys.path.insert(0, REPO)
    import time
    import torch
    import bench
    from adt_amd.sasrec.trainer import FusedTrainer
    m = bench.build_model("cuda:0", "bf16")
    tr = FusedTrainer(m, bench.CFG["lambdas1"], bench.CFG["lambdas2"], weight_decay=1e-3, seed=3, use_graph=True)
    batches = bench.synth_batches(4, B, 200, 3416, 7)
    h = tr.stage_ring(batches, None) if hasattr(tr, "stage_ring") else None
    for _ in range(10):
        tr.step_staged(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        tr.step_staged(h)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"batch": B, "per_sequence_kernels": os.environ.get("ADT_SEQ", "1") != "0", "ms_per_step": round(ms, 4), "sequences_per_s": round(B / ms * 1e3, 1)}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--arm":
        arm(int(sys.argv[2]))
    else:
        for B in (256, 128, 64, 32):
            for seq in ("1", "0"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(B)], env=dict(os.environ, ADT_SEQ=seq), check=False)
