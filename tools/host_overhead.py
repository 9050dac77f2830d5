#!/usr/bin/env python3
"""GPU: where the host time of FusedTrainer.step() goes at cfg-A (B 256, L 200).  Prints ms/step of the resident loop (step_staged), of
step() from host arrays, and the host-side duration of every piece of step() (perf_counter around each call, no device sync between
them, averaged over the loop)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from adt_amd.sasrec.trainer import FusedTrainer
    C = bench.CFG
    B, L = C["batch"], C["maxlen"]
    model = bench.build_model("cuda:0", "bf16")
    tr = FusedTrainer(model, C["lambdas1"], C["lambdas2"], lr=C["lr"], weight_decay=C["weight_decay"], clip=C["clip"], use_graph=True, seed=23)
    batches = bench.synth_batches(4, B, L, C["item_num"], seed=100)
    norms = [(float(np.count_nonzero(b[2])), float(B * L * 64), float(B * L * 2)) for b in batches]
    staged = [tr.stage(b, norms=norms[i]) for i, b in enumerate(batches)]
    n = 300
    for i in range(20):
        tr.step_staged(staged[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        tr.step_staged(staged[i % 4])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_res = time.perf_counter() - t0
    print("resident: %.4f ms/step (host enqueue %.4f ms/step)" % (t_res / n * 1e3, t_host / n * 1e3))
    for i in range(20):
        tr.step(*batches[i % 4], norms=norms[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        tr.step(*batches[i % 4], norms=norms[i % 4])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("step():   %.4f ms/step (host enqueue %.4f ms/step)" % (t_all / n * 1e3, t_host / n * 1e3))
    # pieces of step(): slot wait, pack, submit
    acc = {"slot": 0.0, "pack": 0.0, "submit": 0.0}
    from adt_amd import _hostlib
    hl = _hostlib.load()
    torch.cuda.synchronize()
    t00 = time.perf_counter()
    for i in range(n):
        b = batches[i % 4]
        t0 = time.perf_counter()
        _, flat = tr.slot(B)
        t1 = time.perf_counter()
        st = tr._bind(B)
        hl.adt_host_pack_batch(flat.ctypes.data, b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data, b[3].ctypes.data, st["T"], *norms[i % 4])
        t2 = time.perf_counter()
        tr._submit(st, B, 0)
        t3 = time.perf_counter()
        acc["slot"] += t1 - t0
        acc["pack"] += t2 - t1
        acc["submit"] += t3 - t2
    torch.cuda.synchronize()
    tt = time.perf_counter() - t00
    print("pieces (host ms/step): " + ", ".join("%s %.4f" % (k, v / n * 1e3) for k, v in acc.items()) + " ; loop %.4f ms/step" % (tt / n * 1e3))
    ring = tr.stage_ring(batches, norms)
    for i in range(10):
        tr.step_staged(ring)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        tr.step_staged(ring)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("device ring: %.4f ms/step (host %.4f)" % ((time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3))


if __name__ == "__main__":
    main()
