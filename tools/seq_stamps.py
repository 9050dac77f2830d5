#!/usr/bin/env python3
"""Timing aid: s_memtime stamps of workgroup 0 of the fused encoder-layer forward (ADT_SEQ_STAMPS=1), per wave, in cycles since the
kernel's first stamp.  Phases: 1 weights staged, 2/3 in-projection of slot 0/1, 4 barrier, 5/8 attention of slot 0/1, 6/9 out_proj +
residual, 7/10 FFN + store."""
import ctypes
import os
import sys

MODE = sys.argv[1] if len(sys.argv) > 1 else "fwd"          # fwd: fused encoder forward ; attn: per-sequence attention backward
os.environ["ADT_SEQ_STAMPS"] = {"attn": "2", "post": "3"}.get(MODE, "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from adt_amd import _lib  # noqa: E402

m = bench.build_model("cuda:0", "bf16")
batch = bench.synth_batches(1, 256, 200, 3416, 7)[0]
ids = [m._ids(a) for a in batch]
from adt_amd.sasrec.trainer import FusedTrainer  # noqa: E402
tr = FusedTrainer(m, bench.CFG["lambdas1"], bench.CFG["lambdas2"], weight_decay=1e-3, seed=3)
for _ in range(3):
    if MODE in ("attn", "post"):
        tr.step(*batch)
    else:
        m.run_forward(*ids, 256, True)
torch.cuda.synchronize()
NWV = 12 if MODE == "fwd" else 8          # waves per workgroup of the stamped kernel
buf = (ctypes.c_ulonglong * 256)()
rc = _lib.load().adt_seq_stamps_read(buf, 256)
t = np.array(list(buf), dtype=np.int64).reshape(16, 16)[:NWV]
print("rc", rc)
t0 = t[:, 0].min()
names = ["start", "staged", "pre0", "pre1", "barrier", "attn0", "oproj0", "ffn0", "attn1", "oproj1", "ffn1"]
if MODE == "attn":
    names = ["start", "staged", "recomp", "passA", "passB", "dW", "dx"]       # the fused attention-block backward (last launch of the step)
if MODE == "post":      # k_seqtt_post_bwd<., true> (encoder), last launch of the step
    names = ["start", "issued", "staged", "loadsA", "slot0A", "endA", "bar", "dW2", "bar", "endB", "dW1", "bar", "endC", "dWo", "end"]
print("wave " + " ".join("%8s" % n for n in names))
if MODE == "attn":
    print("prologue (loads issued, zero-fill done, small tables stored, images stored):", [[int(t[w, k] - t0) for k in (11, 12, 13, 14)] for w in (0, 4)])
if MODE == "attn":
    print("P4/P5 detail (rows put, colsums flushed, product 1 done, P5 slot-0 products issued, slot 0 stored):", [[int(t[w, k] - t0) for k in (7, 8, 9, 10, 15)] for w in (0, 4)])
if MODE == "fwd":
    print("prologue (issue loads, zero-fill, image stores, vector stores):", [[int(t[w, k] - t0) for k in (11, 12, 13, 14)] for w in (0, 4)])
for w in range(NWV):
    print("%4d " % w + " ".join("%8d" % (t[w, k] - t0 if t[w, k] else -1) for k in range(len(names))))
