#!/usr/bin/env python3
"""GPU diagnostic: run-to-run determinism of one bf16 FusedTrainer step (fresh model + workspace each time) -- forward tensors, loss, gradients."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("ADT_OLD_LIB"):
    from adt_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libadt_hip_old.so")
from oracle import sasrec_oracle as so
from tests.test_hip_model import build
from tools.gen_golden_inputs import make_batch
from adt_amd.sasrec.trainer import FusedTrainer
from adt_amd.sasrec import model as mm
LAM1, LAM2 = [0.104292, 0.065892], [0.100833, 0.000607]
L = int(sys.argv[1]) if len(sys.argv) > 1 else 52
B = 6
cfg = so.Cfg(300, L, 64, 2, 2, dropout=0.5)
P = so.init_params(cfg, seed=3)
batch = make_batch(np.random.RandomState(4), B, cfg.maxlen, cfg.item_num)
junk = []
def run(prec):
    junk.append(torch.full((1 << 22,), float("nan"), device="cuda:0"))      # poison what the next allocations may reuse
    junk.pop()
    m = build(cfg, P, prec, dropout=0.5); m.train()
    tr = FusedTrainer(m, LAM1, LAM2, lr=1e-3, weight_decay=0.0, clip=1e9, seed=5)
    ws = m.workspace(B); ws.fill_(float("nan"))
    tr.step(*batch)
    torch.cuda.synchronize()
    T = B * L
    out = {"pos": m.ws_view(B, mm.WS_POS_LOGITS, 0, T).cpu().numpy().copy(), "loss": np.array([float(tr.loss())])}
    for i in range(3):
        out["enc_x%d" % i] = m.ws_view(B, mm.WS_ENC_X, i, T * 64).cpu().numpy().copy()
        out["dec_x%d" % i] = m.ws_view(B, mm.WS_DEC_X, i, T * 64).cpu().numpy().copy()
    for k, _ in so.param_shapes(cfg):
        out["g." + k] = m.grad_view(k).cpu().numpy().copy()
    return out
a, b = run("bf16"), run("bf16")
for k in a:
    d = np.abs(a[k] - b[k]).max() / max(np.abs(a[k]).max(), 1e-12)
    nan = not np.isfinite(a[k]).all()
    if d > 1e-6 or nan or not k.startswith("g."):
        print("%-60s %.3e %s" % (k, d, "NaN!" if nan else ""))
