#!/usr/bin/env python3
"""GPU: the per-sequence fused layer kernels (ADT_SEQ=1, default) against the staged kernels (ADT_SEQ=0) on the same weights, batch
and dropout seed, bf16 mode: forward tensors, loss and the flat gradient after one FusedTrainer step.  Each arm runs in its own
process (the switch is read once per process).  Usage: python tools/check_seq_vs_staged.py [H] [L] [B]"""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def arm(out, H, L, B):
    import torch
    from oracle import sasrec_oracle as so
    from tests.test_hip_model import build
    from tools.gen_golden_inputs import make_batch
    from adt_amd.sasrec.trainer import FusedTrainer
    from adt_amd.sasrec import model as mm
    cfg = so.Cfg(300, L, 64, H, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), B, cfg.maxlen, cfg.item_num)
    m = build(cfg, P, "bf16", dropout=0.5)
    m.train()
    tr = FusedTrainer(m, [0.104292, 0.065892], [0.100833, 0.000607], lr=1e-3, weight_decay=1e-3, clip=5.0, seed=5)
    tr.step(*batch)
    torch.cuda.synchronize()
    T = B * L
    res = {"loss": float(tr.loss()), "gn": float(tr.grad_norm()), "grad": m.flat_grad.cpu().numpy(), "w": m.flat.cpu().numpy(),
           "pos": m.ws_view(B, mm.WS_POS_LOGITS, 0, T).cpu().numpy()}
    for i in range(3):
        res["enc_x%d" % i] = m.ws_view(B, mm.WS_ENC_X, i, T * 64).cpu().numpy()
        res["dec_x%d" % i] = m.ws_view(B, mm.WS_DEC_X, i, T * 64).cpu().numpy()
    for i in range(2):
        res["rec%d" % i] = m.ws_view(B, mm.WS_REC, i, T * H * H).cpu().numpy()
    np.savez(out, **res)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--arm":
        return arm(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    H, L, B = (int(x) for x in (sys.argv[1:4] + ["2", "200", "6"][len(sys.argv) - 1:]))
    outs = []
    for flag in ("0", "1"):
        out = "/tmp/seq_arm_%s.npz" % flag
        env = dict(os.environ, ADT_SEQ=flag)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--arm", out, str(H), str(L), str(B)], env=env)
        outs.append(np.load(out))
    a, b = outs
    worst = 0.0
    for k in a.files:
        x, y = np.asarray(a[k], np.float64), np.asarray(b[k], np.float64)
        err = np.abs(x - y).max() / max(np.abs(x).max(), 1e-9)
        worst = max(worst, err) if k not in ("w",) else worst
        print("%-8s staged-vs-fused rel err %.3e  (|ref| max %.3e)" % (k, err, np.abs(x).max()))
    print("H=%d L=%d B=%d worst %.3e" % (H, L, B, worst))
    return 0 if worst < 1e-3 else 1


if __name__ == "__main__":
    sys.exit(main())
