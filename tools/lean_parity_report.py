#!/usr/bin/env python3
"""GPU: per-tensor parity report of the per-sequence fused ("lean" / transposed-tile) kernels -- the kernels bench.py times -- with
dropout ON (p = 0.5, bf16 operands) against (a) the fp32 numpy oracle, (b) the oracle with bf16-rounded matrix operands
(oracle.sasrec_oracle.operands("bf16")) and (c) the staged stage kernels (ADT_SEQ=0, separate process) on the same weights, batch and
dropout seed.  For every output tensor and every parameter gradient: max-norm error relative to the tensor's max magnitude and relative
Frobenius error.  This is the measurement behind the thresholds of tests/test_hip_model.py::test_lean_step_with_dropout_vs_oracle.

    python tools/lean_parity_report.py [--out gpurun_out/lean_parity.json]
"""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

SHAPES = [(2, 200, 8, 3416), (4, 48, 3, 300), (1, 100, 4, 300), (2, 52, 5, 300)]     # (H, L, B, V)
LAM1, LAM2, WD = [0.104292, 0.065892], [0.100833, 0.000607], 1e-3


def gpu_arm(H, L, B, V, out):
    """One FusedTrainer step in this process; writes every compared tensor to `out`."""
    import torch
    from oracle import sasrec_oracle as so
    from tests.test_hip_model import build
    from tools.gen_golden_inputs import make_batch
    from adt_amd.sasrec.trainer import FusedTrainer
    from adt_amd.sasrec import model as mm
    cfg = so.Cfg(V, L, 64, H, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), B, L, V)
    m = build(cfg, P, "bf16", dropout=0.5)
    m.train()
    lean = int(m.lib.adt_seq_layer_supported(1, L, 64, 64 // H))
    tr = FusedTrainer(m, LAM1, LAM2, lr=1e-3, weight_decay=WD, clip=5.0, seed=5)
    tr.step(*batch)
    torch.cuda.synchronize()
    T = B * L
    res = {"lean": lean, "seed": int(m._seed.cpu().numpy().view(np.uint32)[0]), "loss": float(tr.loss()), "gn": float(tr.grad_norm()),
           "pos_logits": m.ws_view(B, mm.WS_POS_LOGITS, 0, T).view(B, L).cpu().numpy(),
           "neg_logits": m.ws_view(B, mm.WS_NEG_LOGITS, 0, T).view(B, L).cpu().numpy()}
    for i in range(2):
        res["enc_in.%d" % i] = m.ws_view(B, mm.WS_ENC_X, i, T * 64).view(B, L, 64).cpu().numpy()
        res["dec_out.%d" % i] = m.ws_view(B, mm.WS_DEC_X, 2 - i, T * 64).view(B, L, 64).cpu().numpy()
        if H > 1:
            res["rec.%d" % i] = m.ws_view(B, mm.WS_REC, i, T * H * H).view(B, L, H, H).cpu().numpy()       # reference row order
    for k, _ in so.param_shapes(cfg):
        res["g." + k] = m.grad_view(k).cpu().numpy()
    np.savez(out, **res)


def errs(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    mx = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-12))
    fro = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-12))
    return mx, fro


def oracle_arm(H, L, B, V, seed, mode):
    from oracle import sasrec_oracle as so
    from tools.gen_golden_inputs import make_batch
    cfg = so.Cfg(V, L, 64, H, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), B, L, V)
    with so.operands(mode):
        out = so.forward(P, cfg, *batch, training=True, seed=seed)
        loss, parts, seeds = so.loss_and_seeds(P, cfg, out, batch[2], LAM1, LAM2, WD)
        G = so.backward(P, cfg, out[5], seeds, WD)
    res = {"loss": loss, "gn": so.grad_norm(G), "pos_logits": out[0], "neg_logits": out[1]}
    for i in range(2):
        res["enc_in.%d" % i] = out[2][i]
        res["dec_out.%d" % i] = out[3][i]
        if H > 1:
            res["rec.%d" % i] = so.rec_reference_order(out[4][i])
    for k, g in G.items():
        if g is not None:
            res["g." + k] = g.reshape(dict(so.param_shapes(cfg))[k])
    return res


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--arm":
        return gpu_arm(*[int(x) for x in sys.argv[2:6]], sys.argv[6])
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(REPO, "gpurun_out", "lean_parity.json")
    report = {}
    for (H, L, B, V) in SHAPES:
        arms = {}
        for flag in ("1", "0"):
            out = "/tmp/lean_arm_%s.npz" % flag
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--arm", str(H), str(L), str(B), str(V), out],
                                  env=dict(os.environ, ADT_SEQ=flag))
            arms[flag] = dict(np.load(out))
        fused, staged = arms["1"], arms["0"]
        assert int(fused["seed"]) == int(staged["seed"])
        o32 = oracle_arm(H, L, B, V, int(fused["seed"]), None)
        o16 = oracle_arm(H, L, B, V, int(fused["seed"]), "bf16")
        tag = "H%d_L%d_B%d" % (H, L, B)
        rows = {}
        print("== %s  lean=%d  loss fused %.6f staged %.6f oracle32 %.6f oracle16 %.6f ; gn %.5f %.5f %.5f %.5f" % (
            tag, int(fused["lean"]), float(fused["loss"]), float(staged["loss"]), o32["loss"], o16["loss"],
            float(fused["gn"]), float(staged["gn"]), o32["gn"], o16["gn"]))
        print("%-62s %9s %9s | %9s %9s | %9s %9s | %9s" % ("tensor", "f-o32 max", "fro", "f-o16 max", "fro", "f-stg max", "fro", "stg-o32 f"))
        for k in sorted(o32.keys()):
            if k in ("loss", "gn"):
                continue
            e32, e16, est, es32 = errs(fused[k], o32[k]), errs(fused[k], o16[k]), errs(fused[k], staged[k]), errs(staged[k], o32[k])
            rows[k] = {"fused_vs_o32": e32, "fused_vs_o16": e16, "fused_vs_staged": est, "staged_vs_o32": es32,
                       "max_abs": float(np.abs(o32[k]).max()), "numel": int(np.asarray(o32[k]).size)}
            print("%-62s %9.2e %9.2e | %9.2e %9.2e | %9.2e %9.2e | %9.2e" % (k, e32[0], e32[1], e16[0], e16[1], est[0], est[1], es32[1]))
        # the flat-gradient max-norm figure tools/check_seq_vs_staged.py printed, and the tensor that carries it
        gk = [k for k in rows if k.startswith("g.")]
        gmax = max(float(np.abs(staged[k]).max()) for k in gk)
        worst = max(gk, key=lambda k: float(np.abs(fused[k] - staged[k]).max()))
        wv = float(np.abs(fused[worst] - staged[worst]).max()) / gmax
        print("flat-gradient max-norm deviation fused vs staged: %.3e, carried by %s (its own max |g| %.3e, global max |g| %.3e)" % (
            wv, worst, float(np.abs(staged[worst]).max()), gmax))
        report[tag] = {"lean": int(fused["lean"]), "loss": {"fused": float(fused["loss"]), "staged": float(staged["loss"]), "o32": o32["loss"], "o16": o16["loss"]},
                       "grad_norm": {"fused": float(fused["gn"]), "staged": float(staged["gn"]), "o32": o32["gn"], "o16": o16["gn"]},
                       "flat_grad_maxnorm_fused_vs_staged": {"value": wv, "tensor": worst}, "tensors": rows}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(report, open(out_path, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
