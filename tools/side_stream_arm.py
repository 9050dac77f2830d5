#!/usr/bin/env python3
"""One arm of tests/test_hip_model.py::test_side_stream_matches_single_stream: three fused training steps (HIP graph: eager warm-up,
capture, replay; learning rate 0, so that every step sees the same weights and only the dropout seed advances) of the bf16 flagship
kernels at L = 200, then the forward tensors and the gradients of the last -- replayed -- step to an .npz.  ADT_SIDE_STREAM (read once per process by the library) selects which backward kernels run on the side stream.

    ADT_SIDE_STREAM=0 python tools/side_stream_arm.py /tmp/a.npz ; python tools/side_stream_arm.py /tmp/b.npz
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import sasrec_oracle as so  # noqa: E402  (test infrastructure: initial weights only)
from adt_amd.sasrec import model as mm  # noqa: E402
from adt_amd.sasrec.model import SASRecADT  # noqa: E402
from adt_amd.sasrec.trainer import FusedTrainer  # noqa: E402


def main(out):
    B, L = 8, 200
    cfg = so.Cfg(3416, L, 64, 2, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    r = np.random.RandomState(4)
    a = type("Args", (), {})()
    a.device, a.num_heads, a.maxlen, a.num_layers = "cuda:0", cfg.num_heads, cfg.maxlen, cfg.num_layers
    a.hidden_units, a.dropout, a.precision = cfg.hidden_units, 0.5, "bf16"
    m = SASRecADT(1, cfg.item_num, a)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in P.items()}, strict=True)
    assert m.lib.adt_seq_layer_supported(1, L, 64, 32) == 1
    m.train()
    tr = FusedTrainer(m, [0.104292, 0.065892], [0.100833, 0.000607], lr=0.0, weight_decay=1e-3, clip=5.0, seed=5, use_graph=True)
    for step in range(3):
        seq = r.randint(1, cfg.item_num + 1, size=(B, L)).astype(np.int64)
        seq[:, : r.randint(0, L // 3)] = 0
        pos = np.where(seq > 0, r.randint(1, cfg.item_num + 1, size=(B, L)), 0).astype(np.int64)
        neg = np.where(seq > 0, r.randint(1, cfg.item_num + 1, size=(B, L)), 0).astype(np.int64)
        tr.step(seq, seq.copy(), pos, neg)
    torch.cuda.synchronize()
    T = B * L
    d = {"pos": m.ws_view(B, mm.WS_POS_LOGITS, 0, T), "neg": m.ws_view(B, mm.WS_NEG_LOGITS, 0, T), "grad": m.flat_grad, "param": m.flat}
    for i in range(3):
        d["enc_x%d" % i] = m.ws_view(B, mm.WS_ENC_X, i, T * 64)
        d["dec_x%d" % i] = m.ws_view(B, mm.WS_DEC_X, i, T * 64)
    np.savez(out, loss=np.float64(float(tr.loss())), **{k: v.detach().cpu().numpy() for k, v in d.items()})


if __name__ == "__main__":
    main(sys.argv[1])
