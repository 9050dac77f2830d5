"""GPU parity of the model-level executor (adt_sasrec_forward / loss_seed / backward / clip_adam / predict)
against (a) golden vectors recorded from the reference and (b) the numpy oracle with dropout ON (shared hash
RNG => identical masks).  fp32-MFMA precision is held to 5e-5 of each tensor's magnitude; bf16 to 3e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import sasrec_oracle as so
from tools.gen_golden_inputs import make_batch, sample_idx

pytestmark = pytest.mark.gpu

TOL = {"f32": 5e-5, "bf16": 3e-2}


class Args:
    pass


def build(cfg, P, prec, dropout=0.0):
    from adt_amd.sasrec.model import SASRecADT
    a = Args()
    a.device, a.num_heads, a.maxlen, a.num_layers = "cuda:0", cfg.num_heads, cfg.maxlen, cfg.num_layers
    a.hidden_units, a.dropout, a.precision = cfg.hidden_units, dropout, prec
    m = SASRecADT(1, cfg.item_num, a)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in P.items()}
    missing = m.load_state_dict(sd, strict=True)
    return m


def relerr(got, want):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all()
    return float(np.abs(got.astype(np.float64) - want).max()) / max(float(np.abs(want).max()), 1e-6)


def check(got, want, tol, what):
    e = relerr(got, want)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)


def check_grad(got, want, prec, what, f32_tol=2e-4):
    """fp32-MFMA: max-norm relative error.  bf16: relative Frobenius error <= 0.15 -- with bf16 operands a few
    pre-activations near zero land on the other side of the ReLU than in the fp32 oracle, which moves single
    entries of a weight gradient by a whole summand on these tiny batches; the norm-wise error stays small."""
    if prec == "f32":
        return check(got, want, f32_tol, what)
    got = got.detach().cpu().numpy().astype(np.float64)
    want = np.asarray(want, np.float64)
    assert np.isfinite(got).all()
    e = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-12)
    # (0.06-0.11 measured over dropout streams on these 6-sequence batches; the exact-fp32 mode carries the tight pin)
    assert e <= 0.15, "%s: relative Frobenius err %.3e > 1.5e-1" % (what, e)


def load_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    V, L, d, H, nl = [int(x) for x in z["cfg"]]
    return z, so.Cfg(V, L, d, H, nl, 0.0)


def test_state_dict_surface_matches_reference_names():
    cfg = so.Cfg(50, 20, 64, 2, 2, 0.0)
    P = so.init_params(cfg, 0)
    m = build(cfg, P, "f32")
    sd = m.state_dict()
    assert list(sd.keys()) != [] and set(sd.keys()) == set(P.keys())
    for k, v in P.items():
        assert tuple(sd[k].shape) == v.shape, k
        assert torch.equal(sd[k].cpu(), torch.from_numpy(v)), k
    names = [n for n, _ in m.named_parameters()]
    assert set(names) == set(P.keys())
    # xavier init as sasrec/main.py:95-99 does it must write through to the flat buffer
    for name, param in m.named_parameters():
        try:
            torch.nn.init.xavier_normal_(param.data)
        except Exception:
            pass
    off, n, shape = m._views["item_emb.weight"]
    assert torch.equal(m.flat[off:off + n].view(shape), m.item_emb.weight.data)
    assert float(m.item_emb.weight.data[0].abs().sum()) > 0   # padding row is initialised too


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_forward_loss_grads_vs_reference_golden_h4(golden_dir, prec):
    """sasrec_small_h4: d=64, H=4 (hd=16), L=20, 1 block -- tensors recorded from the reference itself."""
    z, cfg = load_golden(golden_dir, "sasrec_small_h4")
    P = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    m = build(cfg, P, prec)
    m.train()   # dropout p = 0
    tol = TOL[prec]
    out = m(None, z["seq"], z["dec"], z["pos"], z["neg"])
    check(out[0], z["pos_logits"], tol, "pos_logits")
    check(out[1], z["neg_logits"], tol, "neg_logits")
    for i in range(cfg.num_layers):
        check(out[2][i], z["enc_in.%d" % i], tol, "enc_in")
        check(out[3][i], z["dec_out.%d" % i], tol, "dec_out")
        check(out[4][i], z["rec_ind.%d" % i], tol, "rec_ind (reference row order)")
    # the reference's own loss assembly (sasrec/main.py:146-170) on our autograd-wired outputs
    import torch.nn.functional as F
    pos = z["pos"]
    lam1, lam2, wd = list(z["lam1"]), list(z["lam2"]), float(z["wd"])
    bce = torch.nn.BCEWithLogitsLoss()
    idx = np.where(pos != 0)
    loss = bce(out[0][idx], torch.ones_like(out[0])[idx]) + bce(out[1][idx], torch.zeros_like(out[1])[idx])
    for i in range(len(out[2])):
        loss = loss + lam1[i] * F.mse_loss(out[2][i], out[3][i])
    if cfg.num_heads > 1:
        B = out[4][0].shape[0]
        label = torch.tile(torch.arange(cfg.num_heads), [B * cfg.maxlen, 1]).to("cuda:0")
        for l in range(len(out[4])):
            loss = loss + lam2[i] * F.nll_loss(out[4][l].view(B * cfg.maxlen, cfg.num_heads, cfg.num_heads), label)
    for prm in m.item_emb.parameters():
        loss = loss + wd * torch.norm(prm)
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < tol * 10
    for k, prm in m.named_parameters():
        if "gnone." + k in z.files:
            assert prm.grad is None, k
        else:
            check_grad(prm.grad, z["g." + k], prec, "grad " + k)
    tn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
    assert abs(float(tn) - float(z["total_norm"])) < 4 * tol * float(z["total_norm"])


def test_fused_trainer_three_steps_vs_reference_golden_h4(golden_dir):
    from adt_amd.sasrec.trainer import FusedTrainer
    z, cfg = load_golden(golden_dir, "sasrec_small_h4")
    P = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    m = build(cfg, P, "f32")
    m.train()
    tr = FusedTrainer(m, list(z["lam1"]), list(z["lam2"]), lr=1e-3, weight_decay=float(z["wd"]), clip=5.0)
    for step in range(3):
        tr.step(z["seq"], z["dec"], z["pos"], z["neg"])
        if step in (0, 2):
            assert abs(float(tr.loss()) - float(z["loss_step%d" % (step + 1)])) < 1e-4
            sd = m.state_dict()
            for k in P:
                ref = z["w%d.%s" % (step + 1, k)]
                g = z["g." + k] if "g." + k in z.files else np.zeros_like(ref)
                noisy = np.abs(g) < 1e-6   # Adam turns rounding noise on exactly-zero gradients into +-lr (see test_oracle_golden)
                got = sd[k].cpu().numpy()
                err = np.abs(np.where(noisy, 0, got - ref)).max()
                assert err < 3e-5, (k, step, err)
                assert np.abs(got - ref).max() <= (step + 1) * 1e-3 * 1.01 + 1e-6, k
        if step == 0:
            assert abs(float(tr.grad_norm()) - float(z["total_norm"])) < 1e-4 * float(z["total_norm"])


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_cfga_slice_vs_reference_golden(golden_dir, prec):
    """cfg-A shape (L=200, d=64, H=2, 2 blocks, V=3416), B=8, against samples recorded from the reference."""
    from adt_amd.sasrec.trainer import FusedTrainer
    z, cfg = load_golden(golden_dir, "sasrec_cfga_b8")
    seed, B = int(z["seed"]), int(z["B"])
    P = so.init_params(cfg, seed=seed)
    batch = make_batch(np.random.RandomState(seed + 1), B, cfg.maxlen, cfg.item_num)
    m = build(cfg, P, prec)
    m.train()
    tol = TOL[prec]
    tr = FusedTrainer(m, list(z["lam1"]), list(z["lam2"]), lr=1e-3, weight_decay=float(z["wd"]), clip=5.0)
    w0 = m.flat.clone()
    tr.step(*batch)
    T = B * cfg.maxlen
    from adt_amd.sasrec import model as mm
    check(m.ws_view(B, mm.WS_POS_LOGITS, 0, T).view(B, -1), z["pos_logits"], tol, "pos_logits")
    check(m.ws_view(B, mm.WS_NEG_LOGITS, 0, T).view(B, -1), z["neg_logits"], tol, "neg_logits")
    assert abs(float(tr.loss()) - float(z["loss"])) < (1e-4 if prec == "f32" else 2e-2)
    assert abs(float(tr.grad_norm()) - float(z["total_norm"])) < (1e-3 if prec == "f32" else 5e-2) * float(z["total_norm"])
    for k, _ in so.param_shapes(cfg):
        if "gnone." + k in z.files:
            continue
        gn = float(m.grad_view(k).norm())
        # flat_grad holds the clipped? no: the raw reduced gradient incl. the weight-decay term
        want = float(z["gnorm." + k])
        assert abs(gn - want) <= (2e-3 if prec == "f32" else 6e-2) * want + 1e-7, (k, gn, want)
        got = m.grad_view(k).reshape(-1).cpu().numpy()[sample_idx(m.grad_view(k).numel())]
        scale = max(float(np.abs(z["gsample." + k]).max()), want / np.sqrt(max(m.grad_view(k).numel(), 1)), 1e-9)
        # sampled entries: fp32 sums over 1600 tokens re-associate differently (wave-local tiles vs torch), so the
        # bound is relative to the tensor's scale, 1e-3 in fp32-MFMA mode
        assert np.abs(got - z["gsample." + k]).max() <= max(4 * tol, 1e-3) * scale + 1e-8, k


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_training_step_with_dropout_vs_oracle(prec):
    """Dropout ON (p = 0.5): HIP and oracle share the hash RNG, so the whole step is comparable.  Also checks
    the data-parallel contract: a shard with b_offset and global normalisers reproduces its slice."""
    from adt_amd.sasrec.trainer import FusedTrainer
    from adt_amd.sasrec import model as mm
    cfg = so.Cfg(300, 50, 64, 2, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    B = 6
    batch = make_batch(np.random.RandomState(4), B, cfg.maxlen, cfg.item_num)
    lam1, lam2, wd = [0.104292, 0.065892], [0.100833, 0.000607], 1e-3
    m = build(cfg, P, prec, dropout=0.5)
    m.train()
    tr = FusedTrainer(m, lam1, lam2, lr=1e-3, weight_decay=wd, clip=5.0, seed=5)
    tr.step(*batch)
    seed = int(m._seed.cpu().numpy().view(np.uint32)[0])
    Po = {k: v.copy() for k, v in P.items()}
    out = so.forward(Po, cfg, *batch, training=True, seed=seed)
    loss, parts, seeds = so.loss_and_seeds(Po, cfg, out, batch[2], lam1, lam2, wd)
    G = so.backward(Po, cfg, out[5], seeds, wd)
    tol = TOL[prec]
    T = B * cfg.maxlen
    check(m.ws_view(B, mm.WS_POS_LOGITS, 0, T).view(B, -1), out[0], tol, "pos_logits (dropout)")
    for i in range(cfg.num_layers):
        check(m.ws_view(B, mm.WS_ENC_X, i, T * 64).view(B, -1, 64), out[2][i], tol, "enc_in (dropout)")
        check(m.ws_view(B, mm.WS_DEC_X, cfg.num_layers - i, T * 64).view(B, -1, 64), out[3][i], tol, "dec_out (dropout)")
    assert abs(float(tr.loss()) - loss) < (2e-4 if prec == "f32" else 3e-2)
    num = den = 0.0
    for k, _ in so.param_shapes(cfg):
        if G[k] is None:
            assert float(m.grad_view(k).abs().max()) == 0.0, k
        elif prec == "f32":
            check_grad(m.grad_view(k), G[k], prec, "grad (dropout) " + k, f32_tol=3e-4)
        else:
            # bf16 operands against the fp32 oracle under a p = 0.5 mask on 6 sequences: single tensors move by 0.06 - 0.17 in
            # relative Frobenius norm from one dropout stream to the next (ReLU / mask-edge flips move whole summands), so the
            # bound is per tensor 0.25 and over the whole gradient 0.1; the exact-fp32 mode above carries the tight pin
            got, want = m.grad_view(k).cpu().numpy().astype(np.float64), np.asarray(G[k], np.float64)
            assert np.isfinite(got).all(), k
            e2, w2 = float(((got - want) ** 2).sum()), float((want ** 2).sum())
            assert e2 <= (0.25 ** 2) * max(w2, 1e-24), "grad (dropout) %s: relative Frobenius err %.3e" % (k, (e2 / max(w2, 1e-24)) ** 0.5)
            num, den = num + e2, den + w2
    if prec != "f32":
        assert num <= (0.1 ** 2) * den, "whole gradient: relative Frobenius err %.3e" % ((num / den) ** 0.5)
    tn = so.grad_norm(G)
    assert abs(float(tr.grad_norm()) - tn) < (1e-3 if prec == "f32" else 5e-2) * tn
    # shard [2:5) of the same global batch with b_offset = 2 and global normalisers
    m2 = build(cfg, P, prec, dropout=0.5)
    m2.train()
    tr2 = FusedTrainer(m2, lam1, lam2, lr=1e-3, weight_decay=wd, clip=5.0, seed=5)
    norms = (float((batch[2] != 0).sum()), float(B * cfg.maxlen * 64), float(B * cfg.maxlen * 2))
    tr2.step(*[a[2:5] for a in batch], norms=norms, b_offset=2)
    Ts = 3 * cfg.maxlen
    got = m2.ws_view(3, mm.WS_ENC_X, 1, Ts * 64).view(3, -1, 64)
    check(got, out[2][1][2:5], tol, "shard enc_in[1] equals the global batch's slice")


LEAN_SHAPES = [(2, 200, 8, 3416), (4, 48, 3, 300), (1, 100, 4, 300), (2, 52, 5, 300)]      # (H, L, B, V); the first is cfg-A's shape


def _lean_bounds(name):
    """Relative Frobenius bound of one parameter gradient of the bf16 kernels against the fp32 oracle under a p = 0.5 mask.
    Measured over the four shapes (tools/lean_parity_report.py, profiles/r03_lean_parity.json): matrices <= 0.118 (a 64 x 64
    feed-forward weight on the 5-sequence batch; the oracle run with bf16-rounded operands is itself 0.03 - 0.06 away from the fp32
    oracle on the same tensors: a pre-activation that lands on the other side of the ReLU / of a dropped element moves a whole
    summand), 1-D tensors (biases, LayerNorm weights) <= 0.063.  A wrong factor, a missing term or a missing tensor is an error of
    0.5 - 1."""
    return 0.10 if name.endswith("bias") or "norm.weight" in name else 0.15


@pytest.mark.parametrize("H,L,B,V", LEAN_SHAPES)
def test_lean_step_with_dropout_vs_oracle(H, L, B, V):
    """The kernels bench.py times -- the per-sequence fused layer kernels on transposed tiles with bf16 saved tensors, keep-bits and
    private weight-gradient partials (k_seqtt_*, k_seq_attn_bwd, k_dwpart_reduce; L % 4 == 0) -- with dropout ON (p = 0.5, the sites of
    sasrec/modules.py:60-61,629-633 and sasrec/model.py:38) against the oracle on the shared hash RNG: both logit tensors, every
    encoder input / decoder output, the head-classifier log-probabilities, the loss, the gradient norm and EVERY parameter gradient.
    The assert on adt_seq_layer_supported makes a silent fall-back to the staged kernels a failure."""
    from adt_amd.sasrec.trainer import FusedTrainer
    from adt_amd.sasrec import model as mm
    cfg = so.Cfg(V, L, 64, H, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), B, L, V)
    lam1, lam2, wd = [0.104292, 0.065892], [0.100833, 0.000607], 1e-3
    m = build(cfg, P, "bf16", dropout=0.5)
    assert m.lib.adt_seq_layer_supported(1, L, 64, 64 // H) == 1, "the lean per-sequence kernels do not cover this shape"
    assert os.environ.get("ADT_SEQ", "1") != "0"
    m.train()
    tr = FusedTrainer(m, lam1, lam2, lr=1e-3, weight_decay=wd, clip=5.0, seed=5)
    tr.step(*batch)
    seed = int(m._seed.cpu().numpy().view(np.uint32)[0])
    out = so.forward(P, cfg, *batch, training=True, seed=seed)
    loss, parts, seeds = so.loss_and_seeds(P, cfg, out, batch[2], lam1, lam2, wd)
    G = so.backward(P, cfg, out[5], seeds, wd)
    T = B * L
    tol = 2e-2          # measured <= 1.2e-2 (profiles/r03_lean_parity.json)
    check(m.ws_view(B, mm.WS_POS_LOGITS, 0, T).view(B, L), out[0], tol, "pos_logits")
    check(m.ws_view(B, mm.WS_NEG_LOGITS, 0, T).view(B, L), out[1], tol, "neg_logits")
    for i in range(cfg.num_layers):
        check(m.ws_view(B, mm.WS_ENC_X, i, T * 64).view(B, L, 64), out[2][i], tol, "enc_in.%d" % i)
        check(m.ws_view(B, mm.WS_DEC_X, cfg.num_layers - i, T * 64).view(B, L, 64), out[3][i], tol, "dec_out.%d" % i)
        if H > 1:
            check(m.ws_view(B, mm.WS_REC, i, T * H * H).view(B, L, H, H), so.rec_reference_order(out[4][i]), tol, "rec_ind.%d" % i)
    assert abs(float(tr.loss()) - loss) < 2e-3 * abs(loss), (float(tr.loss()), loss)          # measured <= 2e-4
    tn = so.grad_norm(G)
    assert abs(float(tr.grad_norm()) - tn) < 1e-2 * tn, (float(tr.grad_norm()), tn)          # measured <= 1.2e-3
    num = den = 0.0
    for k, _ in so.param_shapes(cfg):
        got = m.grad_view(k).cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), k
        if G[k] is None:
            assert float(np.abs(got).max()) == 0.0, k
            continue
        want = np.asarray(G[k], np.float64).reshape(got.shape)
        e2, w2 = float(((got - want) ** 2).sum()), float((want ** 2).sum())
        assert w2 > 0.0, k
        assert e2 <= _lean_bounds(k) ** 2 * w2, "grad %s: relative Frobenius err %.3e > %.2f" % (k, (e2 / w2) ** 0.5, _lean_bounds(k))
        num, den = num + e2, den + w2
    assert num <= 0.06 ** 2 * den, "whole gradient: relative Frobenius err %.3e" % ((num / den) ** 0.5)          # measured <= 0.035


def test_lean_step_vs_bf16_operand_oracle_cfga_shape():
    """The tight bound at the benchmarked shape (H 2, L 200, B 8, V 3416, p = 0.5): the HIP bf16 step against the oracle run with the SAME
    operand rounding (`so.operands("bf16")`: every matrix product rounds both operands to bfloat16 and accumulates in fp32, everything else
    fp32, as in the kernels).  With the rounding shared, what is left is summation order and the bf16 SAVED tensors of the lean path:
    every parameter gradient within 0.05 in relative Frobenius norm, the whole gradient within 0.03 (measured: worst tensor 0.036 --
    pos_emb.weight, which with item_emb.weight carries most of the gradient's norm and receives the input gradient of BOTH stacks, i.e. every
    bf16-saved tensor's rounding -- whole gradient 0.0255).  A wrong 1 / (1 - p) on one small tensor is an error of 0.5 - 1 there; the 0.15 of
    test_lean_step_with_dropout_vs_oracle (fp32 oracle) is kept for the 3-5-sequence shapes only."""
    from adt_amd.sasrec.trainer import FusedTrainer
    from adt_amd.sasrec import model as mm
    H, L, B, V = 2, 200, 8, 3416
    cfg = so.Cfg(V, L, 64, H, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), B, L, V)
    lam1, lam2, wd = [0.104292, 0.065892], [0.100833, 0.000607], 1e-3
    m = build(cfg, P, "bf16", dropout=0.5)
    assert m.lib.adt_seq_layer_supported(1, L, 64, 64 // H) == 1
    m.train()
    tr = FusedTrainer(m, lam1, lam2, lr=1e-3, weight_decay=wd, clip=5.0, seed=5)
    tr.step(*batch)
    seed = int(m._seed.cpu().numpy().view(np.uint32)[0])
    with so.operands("bf16"):
        out = so.forward(P, cfg, *batch, training=True, seed=seed)
        loss, parts, seeds = so.loss_and_seeds(P, cfg, out, batch[2], lam1, lam2, wd)
        G = so.backward(P, cfg, out[5], seeds, wd)
    T = B * L
    check(m.ws_view(B, mm.WS_POS_LOGITS, 0, T).view(B, L), out[0], 1e-2, "pos_logits")
    check(m.ws_view(B, mm.WS_NEG_LOGITS, 0, T).view(B, L), out[1], 1e-2, "neg_logits")
    assert abs(float(tr.loss()) - loss) < 1e-3 * abs(loss), (float(tr.loss()), loss)
    tn = so.grad_norm(G)
    assert abs(float(tr.grad_norm()) - tn) < 5e-3 * tn, (float(tr.grad_norm()), tn)
    num = den = 0.0
    worst = ("", 0.0)
    for k, _ in so.param_shapes(cfg):
        got = m.grad_view(k).cpu().numpy().astype(np.float64)
        if G[k] is None:
            assert float(np.abs(got).max()) == 0.0, k
            continue
        want = np.asarray(G[k], np.float64).reshape(got.shape)
        e2, w2 = float(((got - want) ** 2).sum()), float((want ** 2).sum())
        assert w2 > 0.0, k
        if (e2 / w2) ** 0.5 > worst[1]:
            worst = (k, (e2 / w2) ** 0.5)
        assert e2 <= 0.05 ** 2 * w2, "grad %s: relative Frobenius err %.3e > 0.05 against the bf16-operand oracle" % (k, (e2 / w2) ** 0.5)
        num, den = num + e2, den + w2
    assert num <= 0.03 ** 2 * den, "whole gradient: relative Frobenius err %.3e (worst tensor %s %.3e)" % ((num / den) ** 0.5, worst[0], worst[1])


def test_fused_kernels_vs_staged_kernels_cfga_shape(tmp_path):
    """tools/check_seq_vs_staged.py as a test: the per-sequence fused kernels (default) against the staged stage kernels (ADT_SEQ=0; the
    switch is read once per process, so each arm is its own process) on the same weights, batch and dropout seed, bf16, p = 0.5, at
    cfg-A's shape (H 2, L 200, V 3416, B 8).  Outputs agree to 1e-2 of their magnitude, every parameter gradient to 0.15 in relative
    Frobenius norm, the whole gradient to 0.05 (measured 4.6e-3 / 0.045 / 0.02).  The MAX-NORM deviation of the flat gradient, which
    that tool printed as 9e-2 (and 1.8e-1 at this shape), is carried by single rows of item_emb.weight: items that occur once or twice in
    the batch, whose whole gradient row is one token's dX -- with bf16 saved tensors a different ReLU / dropout-edge decision in one
    arm replaces that row; the fp32 oracle against the SAME oracle with bf16-rounded operands shows the same 0.20 max-norm / 0.035
    Frobenius on that tensor (profiles/r03_lean_parity.json), so it is rounding, not a defect, and it is bounded here in Frobenius norm."""
    import subprocess
    import sys
    arms = []
    for flag in ("1", "0"):
        out = str(tmp_path / ("arm%s.npz" % flag))
        subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lean_parity_report.py"),
                               "--arm", "2", "200", "8", "3416", out], env=dict(os.environ, ADT_SEQ=flag))
        arms.append(np.load(out))
    fused, staged = arms
    assert int(fused["lean"]) == 1 and int(fused["seed"]) == int(staged["seed"])
    assert abs(float(fused["loss"]) - float(staged["loss"])) < 1e-3 * float(staged["loss"])
    num = den = 0.0
    for k in fused.files:
        if k in ("lean", "seed", "loss", "gn"):
            continue
        a, b = np.asarray(fused[k], np.float64), np.asarray(staged[k], np.float64)
        if not k.startswith("g."):
            assert np.abs(a - b).max() <= 1e-2 * max(np.abs(b).max(), 1e-6), k
            continue
        e2, w2 = float(((a - b) ** 2).sum()), float((b ** 2).sum())
        if w2 == 0.0:
            assert e2 == 0.0, k
            continue
        assert e2 <= 0.15 ** 2 * w2, (k, (e2 / w2) ** 0.5)
        num, den = num + e2, den + w2
    assert num <= 0.05 ** 2 * den, (num / den) ** 0.5


@pytest.mark.parametrize("L", [52, 200])
def test_lean_step_is_deterministic_and_ignores_workspace_garbage(L):
    """Two launches of the same bf16 step (fresh model each, the workspace filled with NaN beforehand) give the SAME forward tensors bit for
    bit and the same gradients up to the order of fp32 atomic sums.  This is the test that caught an inline-asm instruction reading MFMA
    accumulators before the matrix pipe had written them (the compiler's hazard recognizer does not look inside asm statements): every
    oracle test passed at bf16 tolerance while the forward differed by 2e-3 from run to run."""
    from adt_amd.sasrec.trainer import FusedTrainer
    from adt_amd.sasrec import model as mm
    B = 6
    cfg = so.Cfg(300, L, 64, 2, 2, dropout=0.5)
    P = so.init_params(cfg, seed=3)
    batch = make_batch(np.random.RandomState(4), B, L, cfg.item_num)
    runs = []
    for rep in range(3):
        m = build(cfg, P, "bf16", dropout=0.5)
        assert m.lib.adt_seq_layer_supported(1, L, 64, 32) == 1
        m.train()
        m.workspace(B).fill_(float("nan"))
        tr = FusedTrainer(m, [0.104292, 0.065892], [0.100833, 0.000607], lr=1e-3, weight_decay=1e-3, clip=5.0, seed=5)
        tr.step(*batch)
        torch.cuda.synchronize()
        T = B * L
        fwd = [m.ws_view(B, mm.WS_POS_LOGITS, 0, T).clone(), m.ws_view(B, mm.WS_NEG_LOGITS, 0, T).clone()]
        fwd += [m.ws_view(B, mm.WS_ENC_X, i, T * 64).clone() for i in range(3)] + [m.ws_view(B, mm.WS_DEC_X, i, T * 64).clone() for i in range(3)]
        runs.append((fwd, m.flat_grad.clone(), float(tr.loss())))
    for fwd, grad, loss in runs[1:]:
        for a, b in zip(runs[0][0], fwd):
            assert torch.isfinite(a).all() and torch.equal(a, b)
        assert abs(loss - runs[0][2]) <= 1e-6 * abs(loss)
        assert torch.isfinite(grad).all()
        assert float((grad - runs[0][1]).abs().max()) <= 2e-6 * float(runs[0][1].abs().max())
        # every parameter gradient EXCEPT the two embedding tables is one ordered sum (weight-gradient and vector partials per workgroup,
        # summed in workgroup order; LDS sums in wave order): bit-equal.  The tables are float-atomic scatters unless ADT_ITEM_SORT=1
        # (test_step_is_bit_deterministic_with_sorted_item_gradient).
        n_tab = m.offsets[2]      # item table + positional table
        assert torch.equal(grad[n_tab:], runs[0][1][n_tab:]), "a non-table gradient differs between two runs: %g" % float((grad[n_tab:] - runs[0][1][n_tab:]).abs().max())


def test_step_is_bit_deterministic_with_sorted_item_gradient(tmp_path):
    """ADT_ITEM_SORT=1: the item / positional table gradients are sorted segmented sums (adt_itemgrad.cuh) instead of float-atomic scatters, so
    with the ordered partial sums of everything else the WHOLE step is a pure function of its inputs: two processes-worth of fresh models give
    bit-equal gradients, gradient norm and updated weights after three steps (the reference's CPU step is run-to-run deterministic too:
    sasrec/model.py:34-41,72-76 via autograd's index_add).  Own process: the switch is read once per process."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import sasrec_oracle as so
from tests.test_hip_model import build, make_batch
from adt_amd.sasrec.trainer import FusedTrainer
L, B = 52, 6
cfg = so.Cfg(300, L, 64, 2, 2, dropout=0.5)
P = so.init_params(cfg, seed=3)
batches = [make_batch(np.random.RandomState(4 + i), B, L, cfg.item_num) for i in range(3)]
outs = []
for rep in range(3):
    m = build(cfg, P, "bf16", dropout=0.5)
    m.train()
    m.workspace(B).fill_(float("nan"))
    tr = FusedTrainer(m, [0.104292, 0.065892], [0.100833, 0.000607], lr=1e-3, weight_decay=1e-3, clip=5.0, seed=5, use_graph=(rep == 2))
    g = []
    for b in batches:
        tr.step(*b)
        torch.cuda.synchronize()
        g.append(m.flat_grad.clone())
    outs.append((g, m.flat.clone(), float(tr.grad_norm())))
for g, w, gn in outs[1:]:
    for a, b in zip(outs[0][0], g):
        assert torch.isfinite(a).all() and torch.equal(a, b), float((a - b).abs().max())
    assert torch.equal(outs[0][1], w), float((outs[0][1] - w).abs().max())
    assert gn == outs[0][2]
print("bit-equal")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ADT_ITEM_SORT="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and "bit-equal" in p.stdout, p.stderr[-3000:]


def test_lean_parity_with_one_workgroup_per_sequence():
    """Below 129 sequences per GPU the per-sequence kernels map several workgroups to a sequence (adt_sasrec.hip: seq_split / attn_split), so
    the small batches of this suite exercise the SPLIT mapping; the benchmarked batch of 256 runs one workgroup per sequence.  This runs the
    oracle-parity, bf16-operand-parity, determinism and cfg-A golden cases again in a child process with ADT_SEQ_SPLIT=1 (the switch is read
    once per process): the mapping bench.py times."""
    import subprocess
    import sys
    here = os.path.abspath(__file__)
    cases = ["test_lean_step_with_dropout_vs_oracle", "test_lean_step_vs_bf16_operand_oracle_cfga_shape",
             "test_lean_step_is_deterministic_and_ignores_workspace_garbage", "test_cfga_slice_vs_reference_golden", "test_graph_replay_matches_eager"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + ["%s::%s" % (here, c) for c in cases],
                       cwd=os.path.dirname(os.path.dirname(here)), env=dict(os.environ, ADT_SEQ_SPLIT="1"), capture_output=True, text=True, timeout=850)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-2000:]
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.parametrize("switch", ["ADT_EMBED3=0", "ADT_BCE_MERGED=0", "ADT_BCE_SIDE=0", "ADT_LNL_FUSED=0", "ADT_ATTN_SPLIT=0", "ADT_FWD_FUSED=0",
                                    "ADT_FOLD_PARTS=0", "ADT_SIDE_STREAM=0"])
def test_ab_switches_keep_parity(switch):
    """The process-wide A/B switches of the flagship step select older forms of a stage -- separate scatters, the logits
    kernel on the side stream / in the backward, the last LayerNorm's own kernel, one workgroup per sequence in the attention-block backward,
    the unfused loss assembly, replica folds, no side stream.  Each must still match the oracle and the reference golden: one child process per
    switch (they are read once per process) runs the dropout parity case, the fused-trainer golden and the determinism case."""
    import subprocess
    import sys
    here = os.path.abspath(__file__)
    cases = ["test_lean_step_with_dropout_vs_oracle", "test_fused_trainer_three_steps_vs_reference_golden_h4",
             "test_lean_step_is_deterministic_and_ignores_workspace_garbage"]
    k, v = switch.split("=")
    if k == "ADT_FOLD_PARTS":      # float atomics for the bias / LayerNorm sums again: parity holds, bit-equality across runs does not
        cases = cases[:2]
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + ["%s::%s" % (here, c) for c in cases],
                       cwd=os.path.dirname(os.path.dirname(here)), env=dict(os.environ, **{k: v}), capture_output=True, text=True, timeout=850)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-2000:]
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, tail


def test_side_stream_matches_single_stream(tmp_path):
    """The backward puts its scatter / fold kernels on a side stream under the chain kernels (adt_sasrec.hip: side_stream; in the captured
    step they are parallel branches of the HIP graph).  Two processes run the same three steps at the flagship shape's L = 200 (eager
    warm-up, capture, replay; learning rate 0: with bf16 operand rounding a 1e-9 difference in a weight can flip a rounding and move a
    gradient by 1e-4, so the weights are held), one with everything on one stream (ADT_SIDE_STREAM=0): the forward tensors of the replayed
    step are bit-equal, its gradients agree to the order of the fp32 atomic sums.  A missing dependency edge would show here as a
    gradient read before its producer finished."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, val in (("single", "0"), ("side", "7")):
        out = str(tmp_path / (name + ".npz"))
        env = dict(os.environ, ADT_SIDE_STREAM=val)
        subprocess.run([sys.executable, os.path.join(repo, "tools", "side_stream_arm.py"), out], check=True, env=env, timeout=300)
        outs.append(np.load(out))
    a, b = outs
    for k in a.files:
        assert np.isfinite(b[k]).all(), k
        if k in ("grad", "loss"):
            assert np.abs(a[k] - b[k]).max() <= 4e-6 * max(np.abs(a[k]).max(), 1e-30), (k, np.abs(a[k] - b[k]).max())
        else:
            assert np.array_equal(a[k], b[k]), (k, np.abs(a[k] - b[k]).max())


def test_predict_and_rank_vs_reference_golden(golden_dir):
    z, cfg = load_golden(golden_dir, "sasrec_small_h4")
    P = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    m = build(cfg, P, "f32")
    m.eval()
    check(m.predict(None, z["seq"], z["cand"]), z["predict_cand"], 5e-5, "predict cand")
    check(m.predict(None, z["seq"], None, full=True), z["predict_full"], 5e-5, "predict full")
    logits, rank = m.predict_rank(z["seq"], z["cand"])
    assert (rank.cpu().numpy() == so.rank_of_first(logits.cpu().numpy())).all()


def test_graph_replay_matches_eager():
    from adt_amd.sasrec.trainer import FusedTrainer
    cfg = so.Cfg(200, 50, 64, 2, 2, dropout=0.5)
    P = so.init_params(cfg, seed=8)
    r = np.random.RandomState(9)
    batches = [make_batch(r, 4, cfg.maxlen, cfg.item_num) for _ in range(4)]
    res = []
    for use_graph in (False, True):
        m = build(cfg, P, "f32", dropout=0.5)
        m.train()
        tr = FusedTrainer(m, [0.1, 0.05], [0.1, 0.01], weight_decay=1e-3, use_graph=use_graph, seed=1)
        losses = []
        for b in batches:
            tr.step(*b)
            losses.append(float(tr.loss()))
        res.append((losses, m.flat.clone()))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-6), (res[0][0], res[1][0])
    # atomics reorder fp32 sums between runs, and Adam turns rounding noise on exactly-zero gradients (key
    # bias) into +-lr per step: weights agree to 4 steps * lr, the bulk to rounding
    diff = (res[0][1] - res[1][1]).abs()
    assert float(diff.max()) <= 4 * 1e-3 * 1.01
    assert float((diff > 1e-5).float().mean()) < 1e-2
