"""GPU parity of the general ("wide") stage kernels through the C ABI against numpy restatements (oracle/tape.py
primitives) on seeded inputs: dense forward / input gradient / weight gradient (ragged shapes, every activation,
dropout, residual), masked attention forward/backward (key padding, fully padded sequences, causal), row kernels.

Tolerances: exact-fp32 MFMA mode 3e-5 of the tensor magnitude; bf16-operand mode 2e-2 (operands rounded to 8 bits)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import rng, tape as tp  # noqa: E402


def dev():
    return torch.device("cuda:0")


def T_(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)


TOL = {0: 3e-5, 1: 2e-2}
ACTS = {0: lambda v: v, 1: tp.relu, 2: tp.gelu, 3: tp.elu, 4: lambda v: tp.elu(v, True)}


def seed_tensor(seed):
    return torch.tensor([seed], device=dev(), dtype=torch.int32)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("T,K,N,act,p,resid", [(200, 64, 192, 0, 0.0, False), (130, 256, 100, 2, 0.0, True), (257, 128, 64, 3, 0.3, True),
                                              (64, 64, 1024, 2, 0.0, False), (300, 1024, 256, 0, 0.2, True), (77, 64, 36, 4, 0.0, False),
                                              (129, 96, 252, 1, 0.5, False), (333, 64, 64, 0, 0.3, True), (128, 64, 64, 3, 0.0, False),
                                              (1000, 64, 64, 0, 0.5, True), (257, 64, 64, 2, 0.2, False),
                                              (300, 64, 256, 2, 0.3, False), (300, 256, 64, 0, 0.3, True), (190, 128, 128, 0, 0.0, False),
                                              (300, 256, 256, 0, 0.2, True), (1000, 256, 256, 2, 0.0, False), (33, 256, 256, 0, 0.0, False),
                                              (260, 256, 1024, 2, 0.2, False), (300, 1024, 256, 0, 0.2, True), (130, 256, 512, 0, 0.0, False),
                                              (150, 256, 768, 0, 0.0, False), (97, 512, 768, 0, 0.2, True)])      # packed in-projections: contraction 768 in the input gradient   # 256 x 256: private partials      # 64 x 64: the stage kernel of the weight gradient (k_dense_dw64)
def test_dense_forward_backward(prec, T, K, N, act, p, resid):
    from adt_amd import ops
    r = np.random.RandomState(T + K + N)
    X = r.standard_normal((T, K)).astype(np.float32)
    W = (r.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = (0.1 * r.standard_normal(N)).astype(np.float32)
    R = r.standard_normal((T, N)).astype(np.float32) if resid else None
    dY = r.standard_normal((T, N)).astype(np.float32)
    seed, site, row_off = 1234, 21, 7
    # oracle
    vx, vw, vb = tp.leaf(X), tp.leaf(W), tp.leaf(b)
    y = ACTS[act](tp.linear(vx, vw, vb))
    y = tp.dropout(y, p, seed, site, tp.idx_rows(T, N, row_off))
    if resid:
        y = tp.add(y, tp.const(R))
    tp.backward(y, dY)
    # kernel
    sd = seed_tensor(seed)
    Xg, Wg, bg = T_(X), T_(W), T_(b)
    Y, U = ops.dense_fwd(prec, Xg, Wg, bg, act, act != 0, p, sd, site, row_off, T_(R) if resid else None)
    assert rel(Y.cpu().numpy(), y.v) < TOL[prec]
    dW = torch.zeros_like(Wg)
    db = torch.zeros_like(bg)
    dX = torch.empty_like(Xg)
    ops.dense_bwd(prec, T_(dY), Xg, Wg, dW, db, dX, False, act, U, p, sd, site, row_off)
    # bf16 mode: the saved pre-activation differs from the oracle's in the last bits, so a relu / elu gate near 0 can flip
    # for single elements -- gradients are compared in relative Frobenius norm there
    gerr = rel if prec == 0 else (lambda a, b: np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    gtol = TOL[prec] if prec == 0 else 5e-2
    assert gerr(dX.cpu().numpy(), vx.g) < gtol
    assert gerr(dW.cpu().numpy(), vw.g) < gtol
    assert gerr(db.cpu().numpy(), vb.g) < gtol
    # accumulate into dX (beta) and skip the weight gradient
    dX2 = torch.ones_like(Xg)
    ops.dense_bwd(prec, T_(dY), Xg, Wg, None, None, dX2, True, act, U, p, sd, site, row_off)
    assert gerr(dX2.cpu().numpy() - 1.0, vx.g) < gtol + 1e-6


def test_dense_device_row_count():
    """t_dev caps the rows computed (masked-row batches under a captured graph): rows beyond stay untouched."""
    from adt_amd import ops
    r = np.random.RandomState(5)
    T, K, N, M = 300, 64, 132, 171
    X, W = r.standard_normal((T, K)).astype(np.float32), r.standard_normal((N, K)).astype(np.float32)
    dY = r.standard_normal((T, N)).astype(np.float32)
    Mdev = torch.tensor([M], device=dev(), dtype=torch.int32)
    Y = torch.full((T, 136), -7.0, device=dev())[:, :N]
    ops.dense_fwd(0, T_(X), T_(W), None, Y=Y, t_dev=Mdev)
    Yh = Y.cpu().numpy()
    assert rel(Yh[:M], X[:M] @ W.T) < 3e-5 and np.all(Yh[M:] == -7.0)
    dW = torch.zeros(N, K, device=dev())
    dX = torch.full((T, K), -3.0, device=dev())
    ops.dense_bwd(0, T_(dY), T_(X), T_(W), dW, None, dX, False, t_dev=Mdev)
    assert rel(dW.cpu().numpy(), dY[:M].T @ X[:M]) < 3e-5
    dXh = dX.cpu().numpy()
    assert rel(dXh[:M], dY[:M] @ W) < 3e-5 and np.all(dXh[M:] == -3.0)


def attn_oracle(q, k, v, B, H, L, key_valid, causal, fill, p, seed, site, b_off, dO):
    d = q.shape[1]
    hd = d // H
    vq, vk, vv = tp.leaf(q), tp.leaf(k), tp.leaf(v)

    def split(x):
        return tp.transpose(tp.reshape(x, (B, L, H, hd)), (0, 2, 1, 3))
    s = tp.div_const(tp.matmul(split(vq), tp.transpose(split(vk), (0, 1, 3, 2))), np.sqrt(hd))
    mask = np.broadcast_to(~key_valid[:, None, None, :], s.shape).copy()
    if causal:
        mask |= np.triu(np.ones((L, L), bool), 1)[None, None]
    s = tp.masked_fill(s, mask, fill)
    w = tp.dropout(tp.softmax(s), p, seed, site, tp.idx_attn(B, H, L, b_off))
    o = tp.reshape(tp.transpose(tp.matmul(w, split(vv)), (0, 2, 1, 3)), (B * L, d))
    tp.backward(o, dO)
    return o.v, vq.g, vk.g, vv.g


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,H,L,hd,causal,p", [(3, 2, 50, 32, False, 0.0), (2, 4, 100, 16, False, 0.3), (2, 2, 37, 64, False, 0.2),
                                              (2, 2, 200, 64, False, 0.0), (3, 2, 20, 32, True, 0.1), (2, 1, 128, 64, False, 0.0),
                                              (3, 4, 200, 64, False, 0.2), (2, 2, 160, 64, True, 0.1)])
def test_masked_attention(prec, B, H, L, hd, causal, p):
    from adt_amd import ops
    # (fp32, hd=64, L=200): the whole-(b, h) fp32 images of the backward (248 KB) exceed the LDS; it runs chunked (adt_wide.hip)
    r = np.random.RandomState(B * 1000 + L + hd)
    d = H * hd
    qkv = r.standard_normal((B * L, 3 * d)).astype(np.float32)
    ids = r.randint(1, 50, size=(B, L)).astype(np.int32)
    ids[0, : L // 3] = 0          # left padding
    if B > 2:
        ids[2, :] = 0             # a fully padded sequence: uniform attention, as in the reference (finite fill)
    dO = r.standard_normal((B * L, d)).astype(np.float32)
    seed, site, b_off = 99, 17, 5
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o, dq, dk, dv = attn_oracle(q, k, v, B, H, L, ids > 0, causal, -1e9, p, seed, site, b_off, dO)
    g = T_(qkv)
    sd = seed_tensor(seed)
    kid = T_(ids.reshape(-1))
    O, LSE = ops.attn_masked_fwd(prec, g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, H, L, causal, kid, -1e9, p, sd, site, b_off)
    assert rel(O.cpu().numpy(), o) < TOL[prec]
    dQ, dK, dV = ops.attn_masked_bwd(prec, g[:, :d], g[:, d:2 * d], g[:, 2 * d:], O, LSE, T_(dO), B, H, L, causal, kid, -1e9, p, sd, site, b_off)
    tol = TOL[prec] * (3 if prec else 1)
    assert rel(dQ.cpu().numpy(), dq) < tol
    assert rel(dK.cpu().numpy(), dk) < tol
    assert rel(dV.cpu().numpy(), dv) < tol


def test_row_kernels():
    from adt_amd import ops
    r = np.random.RandomState(3)
    T, L, d, V = 120, 20, 64, 57
    ids = r.randint(0, V, size=T).astype(np.int32)
    E, P, S = r.standard_normal((V, d)).astype(np.float32), r.standard_normal((L, d)).astype(np.float32), r.standard_normal(d).astype(np.float32)
    X = ops.embed_sum_fwd(T_(ids), T_(E), T_(P), L, T_(S), 1.0).cpu().numpy()
    assert np.array_equal(X, E[ids] + P[np.arange(T) % L] + S)
    # dropout + activation
    x = r.standard_normal((T, d)).astype(np.float32)
    dy = r.standard_normal((T, d)).astype(np.float32)
    sd = seed_tensor(77)
    for act in (0, 3, 4, 2):
        vx = tp.leaf(x)
        y = ACTS[act](tp.dropout(vx, 0.3, 77, 9, tp.idx_rows(T, d, 11)))
        tp.backward(y, dy)
        Y = ops.dropact_fwd(T_(x), 0.3, sd, 9, 11 * d, act)
        assert rel(Y.cpu().numpy(), y.v) < 1e-6
        dX = torch.ones(T, d, device=dev())
        ops.dropact_bwd(T_(dy), T_(x), 0.3, sd, 9, dX, True, 11 * d, act)
        assert rel(dX.cpu().numpy() - 1.0, vx.g) < 2e-6
    # masked-row gather / scatter with a device-side count
    rows = np.array(sorted(r.choice(T, 40, replace=False)), np.int32)
    pad = np.zeros(T, np.int32)
    pad[:40] = rows
    Mdev = torch.tensor([40], device=dev(), dtype=torch.int32)
    g = ops.gather_rows(T_(x), T_(pad), T, Mdev).cpu().numpy()
    assert np.array_equal(g[:40], x[rows])
    dF = torch.zeros(T, d, device=dev())
    ops.scatter_rows(T_(dy), T_(pad), dF, False, T, Mdev)
    want = np.zeros((T, d), np.float32)
    want[rows] = dy[:40]
    assert np.array_equal(dF.cpu().numpy(), want)
    # cross-entropy over all items, ignore_index = 0
    Vv, M = 203, 50
    z = (3 * r.standard_normal((M, Vv))).astype(np.float32)
    lab = r.randint(1, Vv, size=M).astype(np.int32)
    lab[::7] = 0
    vz = tp.leaf(z)
    ce = tp.cross_entropy(vz, lab, 0)
    tp.backward(ce)
    zl = torch.zeros(M, 204, device=dev())
    zl[:, :Vv] = T_(z)
    loss = torch.zeros(64, device=dev())
    inv = torch.tensor([1.0 / float((lab != 0).sum())], device=dev(), dtype=torch.float32)
    ops.ce_rows(zl[:, :Vv], T_(lab), Vv, inv, loss)
    assert abs(float(loss.sum()) - float(ce.v)) < 1e-5 * abs(float(ce.v))
    assert rel(zl[:, :Vv].cpu().numpy(), vz.g) < 1e-5
    # long rows: LDS-resident kernel with several sweeps per thread (30,001 items) and the streaming fallback (40,005 > 150 KB)
    for Vv in (30001, 40005):
        M = 6
        z = (2 * r.standard_normal((M, Vv))).astype(np.float32)
        lab = r.randint(1, Vv, size=M).astype(np.int32)
        lab[3] = 0
        z64 = z.astype(np.float64)
        lse = np.log(np.exp(z64 - z64.max(1, keepdims=True)).sum(1)) + z64.max(1)
        live = lab != 0
        want_loss = float((lse[live] - z64[np.arange(M)[live], lab[live]]).sum() / live.sum())
        want_g = np.exp(z64 - lse[:, None])
        want_g[np.arange(M), lab] -= 1.0
        want_g = want_g / live.sum() * live[:, None]
        ld = (Vv + 3) // 4 * 4
        zl = torch.zeros(M, ld, device=dev())
        zl[:, :Vv] = T_(z)
        loss = torch.zeros(64, device=dev())
        inv = torch.tensor([1.0 / float(live.sum())], device=dev(), dtype=torch.float32)
        ops.ce_rows(zl[:, :Vv], T_(lab), Vv, inv, loss)
        assert abs(float(loss.sum()) - want_loss) < 2e-5 * abs(want_loss)
        assert rel(zl[:, :Vv].cpu().numpy(), want_g) < 2e-5


def test_clip_adam_l2_and_score_bias():
    from adt_amd import ops
    r = np.random.RandomState(8)
    n = 5000
    P = {"w": r.standard_normal(n).astype(np.float32)}
    G = {"w": (3 * r.standard_normal(n)).astype(np.float32)}
    Pg, Gg = T_(P["w"]), T_(G["w"])
    Mg, Vg = torch.zeros_like(Pg), torch.zeros_like(Pg)
    scal = torch.zeros(192, device=dev())
    st = {}
    for _ in range(3):
        Gg.copy_(T_(G["w"]))
        ops.clip_adam_l2(Pg, Gg, Mg, Vg, 1e-2, 5.0, 1e-3, 0.9, 0.999, 1e-8, scal)
        tn = tp.clip_adam(P, {"w": G["w"].copy()}, st, 1e-3, 0.9, 0.999, 1e-8, 5.0, 1e-2)
    assert abs(float(scal[1].sqrt()) - tn) < 1e-4 * tn
    assert rel(Pg.cpu().numpy(), P["w"]) < 1e-6
    B, C, d, V = 9, 21, 64, 80
    F_, E, bias = r.standard_normal((B, d)).astype(np.float32), r.standard_normal((V, d)).astype(np.float32), r.standard_normal(V).astype(np.float32)
    cand = r.randint(1, V, size=(B, C)).astype(np.int32)
    logits, rank = ops.score_rank_bias(T_(F_), d, T_(E), T_(bias), T_(cand), B, C)
    want = np.einsum("bd,bcd->bc", F_, E[cand]) + bias[cand]
    assert rel(logits.cpu().numpy(), want) < 1e-5
    assert np.array_equal(rank.cpu().numpy(), (want[:, 1:] > want[:, :1]).sum(1))


# ---- STOSA-ADT kernels ----------------------------------------------------------------------------------------------------
def wattn_oracle(t6, ids, B, H, L, p, seed, site, b_off, dOm, dOc):
    from oracle import stosa_oracle as so
    d = t6[0].shape[1]
    hd = d // H
    vs = [tp.leaf(x) for x in t6]

    def split(x):
        return tp.transpose(tp.reshape(x, (B, L, H, hd)), (0, 2, 1, 3))
    qm, qc, km, kc, vm, vc = [split(v) for v in vs]
    s = tp.div_const(tp.neg(so.wasserstein_distance_matmul(qm, qc, km, kc)), np.sqrt(hd))
    s = tp.add_const(s, so._mask(ids))
    pr = tp.dropout(tp.softmax(s), p, seed, site, tp.idx_attn(B, H, L, b_off))
    om = tp.reshape(tp.transpose(tp.matmul(pr, vm), (0, 2, 1, 3)), (B * L, d))
    oc = tp.reshape(tp.transpose(tp.matmul(tp.square(pr), vc), (0, 2, 1, 3)), (B * L, d))
    loss = tp.add(tp.sum_(tp.mul_mask(om, dOm)), tp.sum_(tp.mul_mask(oc, dOc)))
    tp.backward(loss)
    return om.v, oc.v, [v.g for v in vs]


@pytest.mark.parametrize("prec", [None, "f32", "bf16"])
@pytest.mark.parametrize("B,H,L,hd,p", [(3, 4, 20, 16, 0.0), (2, 2, 37, 32, 0.3), (2, 1, 100, 64, 0.2), (2, 4, 100, 16, 0.3), (2, 2, 128, 32, 0.2)])
def test_wasserstein_attention(B, H, L, hd, p, prec):
    """prec None: the exact vector-ALU kernels; "f32" / "bf16": the matrix-core kernels (adt_wattn_mfma.cuh; they fall back to the former
    where they do not cover the shape -- hd = 64 here).  Tolerances: exact arithmetic 3e-5 / 1e-4 of the tensor magnitude (forward /
    gradients), bf16 operands 3e-2 / 6e-2."""
    from adt_amd import ops
    pk = None if prec is None else {"f32": ops.PREC_F32, "bf16": ops.PREC_BF16}[prec]
    tol_f, tol_g = (3e-2, 6e-2) if prec == "bf16" and hd != 64 else (3e-5, 1e-4)
    r = np.random.RandomState(B * 100 + L + hd)
    d = H * hd
    T = B * L
    qm, km, vm = (r.standard_normal((T, d)).astype(np.float32) for _ in range(3))
    qc, kc, vc = (np.exp(0.5 * r.standard_normal((T, d))).astype(np.float32) for _ in range(3))   # covariances > 0
    ids = r.randint(1, 50, size=(B, L)).astype(np.int32)
    ids[0, : L // 3] = 0          # left padding: the padded prefix rows are fully masked (uniform attention, gradient kept)
    dOm, dOc = r.standard_normal((T, d)).astype(np.float32), r.standard_normal((T, d)).astype(np.float32)
    seed, site, b_off = 4321, 16, 3
    om, oc, grads = wattn_oracle((qm, qc, km, kc, vm, vc), ids, B, H, L, p, seed, site, b_off, dOm, dOc)
    sd = seed_tensor(seed)
    g = [T_(x) for x in (qm, qc, km, kc, vm, vc)]
    kid = T_(ids.reshape(-1))
    Om, Oc, LSE = ops.wattn_fwd(*g, kid, B, H, L, p, sd, site, b_off, prec=pk)
    assert rel(Om.cpu().numpy(), om) < tol_f and rel(Oc.cpu().numpy(), oc) < tol_f
    outs = ops.wattn_bwd(*g, kid, Om, Oc, LSE, T_(dOm), T_(dOc), B, H, L, p, sd, site, b_off, prec=pk)
    for name, got, want in zip(("dQm", "dQc", "dKm", "dKc", "dVm", "dVc"), outs, grads):
        assert rel(got.cpu().numpy(), want) < tol_g, name


def test_wasserstein_bpr_and_full_sort():
    from adt_amd import ops
    from oracle import stosa_oracle as so
    r = np.random.RandomState(12)
    B, L, d, V = 5, 12, 64, 40
    T = B * L
    cfg = so.Cfg(V, L, d, 4, 1, pvn_weight=0.3)
    Em, Ec = r.standard_normal((V, d)).astype(np.float32), r.standard_normal((V, d)).astype(np.float32)
    sm = r.standard_normal((B, L, d)).astype(np.float32)
    sc = np.exp(0.3 * r.standard_normal((B, L, d))).astype(np.float32)
    pos = r.randint(1, V, size=(B, L))
    neg = r.randint(1, V, size=(B, L))
    pos[0, :4] = 0
    neg[0, :4] = 0
    Vv = {"item_mean_embeddings.weight": tp.leaf(Em), "item_cov_embeddings.weight": tp.leaf(Ec)}
    vsm, vsc = tp.leaf(sm), tp.leaf(sc)
    loss, auc, pvn = so.bpr_terms(Vv, cfg, vsm, vsc, pos, neg)
    tp.backward(tp.add(loss, pvn))
    inv = torch.tensor([1.0 / float((pos > 0).sum())], device=dev(), dtype=torch.float32)
    dEm, dEc = torch.zeros(V, d, device=dev()), torch.zeros(V, d, device=dev())
    loss3 = torch.zeros(192, device=dev())
    dSm, dSc = ops.wdist_bpr(T_(sm.reshape(T, d)), T_(sc.reshape(T, d)), T_(Em), T_(Ec), T_(pos.reshape(-1).astype(np.int32)),
                             T_(neg.reshape(-1).astype(np.int32)), 0.3, inv, dEm, dEc, loss3)
    l3 = loss3.view(3, 64).sum(1).cpu().numpy()
    assert abs(l3[0] - float(loss.v)) < 2e-5 * abs(float(loss.v)) and abs(l3[1] - float(pvn.v)) < 2e-5 * abs(float(pvn.v)) and abs(l3[2] - auc) < 1e-6
    assert rel(dSm.cpu().numpy(), vsm.g.reshape(T, d)) < 3e-5 and rel(dSc.cpu().numpy(), vsc.g.reshape(T, d)) < 3e-5
    assert rel(dEm.cpu().numpy(), Vv["item_mean_embeddings.weight"].g) < 3e-5
    assert rel(dEc.cpu().numpy(), Vv["item_cov_embeddings.weight"].g) < 3e-5
    last_m, last_c = sm[:, -1, :], sc[:, -1, :]
    want = so.wasserstein_distance_matmul(tp.const(last_m), tp.const(last_c), tp.const(Em), tp.elu(tp.const(Ec), True)).v
    got = ops.wdist_full(T_(last_m), T_(last_c), T_(Em), T_(Ec), V)
    assert rel(got.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("B,N,k", [(7, 1000, 40), (3, 12103, 40), (5, 41, 40), (2, 300, 1), (4, 70000, 40)])
def test_topk_masked_selection_is_exact(B, N, k):
    """adt_topk_masked vs the reference's numpy path (stosa/trainer.py:598-612: mask to 1e24, argpartition, argsort): integer
    output, tie-free inputs => identical ids in identical order; the distances written back are the masked inputs, bit-exact."""
    from adt_amd import ops
    r = np.random.RandomState(B * 1000 + N)
    dist = r.permutation(B * N).reshape(B, N).astype(np.float32) * 0.5 + 3.0     # distinct values, exact in fp32 (< 2^24)
    lens = r.randint(0, min(N - 1, 200), size=B)
    lens[0] = 0
    if N > 60:
        lens[-1] = N - 17                                   # fewer unmasked entries than k: the tail is filled from the masked ones
    ip = np.zeros(B + 1, np.int32)
    ip[1:] = np.cumsum(lens)
    ix = np.concatenate([r.choice(N, size=l, replace=False) for l in lens]).astype(np.int32) if ip[-1] else np.zeros(0, np.int32)
    ref = dist.copy()
    for b in range(B):
        ref[b, ix[ip[b]:ip[b + 1]]] = 1e24
    want = np.argsort(ref, axis=1, kind="stable")[:, :k]
    idx, val = ops.topk_masked(T_(dist), k, T_(ip) if len(ix) else None, T_(ix) if len(ix) else None, want_val=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert np.array_equal(idx, want)
    assert np.array_equal(val, np.take_along_axis(ref, want, 1))
    # strided rows (a column slice of a wider matrix) and no mask
    wide = T_(np.concatenate([dist, dist[:, :5]], 1))
    idx2 = ops.topk_masked(wide[:, :N], k).cpu().numpy()
    assert np.array_equal(idx2, np.argsort(dist, axis=1, kind="stable")[:, :k])
    with pytest.raises(Exception):
        ops.topk_masked(T_(dist), N + 1)


@pytest.mark.parametrize("T,K,N,act", [(1000, 256, 256, 0), (333, 256, 768, 2), (4100, 64, 64, 4), (517, 128, 1024, 1), (260, 256, 100, 3), (48, 64, 256, 2),
                                         (700, 1024, 256, 2), (300, 512, 128, 0)])
def test_dense_rows_kernels_match_tiled(T, K, N, act):
    """The row-streaming bf16 kernels (adt_dense_rows.cuh) against the tiled ones on the same inputs: same bf16 operands and fp32
    accumulation, different summation order => 2e-5 of the output magnitude (forward, incl. bias / activation / dropout /
    residuals / row mask / device row count / saved pre-activation) and 2e-5 on dX through the full prologue (mask, dropout,
    act') with the contraction chunked (N = 768, 1024) and accumulated into an existing dX (beta)."""
    from adt_amd import ops
    r = np.random.RandomState(T + K + N)
    X = T_(r.standard_normal((T, K)).astype(np.float32))
    W = T_((r.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = T_((0.1 * r.standard_normal(N)).astype(np.float32))
    R = T_(r.standard_normal((T, N)).astype(np.float32))
    R2 = T_(r.standard_normal((T, N)).astype(np.float32))
    ids = T_((r.rand(T) > 0.2).astype(np.int32))
    seed = T_(np.array([12345], np.int32))
    tdev = T_(np.array([T - 37 if T > 100 else T], np.int32))
    dY = T_(r.standard_normal((T, N)).astype(np.float32))
    dX0 = T_(r.standard_normal((T, K)).astype(np.float32))
    outs, U_shared = [], None
    for rows in (False, True):
        was = ops.dense_rows_enable(rows)
        try:
            Y, U = ops.dense_fwd(ops.PREC_BF16, X, W, b, act, act != 0, 0.3, seed, 7, 11, R, ids, None, tdev, None, R2)
            Y2, _ = ops.dense_fwd(ops.PREC_BF16, X, W, None, 0)
            if U_shared is None:
                # both backward kernels get the SAME saved pre-activation: a last-bit difference in U moves act'(U) across a
                # bf16 rounding boundary now and then, which is a 0.4 % change of one operand, not a kernel difference
                U_shared = U
            dX = dX0.clone()
            ops.dense_bwd(ops.PREC_BF16, dY, X, W, None, None, dX, True, act, U_shared, 0.3, seed, 7, 11, ids, tdev)
            dX2 = torch.empty_like(dX0)
            ops.dense_bwd(ops.PREC_BF16, dY, X, W, None, None, dX2, False)
            # weight / bias gradient: the 256 x 128-block kernel (from four output blocks on) against the tiled one, with the
            # fused prologue and the device row count, accumulated into non-zero buffers
            dW = torch.full((N, K), 0.25, device=dev())
            db = torch.full((N,), -0.5, device=dev())
            ops.dense_bwd(ops.PREC_BF16, dY, X, W, dW, db, None, False, act, U_shared, 0.3, seed, 7, 11, ids, tdev)
        finally:
            ops.dense_rows_enable(was)
        n_live = int(tdev.item())
        outs.append([t[:n_live].cpu().numpy() for t in (Y, U if U is not None else Y, dX, Y2, dX2)] + [dW.cpu().numpy(), db.cpu().numpy()])
    for a, bb, name in zip(outs[0], outs[1], ("Y", "U", "dX", "Y_plain", "dX_plain", "dW", "db")):
        assert np.isfinite(bb).all(), name
        # dW: sums over T rows in different orders (fp32 atomics).  db: the tiled kernel sums the bf16-ROUNDED tile it has in LDS
        # (error ~ 2^-9 * sum |g|), the 256 x 128-block kernel sums the fp32 values before rounding, like the reference does
        tol = {"dW": 1e-4, "db": 1e-2}.get(name, 2e-5)
        assert np.abs(a - bb).max() <= tol * max(1.0, np.abs(a).max()), (name, np.abs(a - bb).max(), np.abs(a).max())


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_dense_gradsrc_equals_fused_prologue(prec):
    """adt_dense_gradsrc + a plain adt_dense_bwd against adt_dense_bwd with the fused prologue (row mask, dropout, GELU'):
    the materialised G is the same fp32 value the fused path forms, so dX / dW / db agree to summation order."""
    from adt_amd import ops
    P = ops.PREC_F32 if prec == "f32" else ops.PREC_BF16
    r = np.random.RandomState(5)
    T, K, N = 700, 256, 512
    X = T_(r.standard_normal((T, K)).astype(np.float32))
    W = T_((r.standard_normal((N, K)) / 16).astype(np.float32))
    U = T_(r.standard_normal((T, N)).astype(np.float32))
    dY = T_(r.standard_normal((T, N)).astype(np.float32))
    ids = T_((r.rand(T) > 0.2).astype(np.int32))
    seed = T_(np.array([99], np.int32))
    tdev = T_(np.array([T - 21], np.int32))
    outs = []
    for fused in (True, False):
        dX = torch.zeros(T, K, device=dev()); dW = torch.zeros(N, K, device=dev()); db = torch.zeros(N, device=dev())
        if fused:
            ops.dense_bwd(P, dY, X, W, dW, db, dX, False, ops.ACT_GELU, U, 0.25, seed, 5, 13, ids, tdev)
        else:
            G = ops.dense_gradsrc(dY, ops.ACT_GELU, U, 0.25, seed, 5, 13, ids, tdev)
            ops.dense_bwd(P, G, X, W, dW, db, dX, False, ops.ACT_NONE, None, 0.0, None, 0, 0, None, tdev)
        outs.append([t.cpu().numpy() for t in (dX[:T - 21], dW, db)])
    for a, b, name in zip(outs[0], outs[1], ("dX", "dW", "db")):
        assert np.abs(a - b).max() <= 3e-5 * max(1.0, np.abs(a).max()), (name, np.abs(a - b).max(), np.abs(a).max())
