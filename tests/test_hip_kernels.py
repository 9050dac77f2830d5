"""GPU parity tests of every C-ABI stage kernel against the numpy oracle on the same seeded inputs.
Tolerances (stated per precision): PREC_F32 (exact-fp32 MFMA) 2e-5 of the tensor's max magnitude;
PREC_BF16 (bf16 operands, fp32 accumulate) 2e-2 of it."""
import math

import numpy as np
import pytest
import torch

from oracle import rng
from oracle import sasrec_oracle as so

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 2e-2}


def dev():
    return torch.device("cuda:0")


def T_(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


def seed_t(seed):
    return torch.from_numpy(np.array([seed], dtype=np.uint32).view(np.int32)).to(dev())


def relerr(got, want):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all(), "non-finite values in HIP output"
    scale = max(float(np.abs(want).max()), 1e-6)
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) / scale


def check(got, want, tol, what):
    e = relerr(got, want)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)


@pytest.fixture(scope="module")
def ops():
    from adt_amd import ops as o
    torch.cuda.init()
    return o


def ids_with_padding(r, B, L, V):
    ids = r.randint(1, V + 1, size=(B, L)).astype(np.int32)
    for b in range(1, B):
        ids[b, : r.randint(0, L)] = 0
    return ids


@pytest.mark.parametrize("p", [0.0, 0.5])
def test_embed_fwd_bwd(ops, p):
    r = np.random.RandomState(0)
    B, L, d, V = 5, 37, 64, 90
    E = r.randn(V + 1, d).astype(np.float32)
    P = r.randn(L, d).astype(np.float32)
    ids = ids_with_padding(r, B, L, V)
    seed, site, boff = 99, so.SITE_EMB_SEQ, 3
    want, (keep, m) = so.embed(ids, E, P, p, seed, site, boff)
    got = ops.embed_fwd(T_(ids.reshape(-1)), T_(E), T_(P), L, p, seed_t(seed), site, boff * L)
    check(got.view(B, L, d), want, 1e-6, "embed_fwd")
    g = r.randn(B, L, d).astype(np.float32)
    gm = g * m
    if keep is not None:
        gm = gm * keep / (1 - p)
    dE = np.zeros_like(E)
    np.add.at(dE, ids, gm * np.float32(math.sqrt(d)))
    dP = gm.sum(0)
    dE_t, dP_t = torch.zeros(V + 1, d, device=dev()), torch.zeros(L, d, device=dev())
    ops.embed_bwd(T_(ids.reshape(-1)), T_(g.reshape(-1, d)), L, p, seed_t(seed), site, dE_t, dP_t, boff * L)
    check(dE_t, dE, 1e-5, "embed dE")
    check(dP_t, dP, 1e-5, "embed dP")


@pytest.mark.parametrize("layout", ["sampler", "random", "mixed"])
@pytest.mark.parametrize("p", [0.0, 0.25])      # (rates are quantised to 1 / 256: 0.25 is exact)
def test_embed_bwd3_one_pass_item_scatter(ops, layout, p):
    """adt_embed_bwd3: encoder + decoder embedding gradients + positive-logit rows into the item table in one pass.  With the reference sampler's
    layout (seq = items[:-1], pos = items[1:], dec[:, 1:] = seq[:, :-1]: sasrec/utils.py:288-307) the three rows of a token share ONE atomic
    row-add; with random ids every contribution is an 'orphan' added on its own; 'mixed' breaks the shifts in a third of the places.  All three
    must equal the three separate scatters (sasrec/model.py:34-41, :53-59, :72-76 reversed), positional sums included."""
    r = np.random.RandomState(7)
    B, L, d, V = 6, 44, 64, 50
    if layout == "random":
        seq, dec, pos = (ids_with_padding(r, B, L, V) for _ in range(3))
    else:
        seq = np.zeros((B, L), np.int64); dec = np.zeros((B, L), np.int64); pos = np.zeros((B, L), np.int64)
        for b in range(B):
            n = r.randint(3, L + 1)
            items = r.randint(1, V + 1, size=n + 1)
            seq[b, L - n:] = items[:-1]
            pos[b, L - n:] = items[1:]
            dec[b, 1:] = seq[b, :-1]
        if layout == "mixed":
            for a in (seq, dec, pos):
                m = r.rand(B, L) < 0.33
                a[m] = r.randint(0, V + 1, size=int(m.sum()))
    seed, boff = 77, 2
    gs, gd, F = (r.randn(B, L, d).astype(np.float32) for _ in range(3))
    gp = (r.randn(B, L) * (pos != 0)).astype(np.float32)
    dE = np.zeros((V + 1, d), np.float32)
    dP = np.zeros((L, d), np.float32)
    for ids, g, site in ((seq, gs, so.SITE_EMB_SEQ), (dec, gd, so.SITE_EMB_DEC)):
        _, (keep, m) = so.embed(ids, np.zeros((V + 1, d), np.float32), np.zeros((L, d), np.float32), p, seed, site, boff)
        gm = g * m
        if keep is not None:
            gm = gm * keep / (1 - p)
        np.add.at(dE, ids, gm * np.float32(math.sqrt(d)))
        dP += gm.sum(0)
    np.add.at(dE, pos, F * gp[..., None] * (pos != 0)[..., None])
    dE[0] = 0.0
    dE_t, dP_t = torch.zeros(V + 1, d, device=dev()), torch.zeros(L, d, device=dev())
    ops.embed_bwd3(T_(seq.reshape(-1), torch.int32), T_(dec.reshape(-1), torch.int32), T_(pos.reshape(-1), torch.int32), T_(gs.reshape(-1, d)), T_(gd.reshape(-1, d)), T_(F.reshape(-1, d)),
                   T_(gp.reshape(-1)), L, p, seed_t(seed), so.SITE_EMB_SEQ, so.SITE_EMB_DEC, dE_t, dP_t, 1, 0, boff * L)
    check(dE_t, dE, 2e-5, "embed_bwd3 dE")
    check(dP_t, dP, 2e-5, "embed_bwd3 dP")


@pytest.mark.parametrize("d", [64, 256])
def test_layernorm_fwd_bwd(ops, d):
    r = np.random.RandomState(1)
    T = 333
    x = (r.randn(T, d) * 2 + 0.3).astype(np.float32)
    x[5] = 0.0   # zero row: var = 0, eps = 1e-8 decides
    w = (1 + 0.1 * r.randn(d)).astype(np.float32)
    b = (0.1 * r.randn(d)).astype(np.float32)
    want, cache = so.layer_norm(x, w, b)
    got = ops.layernorm_fwd(T_(x), T_(w), T_(b), 1e-8)
    check(got, want, 2e-6, "ln fwd")
    dy = r.randn(T, d).astype(np.float32)
    dy[5] = 0.0
    dx, dw, db = so.layer_norm_bwd(dy, cache, w)
    base = r.randn(T, d).astype(np.float32)
    for acc in (0, 1):
        dX = T_(base.copy())
        dg, dbt = torch.zeros(d, device=dev()), torch.zeros(d, device=dev())
        ops.layernorm_bwd(T_(dy), T_(x), T_(w), 1e-8, dX, acc, dg, dbt)
        check(dX, dx + (base if acc else 0), 1e-5, "ln dx acc=%d" % acc)
        check(dg, dw, 1e-5, "ln dgamma")
        check(dbt, db, 1e-5, "ln dbeta")


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("N", [64, 128, 192, 16])
def test_linear_fwd_plain(ops, prec, N):
    r = np.random.RandomState(2)
    T, K = 200, 64   # T not a multiple of 64: ragged last tile
    x = r.randn(T, K).astype(np.float32)
    w = (r.randn(N, K) / 8).astype(np.float32)
    b = r.randn(N).astype(np.float32)
    got = ops.linear_fwd(prec, T_(x), T_(w), T_(b))
    check(got, x @ w.T + b, TOL[prec], "linear_fwd N=%d" % N)


@pytest.mark.parametrize("prec", [0, 1])
def test_linear_fwd_strided_views_and_epilogue(ops, prec):
    """The fused FFN/out_proj epilogue: mask(R1 + R2 + relu(dropout(xW^T + b))) on strided in/out views."""
    r = np.random.RandomState(3)
    B, L, K, N = 3, 50, 64, 64
    T = B * L
    xbig = r.randn(T, 3 * K).astype(np.float32)
    w = (r.randn(N, K) / 8).astype(np.float32)
    b = r.randn(N).astype(np.float32)
    r1 = r.randn(T, N).astype(np.float32)
    r2 = r.randn(T, N).astype(np.float32)
    ids = ids_with_padding(r, B, L, 30).reshape(-1)
    seed, site, boff, p = 1234, 18, 7, 0.5
    x = xbig[:, K:2 * K]
    t = x @ w.T + b
    keep = rng.keep_mask(seed, site, so._row_idx(B, L, N, boff).reshape(T, N), p)
    want = (r1 + r2 + np.maximum(t * keep / (1 - p), 0)) * (ids != 0)[:, None]
    xb = T_(xbig)
    ybig = torch.zeros(T, 2 * N, device=dev())
    ops.linear_fwd(prec, xb[:, K:2 * K], T_(w), T_(b), Y=ybig[:, N:], p=p, seed=seed_t(seed), site=site, row_offset=boff * L,
                   relu=True, R1=T_(r1), R2=T_(r2), mask_ids=T_(ids))
    check(ybig[:, N:], want, TOL[prec], "linear_fwd epilogue")
    assert float(ybig[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("N", [64, 128, 192])
def test_linear_bwd(ops, prec, N):
    r = np.random.RandomState(4)
    B, L, K = 7, 61, 64   # T = 427: several tiles + ragged tail
    T = B * L
    x = r.randn(T, K).astype(np.float32)
    w = (r.randn(N, K) / 8).astype(np.float32)
    dy = r.randn(T, N).astype(np.float32)
    ids = ids_with_padding(r, B, L, 30).reshape(-1)
    u = r.randn(T, N).astype(np.float32)
    radd = r.randn(T, K).astype(np.float32)
    seed, site, boff, p = 77, 131, 2, 0.5
    keep = rng.keep_mask(seed, site, so._row_idx(B, L, N, boff).reshape(T, N), p)
    dyp = dy * (ids != 0)[:, None] * keep / (1 - p) * (u > 0)
    base = r.randn(T, K).astype(np.float32)
    want_dx = base + dyp @ w + radd * (ids != 0)[:, None]
    dW, db = torch.zeros(N, K, device=dev()), torch.zeros(N, device=dev())
    dX = T_(base.copy())
    ops.linear_bwd(prec, T_(dy), T_(x), T_(w), dW, db, dX=dX, beta=True, mask_ids=T_(ids), p=p, seed=seed_t(seed), site=site,
                   row_offset=boff * L, U=T_(u), Radd=T_(radd), radd_ids=T_(ids))
    check(dX, want_dx, TOL[prec], "linear_bwd dX")
    check(dW, dyp.T @ x, TOL[prec], "linear_bwd dW")
    check(db, dyp.sum(0), 1e-5, "linear_bwd db")
    # plain form: no prologue, beta = 0, strided dY view (the q slice of a packed qkv gradient)
    dybig = r.randn(T, 3 * 64).astype(np.float32)
    if N == 64:
        dW.zero_(); db.zero_()
        dX2 = torch.full((T, K), float("nan"), device=dev())
        ops.linear_bwd(prec, T_(dybig)[:, 64:128], T_(x), T_(w), dW, db, dX=dX2)
        check(dX2, dybig[:, 64:128] @ w, TOL[prec], "linear_bwd plain dX")
        check(dW, dybig[:, 64:128].T @ x, TOL[prec], "linear_bwd plain dW")


def _attn_case(r, B, H, L, hd):
    d = H * hd
    q = r.randn(B, L, d).astype(np.float32)
    k = r.randn(B, L, d).astype(np.float32)
    v = r.randn(B, L, d).astype(np.float32)
    return q, k, v


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("H,hd,L", [(2, 32, 200), (2, 32, 50), (4, 16, 100), (1, 64, 100), (2, 32, 33), (4, 16, 20)])
@pytest.mark.parametrize("causal,p", [(True, 0.0), (True, 0.5), (False, 0.2)])
def test_attention_fwd_bwd(ops, prec, H, hd, L, causal, p):
    r = np.random.RandomState(5)
    B = 3
    d = H * hd
    q, k, v = _attn_case(r, B, H, L, hd)
    seed, site, boff = 4242, 16, 5
    want_o, cache = so.attention(q, k, v, H, causal, p, seed, site, boff)
    # packed qkv buffer: exercises leading dimensions
    qkv = np.concatenate([q, k, v], -1).reshape(B * L, 3 * d)
    qkv_t = T_(qkv)
    O, LSE = ops.attn_fwd(prec, qkv_t[:, :d], qkv_t[:, d:2 * d], qkv_t[:, 2 * d:], B, H, L, causal, p, seed_t(seed), site, boff)
    check(O.view(B, L, d), want_o, TOL[prec], "attn fwd O")
    check(LSE.view(B, H, L), cache[7], TOL[prec], "attn LSE")
    do = r.randn(B, L, d).astype(np.float32)
    wq, wk, wv = so.attention_bwd(do, cache, H)
    dQ, dK, dV = ops.attn_bwd(prec, qkv_t[:, :d], qkv_t[:, d:2 * d], qkv_t[:, 2 * d:], O, LSE, T_(do.reshape(B * L, d)), B, H, L, causal,
                              p, seed_t(seed), site, boff)
    check(dQ.view(B, L, d), wq, TOL[prec] * 2, "attn dQ")
    check(dK.view(B, L, d), wk, TOL[prec] * 2, "attn dK")
    check(dV.view(B, L, d), wv, TOL[prec] * 2, "attn dV")
    if p > 0:
        # dropout keep-bit buffer: the forward records its decisions, the backward reads them instead of re-hashing
        mask = torch.zeros(B * H * L * 8, device=dev(), dtype=torch.int32)
        O2, LSE2 = ops.attn_fwd(prec, qkv_t[:, :d], qkv_t[:, d:2 * d], qkv_t[:, 2 * d:], B, H, L, causal, p, seed_t(seed), site, boff, mask)
        assert torch.equal(O2, O)
        dQ2, dK2, dV2 = ops.attn_bwd(prec, qkv_t[:, :d], qkv_t[:, d:2 * d], qkv_t[:, 2 * d:], O2, LSE2, T_(do.reshape(B * L, d)), B, H, L,
                                     causal, p, seed_t(seed), site, boff, mask)
        check(dQ2.view(B, L, d), wq, TOL[prec] * 2, "attn dQ (mask bits)")
        check(dK2.view(B, L, d), wk, TOL[prec] * 2, "attn dK (mask bits)")
        check(dV2.view(B, L, d), wv, TOL[prec] * 2, "attn dV (mask bits)")


def test_attention_softmax_extremes(ops):
    """Large score spread: one key dominates each row; LSE must stay finite (fp32 path)."""
    r = np.random.RandomState(6)
    B, H, L, hd = 2, 2, 64, 32
    q, k, v = _attn_case(r, B, H, L, hd)
    q *= 30.0
    want_o, cache = so.attention(q, k, v, H, True, 0.0, 0, 0)
    O, LSE = ops.attn_fwd(0, T_(q.reshape(-1, H * hd)), T_(k.reshape(-1, H * hd)), T_(v.reshape(-1, H * hd)), B, H, L, True)
    check(O.view(B, L, H * hd), want_o, 1e-4, "attn extremes O")
    check(LSE.view(B, H, L), cache[7], 1e-5, "attn extremes LSE")


@pytest.mark.parametrize("H,hd", [(2, 32), (4, 16), (1, 64)])
def test_headcls_fwd_bwd(ops, H, hd):
    r = np.random.RandomState(7)
    B, L = 4, 23
    d = H * hd
    o = r.randn(B, L, d).astype(np.float32)
    Ws = (r.randn(H, hd) / 4).astype(np.float32)
    bs = r.randn(H).astype(np.float32)
    z = o.reshape(B, L, H, hd) @ Ws.T + bs
    zm = z.max(-1, keepdims=True)
    rec = z - zm - np.log(np.exp(z - zm).sum(-1, keepdims=True))
    got = ops.headcls_fwd(T_(o.reshape(-1, d)), T_(Ws), T_(bs), B, L)
    check(got.view(B, L, H, H), so.rec_reference_order(rec), 1e-5, "headcls fwd")
    drec_tok = r.randn(B, L, H, H).astype(np.float32)
    sm = np.exp(rec)
    dz = drec_tok - sm * drec_tok.sum(-1, keepdims=True)
    want_dWs = dz.reshape(-1, H).T @ o.reshape(-1, hd) if False else np.einsum("blhc,blhj->cj", dz, o.reshape(B, L, H, hd))
    want_dbs = dz.reshape(-1, H).sum(0)
    base = r.randn(B * L, d).astype(np.float32)
    want_do = base + (dz @ Ws).reshape(B * L, d)
    dO = T_(base.copy())
    dWs, dbs = torch.zeros(H, hd, device=dev()), torch.zeros(H, device=dev())
    ops.headcls_bwd(T_(o.reshape(-1, d)), T_(Ws), got, T_(so.rec_reference_order(drec_tok)), B, L, dO, dWs, dbs)
    check(dO, want_do, 1e-5, "headcls dO")
    check(dWs, want_dWs, 1e-5, "headcls dWs")
    check(dbs, want_dbs, 1e-5, "headcls dbs")


def test_logits_and_loss_seeds(ops):
    r = np.random.RandomState(8)
    B, L, d, V, H = 6, 41, 64, 120, 2
    T = B * L
    f = r.randn(T, d).astype(np.float32)
    E = (r.randn(V + 1, d) / 4).astype(np.float32)
    pos = ids_with_padding(r, B, L, V).reshape(-1)
    neg = np.where(pos != 0, r.randint(1, V + 1, size=T), 0).astype(np.int32)
    pl, nl = ops.logits_fwd(T_(f), T_(E), T_(pos), T_(neg))
    wp, wn = (f * E[pos]).sum(-1), (f * E[neg]).sum(-1)
    check(pl, wp, 1e-5, "pos logits")
    check(nl, wn, 1e-5, "neg logits")
    norms = np.array([float((pos != 0).sum()), float(T * d), float(T * H)], np.float32)
    loss = torch.zeros(4 * 64, device=dev())   # 64 sub-slots per loss term
    dpos, dneg = ops.bce_seed(pl, nl, T_(pos), T_(norms), loss)
    mk = pos != 0
    sig = lambda t: 1 / (1 + np.exp(-t))
    check(dpos, (sig(wp) - 1) * mk / norms[0], 1e-5, "dpos")
    check(dneg, sig(wn) * mk / norms[0], 1e-5, "dneg")
    want_loss = [(so.softplus(-wp) * mk).sum() / norms[0], (so.softplus(wn) * mk).sum() / norms[0]]
    check(loss[:128].view(2, 64).sum(1), np.array(want_loss, np.float32), 1e-5, "bce loss")
    dE = torch.zeros(V + 1, d, device=dev())
    dF = ops.logits_bwd(T_(f), T_(E), T_(pos), T_(neg), dpos, dneg, dE)
    gp, gn = dpos.cpu().numpy(), dneg.cpu().numpy()
    check(dF, gp[:, None] * E[pos] + gn[:, None] * E[neg], 1e-5, "dF")
    wdE = np.zeros_like(E)
    np.add.at(wdE, pos, gp[:, None] * f)
    np.add.at(wdE, neg, gn[:, None] * f)
    wdE[0] = 0
    check(dE, wdE, 1e-5, "logits dE")
    # mse / nll seeds
    a, b = r.randn(T, d).astype(np.float32), r.randn(T, d).astype(np.float32)
    GA, GB = T_(np.ones((T, d), np.float32)), torch.empty(T, d, device=dev())
    ops.mse_seed(T_(a), T_(b), 0.3, T_(norms), GA, 1, GB, loss[128:192])
    g = 2 * 0.3 / norms[1] * (a - b)
    check(GA, 1 + g, 1e-5, "mse GA")
    check(GB, -g, 1e-5, "mse GB")
    check(loss[128:192].sum().view(1), np.array([((a - b) ** 2).sum() / norms[1]], np.float32), 1e-5, "mse loss")
    rec = np.log(np.random.RandomState(9).dirichlet(np.ones(H), size=(T, H))).astype(np.float32)
    drec = torch.empty(T, H, H, device=dev())
    ops.nll_seed(T_(rec), H, 0.7, T_(norms), drec, loss[192:256])
    check(drec, np.broadcast_to(-0.7 / norms[2] * np.eye(H, dtype=np.float32), (T, H, H)), 1e-6, "nll drec")
    check(loss[192:256].sum().view(1), np.array([-(rec * np.eye(H)).sum() / norms[2]], np.float32), 1e-5, "nll loss")


def test_clip_adam_three_steps(ops):
    r = np.random.RandomState(10)
    n, nE = 5000, 1280
    P0 = r.randn(n).astype(np.float32)
    P = {"item_emb.weight": P0[:nE].reshape(20, 64).copy(), "rest": P0[nE:].copy()}
    state = {}
    Pt = T_(P0.copy())
    M, V = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    scal = torch.zeros(192, device=dev())
    wd = 1e-2
    for step in range(3):
        g = (r.randn(n) * (3.0 if step == 1 else 0.02)).astype(np.float32)   # step 1 clips, others do not
        G = {"item_emb.weight": g[:nE].reshape(20, 64).copy(), "rest": g[nE:].copy()}
        nrm = np.sqrt((P["item_emb.weight"].astype(np.float64) ** 2).sum())
        G["item_emb.weight"] = G["item_emb.weight"] + (wd / nrm * P["item_emb.weight"]).astype(np.float32)
        tn, coef = so.clip_adam(P, G, state, lr=1e-3, clip=5.0)
        ops.clip_adam(Pt, T_(g), M, V, nE, wd, 5.0, 1e-3, 0.9, 0.98, 1e-8, scal)
        want = np.concatenate([P["item_emb.weight"].reshape(-1), P["rest"]])
        check(Pt, want, 2e-6, "adam step %d" % step)
        s = scal.cpu().numpy()
        assert abs(math.sqrt(s[1]) - tn) < 1e-4 * tn
        assert s[2] == step + 1


def test_score_rank(ops):
    r = np.random.RandomState(11)
    B, C, d, V = 9, 101, 64, 300
    E = r.randn(V + 1, d).astype(np.float32)
    f = r.randn(B, 3 * d).astype(np.float32)   # ld = 3d: strided last-position rows
    cand = r.randint(1, V + 1, size=(B, C)).astype(np.int32)
    ft = T_(f)
    logits, rank = ops.score_rank(ft[:, d:2 * d], 3 * d, T_(E), T_(cand), B, C)
    want = np.einsum("bcd,bd->bc", E[cand], f[:, d:2 * d])
    check(logits, want, 1e-5, "cand logits")
    got_l = logits.cpu().numpy()
    assert (rank.cpu().numpy() == so.rank_of_first(got_l)).all()
    full, _ = ops.score_rank(ft[:, d:2 * d], 3 * d, T_(E), None, B, V + 1, want_rank=False)
    check(full, f[:, d:2 * d] @ E.T, 1e-5, "full logits")
