"""Torch-tensor wrappers over the C ABI (include/adt_hip.h).  Tensors must be CUDA (HIP) fp32/int32 and
contiguous in their last dimension; every call enqueues on torch's current stream.  Nothing here computes on
the host: a missing libadt_hip.so or a CPU tensor is an error, not a fallback.
"""
import ctypes

import torch

from . import _lib

PREC_F32, PREC_BF16 = 0, 1


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.AdtError("adt_amd.ops: tensor is not on the GPU (no CPU fallback)")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ld(t):
    return t.stride(-2) if t is not None and t.dim() >= 2 else 0


def _f32(t):
    assert t.dtype == torch.float32 and t.stride(-1) == 1, (t.dtype, t.stride())
    return t


def _i32(t):
    assert t.dtype == torch.int32 and t.is_contiguous()
    return t


def embed_fwd(ids, E, P, L, p, seed, site, row_offset=0):
    T = ids.numel()
    d = E.shape[1]
    X = torch.empty(T, d, device=E.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_embed_fwd(_p(_i32(ids)), _p(_f32(E)), _p(_f32(P)), T, L, d, float(p), _p(seed), site,
                                         row_offset, _p(X), _stream()), "embed_fwd")
    return X


def embed_bwd(ids, dX, L, p, seed, site, dE, dP, row_offset=0):
    T, d = dX.shape
    _lib.check(_lib.load().adt_embed_bwd(_p(_i32(ids)), _p(_f32(dX)), T, L, d, float(p), _p(seed), site, row_offset,
                                         _p(dE), _p(dP), _stream()), "embed_bwd")


def embed_bwd3(seq, dec, pos, dXs, dXd, F, dpos, L, p, seed, site_seq, site_dec, dE_rep, dP, nrep=1, rep_stride=0, row_offset=0):
    """Encoder + decoder embedding gradients and the positive-logit rows into the item-table replicas in one pass (adt_embed_bwd3, d = 64)."""
    T = dXs.shape[0]
    _lib.check(_lib.load().adt_embed_bwd3(_p(_i32(seq)), _p(_i32(dec)), _p(_i32(pos)), _p(_f32(dXs)), _p(_f32(dXd)), _p(_f32(F)), _p(_f32(dpos)), T, L,
                                          float(p), _p(seed), site_seq, site_dec, row_offset, _p(dP), _p(dE_rep), nrep, rep_stride, _stream()), "embed_bwd3")


def layernorm_fwd(X, gamma, beta, eps):
    T, d = X.shape
    Y = torch.empty(T, d, device=X.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_layernorm_fwd(_p(_f32(X)), _ld(X), _p(gamma), _p(beta), eps, T, d, _p(Y), d, _stream()), "ln_fwd")
    return Y


def layernorm_bwd(dY, X, gamma, eps, dX, accumulate, dgamma, dbeta):
    T, d = X.shape
    _lib.check(_lib.load().adt_layernorm_bwd(_p(_f32(dY)), _ld(dY), _p(_f32(X)), _ld(X), _p(gamma), eps, T, d, _p(dX), _ld(dX),
                                             int(accumulate), _p(dgamma), _p(dbeta), _stream()), "ln_bwd")


def linear_fwd(prec, X, W, b, Y=None, p=0.0, seed=None, site=0, row_offset=0, relu=False, R1=None, R2=None, mask_ids=None):
    T, K = X.shape
    N = W.shape[0]
    if Y is None:
        Y = torch.empty(T, N, device=X.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_linear_fwd(prec, _p(_f32(X)), _ld(X), _p(_f32(W)), _p(b), T, K, N, _p(Y), _ld(Y), float(p), _p(seed),
                                          site, row_offset, int(relu), _p(R1), _ld(R1), _p(R2), _ld(R2), _p(mask_ids), _stream()),
               "linear_fwd")
    return Y


def linear_bwd(prec, dY, X, W, dW, db, dX=None, beta=False, mask_ids=None, p=0.0, seed=None, site=0, row_offset=0, U=None,
               Radd=None, radd_ids=None):
    T, K = X.shape
    N = W.shape[0]
    _lib.check(_lib.load().adt_linear_bwd(prec, _p(_f32(dY)), _ld(dY), _p(_f32(X)), _ld(X), _p(_f32(W)), T, K, N, _p(mask_ids),
                                          float(p), _p(seed), site, row_offset, _p(U), _ld(U), _p(dX), _ld(dX), int(beta), _p(Radd),
                                          _ld(Radd), _p(radd_ids), _p(dW), _p(db), _stream()), "linear_bwd")


def attn_fwd(prec, Q, K, V, B, H, L, causal=True, p=0.0, seed=None, site=0, b_offset=0, mask=None):
    """Q, K, V: (B*L, >= H*hd) views (head h in columns [h*hd, (h+1)*hd)); returns O (B*L, H*hd), LSE (B*H*L)."""
    d = Q.shape[1]
    hd = d // H
    O = torch.empty(B * L, d, device=Q.device, dtype=torch.float32)
    LSE = torch.empty(B * H * L, device=Q.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_attn_fwd(prec, _p(_f32(Q)), _ld(Q), _p(_f32(K)), _ld(K), _p(_f32(V)), _ld(V), B, H, L, hd, int(causal),
                                        float(p), _p(seed), site, b_offset, _p(O), d, _p(LSE), _p(mask), _stream()), "attn_fwd")
    return O, LSE


def attn_bwd(prec, Q, K, V, O, LSE, dO, B, H, L, causal=True, p=0.0, seed=None, site=0, b_offset=0, mask=None, out=None):
    d = Q.shape[1]
    hd = d // H
    if out is None:
        dQ = torch.empty(B * L, d, device=Q.device, dtype=torch.float32)
        dK = torch.empty_like(dQ)
        dV = torch.empty_like(dQ)
    else:
        dQ, dK, dV = out
    _lib.check(_lib.load().adt_attn_bwd(prec, _p(_f32(Q)), _ld(Q), _p(_f32(K)), _ld(K), _p(_f32(V)), _ld(V), _p(_f32(O)), _ld(O),
                                        _p(LSE), _p(_f32(dO)), _ld(dO), B, H, L, hd, int(causal), float(p), _p(seed), site, b_offset,
                                        _p(dQ), _ld(dQ), _p(dK), _ld(dK), _p(dV), _ld(dV), _p(mask), _stream()), "attn_bwd")
    return dQ, dK, dV


def headcls_fwd(O, Ws, bs, B, L):
    H, hd = Ws.shape
    rec = torch.empty(L * B, H, H, device=O.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_headcls_fwd(_p(_f32(O)), _ld(O), _p(Ws), _p(bs), B, L, H, hd, _p(rec), _stream()), "headcls_fwd")
    return rec


def headcls_bwd(O, Ws, rec, drec, B, L, dO, dWs, dbs):
    H, hd = Ws.shape
    _lib.check(_lib.load().adt_headcls_bwd(_p(_f32(O)), _ld(O), _p(Ws), _p(rec), _p(drec), B, L, H, hd, _p(dO), _ld(dO), _p(dWs),
                                           _p(dbs), _stream()), "headcls_bwd")


def logits_fwd(F, E, pos, neg):
    T, d = F.shape
    pl = torch.empty(T, device=F.device, dtype=torch.float32)
    nl = torch.empty_like(pl)
    _lib.check(_lib.load().adt_logits_fwd(_p(_f32(F)), _ld(F), _p(E), _p(_i32(pos)), _p(_i32(neg)), T, d, _p(pl), _p(nl), _stream()),
               "logits_fwd")
    return pl, nl


def logits_bwd(F, E, pos, neg, dpos, dneg, dE):
    T, d = F.shape
    dF = torch.empty(T, d, device=F.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_logits_bwd(_p(_f32(F)), _ld(F), _p(E), _p(_i32(pos)), _p(_i32(neg)), _p(dpos), _p(dneg), T, d, _p(dF),
                                          d, _p(dE), _stream()), "logits_bwd")
    return dF


def item_scatter(ids, G, rowscale, scale, p, seed, site, row_offset, rep, nrep, rep_stride):
    """rep[wave % nrep][ids[row]] += rowscale[row] * scale * dropmask * G[row]  (rows with id 0 skipped); see adt_item_scatter."""
    T, d = G.shape
    _lib.check(_lib.load().adt_item_scatter(_p(_i32(ids)), _p(_f32(G)), _ld(G), _p(rowscale), T, d, float(scale), float(p), _p(seed), site, row_offset,
                                            _p(rep), nrep, rep_stride, _stream()), "item_scatter")


def replica_reduce(dE, rep, nrep, rep_stride):
    _lib.check(_lib.load().adt_replica_reduce(_p(dE), _p(rep), dE.numel(), nrep, rep_stride, _stream()), "replica_reduce")


def posemb_bwd(ids, dX, L, p, seed, site, row_offset, dP):
    T, d = dX.shape
    _lib.check(_lib.load().adt_posemb_bwd(_p(_i32(ids)), _p(_f32(dX)), T, L, d, float(p), _p(seed), site, row_offset, _p(dP), _stream()), "posemb_bwd")


def logits_bwd_df(E, pos, neg, dpos, dneg):
    T, d = pos.numel(), E.shape[1]
    dF = torch.empty(T, d, device=E.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_logits_bwd_df(_p(E), _p(_i32(pos)), _p(_i32(neg)), _p(dpos), _p(dneg), T, d, _p(dF), d, _stream()), "logits_bwd_df")
    return dF


def bce_seed(pos_logits, neg_logits, pos, norms, loss2):
    T = pos_logits.numel()
    dpos = torch.empty(T, device=pos_logits.device, dtype=torch.float32)
    dneg = torch.empty_like(dpos)
    _lib.check(_lib.load().adt_bce_seed(_p(pos_logits), _p(neg_logits), _p(_i32(pos)), T, _p(norms), _p(dpos), _p(dneg), _p(loss2),
                                        _stream()), "bce_seed")
    return dpos, dneg


def mse_seed(A, Bm, lam, norms, GA, accumulate_a, GB, loss1):
    _lib.check(_lib.load().adt_mse_seed(_p(A), _p(Bm), A.numel(), float(lam), _p(norms), _p(GA), int(accumulate_a), _p(GB), _p(loss1),
                                        _stream()), "mse_seed")


def nll_seed(rec, H, lam2, norms, drec, loss1):
    _lib.check(_lib.load().adt_nll_seed(_p(rec), rec.numel() // (H * H), H, float(lam2), _p(norms), _p(drec), _p(loss1), _stream()),
               "nll_seed")


def clip_adam(P, G, M, V, nE, wd, clip, lr, b1, b2, eps, scal, grad_scale=1.0, n=None):
    n = P.numel() if n is None else n
    _lib.check(_lib.load().adt_clip_adam(_p(P), _p(G), _p(M), _p(V), n, nE, float(wd), float(clip), float(lr), float(b1), float(b2),
                                         float(eps), float(grad_scale), _p(scal), _stream()), "clip_adam")


def clip_adam_pre(P, G, M, V, nE, wd, clip, lr, b1, b2, eps, scal, grad_scale=1.0, n=None):
    """clip_adam for a step opened by SASRecADT.run_step_begin (||E||^2 partials and zeroed slots already in scal)."""
    n = P.numel() if n is None else n
    _lib.check(_lib.load().adt_clip_adam_pre(_p(P), _p(G), _p(M), _p(V), n, nE, float(wd), float(clip), float(lr), float(b1), float(b2),
                                             float(eps), float(grad_scale), _p(scal), _stream()), "clip_adam_pre")


def score_rank(F, ldf, E, cand, B, C, want_rank=True):
    d = E.shape[1]
    logits = torch.empty(B, C, device=E.device, dtype=torch.float32)
    rank = torch.empty(B, device=E.device, dtype=torch.int32) if want_rank else None
    _lib.check(_lib.load().adt_score_rank(_p(F), ldf, _p(E), _p(cand), B, C, d, _p(logits), _p(rank), _stream()), "score_rank")
    return logits, rank


# ---- general ("wide") stage kernels: BERT4Rec-ADT, STOSA-ADT (include/adt_hip.h) ------------------------------------
ACT_NONE, ACT_RELU, ACT_GELU, ACT_ELU, ACT_ELU1 = 0, 1, 2, 3, 4


def dense_fwd(prec, X, W, b=None, act=ACT_NONE, save_u=False, p=0.0, seed=None, site=0, row_offset=0, R=None, mask_ids=None, Y=None,
              t_dev=None, ldy=None, R2=None):
    """Y = mask(R + dropout(act(X W^T + b))); returns (Y, U) with U the saved pre-activation (or None)."""
    T, K = X.shape
    N = W.shape[0]
    if Y is None:
        ldy = N if ldy is None else ldy
        Y = torch.empty(T, ldy, device=X.device, dtype=torch.float32)[:, :N]
    U = torch.empty(T, N, device=X.device, dtype=torch.float32) if save_u else None
    _lib.check(_lib.load().adt_dense_fwd(prec, _p(_f32(X)), _ld(X), _p(_f32(W)), _ld(W), _p(b), T, K, N, act, _p(U), _ld(U), float(p), _p(seed),
                                         site, row_offset, _p(R), _ld(R), _p(R2), _ld(R2), _p(mask_ids), _p(Y), _ld(Y), _p(t_dev), _stream()), "dense_fwd")
    return Y, U


_DENSE_WS = {}


def _ensure_dense_ws(device):
    """Scratch of the 256-wide stage kernels (adt_dense_workspace: private partials of the weight gradients): one 64 MiB buffer per device, kept for the process.  The library holds ONE pointer, so it is re-registered
    when the calls move to another device (one process per GPU is the normal case and registers once, before any graph capture: the
    trainers warm up eagerly)."""
    if _DENSE_WS.get("current") == device:
        return
    ws = _DENSE_WS.get(device)
    if ws is None:
        ws = _DENSE_WS[device] = torch.empty(64 << 20, device=device, dtype=torch.uint8)
    _lib.check(_lib.load().adt_dense_workspace(_p(ws), ws.numel()), "dense_workspace")
    _DENSE_WS["current"] = device


def dense_bwd(prec, dY, X, W, dW=None, db=None, dX=None, beta=False, act=ACT_NONE, U=None, p=0.0, seed=None, site=0, row_offset=0,
              mask_ids=None, t_dev=None):
    """G = dY * mask * dropmask * act'(U); dX (+)= G W; dW += G^T X; db += colsum(G)."""
    T, N = dY.shape
    K = W.shape[1]
    if dW is not None and prec == PREC_BF16 and ((N % 256 == 0 and K % 256 == 0) or (N % 64 == 0 and K % 64 == 0 and (N // 64) * (K // 64) <= 4)):
        _ensure_dense_ws(dY.device)      # 256-wide layers and 64 x 64 blocks: private partials + an ordered sum instead of an atomic flush
    _lib.check(_lib.load().adt_dense_bwd(prec, _p(_f32(dY)), _ld(dY), T, K, N, _p(mask_ids), float(p), _p(seed), site, row_offset, act, _p(U),
                                         _ld(U), _p(X), _ld(X), _p(_f32(W)), _ld(W), _p(dX), _ld(dX), int(beta), _p(dW), _ld(dW) if dW is not None else 0,
                                         _p(db), _p(t_dev), _stream()), "dense_bwd")


def attn_masked_fwd(prec, Q, K, V, B, H, L, causal=False, key_ids=None, fill=-1e9, p=0.0, seed=None, site=0, b_offset=0):
    d = Q.shape[1]
    hd = d // H
    O = torch.empty(B * L, d, device=Q.device, dtype=torch.float32)
    LSE = torch.empty(B * H * L, device=Q.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_attn_masked_fwd(prec, _p(_f32(Q)), _ld(Q), _p(_f32(K)), _ld(K), _p(_f32(V)), _ld(V), B, H, L, hd, int(causal),
                                               _p(key_ids), float(fill), float(p), _p(seed), site, b_offset, _p(O), d, _p(LSE), _stream()),
               "attn_masked_fwd")
    return O, LSE


def attn_masked_bwd(prec, Q, K, V, O, LSE, dO, B, H, L, causal=False, key_ids=None, fill=-1e9, p=0.0, seed=None, site=0, b_offset=0, out=None):
    d = Q.shape[1]
    hd = d // H
    if out is None:
        dQ = torch.empty(B * L, d, device=Q.device, dtype=torch.float32)
        dK = torch.empty_like(dQ)
        dV = torch.empty_like(dQ)
    else:
        dQ, dK, dV = out
    _lib.check(_lib.load().adt_attn_masked_bwd(prec, _p(_f32(Q)), _ld(Q), _p(_f32(K)), _ld(K), _p(_f32(V)), _ld(V), _p(_f32(O)), _ld(O), _p(LSE),
                                               _p(_f32(dO)), _ld(dO), B, H, L, hd, int(causal), _p(key_ids), float(fill), float(p), _p(seed), site,
                                               b_offset, _p(dQ), _ld(dQ), _p(dK), _ld(dK), _p(dV), _ld(dV), _stream()), "attn_masked_bwd")
    return dQ, dK, dV


def embed_sum_fwd(ids, E, P, L, S0=None, scale=1.0):
    T = ids.numel()
    d = E.shape[1]
    X = torch.empty(T, d, device=E.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_embed_sum_fwd(_p(_i32(ids)), _p(_f32(E)), _p(_f32(P)), _p(S0), float(scale), T, L, d, _p(X), _stream()), "embed_sum_fwd")
    return X


def dropact_fwd(X, p, seed, site, idx_offset=0, act=ACT_NONE):
    Y = torch.empty_like(X)
    _lib.check(_lib.load().adt_dropact_fwd(_p(_f32(X)), X.numel(), float(p), _p(seed), site, idx_offset, act, _p(Y), _stream()), "dropact_fwd")
    return Y


def dropact_bwd(dY, X, p, seed, site, dX, accumulate, idx_offset=0, act=ACT_NONE):
    _lib.check(_lib.load().adt_dropact_bwd(_p(_f32(dY)), _p(_f32(X)), X.numel(), float(p), _p(seed), site, idx_offset, act, _p(dX), int(accumulate),
                                           _stream()), "dropact_bwd")


def gather_rows(F, rows, M=None, m_dev=None):
    M = rows.numel() if M is None else M
    d = F.shape[1]
    out = torch.empty(M, d, device=F.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_gather_rows(_p(_f32(F)), _ld(F), _p(_i32(rows)), M, _p(m_dev), d, _p(out), d, _stream()), "gather_rows")
    return out


def scatter_rows(G, rows, dF, accumulate, M=None, m_dev=None):
    M = rows.numel() if M is None else M
    _lib.check(_lib.load().adt_scatter_rows(_p(_f32(G)), _ld(G), _p(_i32(rows)), M, _p(m_dev), G.shape[1], _p(dF), _ld(dF), int(accumulate), _stream()),
               "scatter_rows")


def ce_rows(logits, labels, V, inv_count, loss64, M=None, m_dev=None):
    M = logits.shape[0] if M is None else M
    _lib.check(_lib.load().adt_ce_rows(_p(_f32(logits)), _ld(logits), _p(_i32(labels)), M, _p(m_dev), V, _p(inv_count), _p(loss64), _stream()), "ce_rows")


def lce_supported(prec, K):
    return bool(_lib.load().adt_lce_supported(prec, K))


_LCE_WS = {}


def lce_fwd_bwd(h, rows, labels, mcap, m_dev, E, bias, inv_count, loss64, dh, dE, dbias, lse_out=None):
    """Fused all-item logits + cross-entropy on the masked rows (include/adt_hip.h: adt_lce_fwd_bwd).  The workspace is cached per
    (device, mcap, V, K): a captured graph keeps pointing at it."""
    V, K = E.shape
    lib = _lib.load()
    key = (h.device, mcap, V, K, lib.adt_lce_slots(0))
    ws = _LCE_WS.get(key)
    if ws is None:
        ws = _LCE_WS[key] = torch.empty(int(lib.adt_lce_workspace_bytes(mcap, V, K)), device=h.device, dtype=torch.uint8)
    _lib.check(lib.adt_lce_fwd_bwd(_p(_f32(h)), _ld(h), _p(_i32(rows)), _p(_i32(labels)), mcap, _p(m_dev), _p(_f32(E)), _ld(E), _p(bias), V, K,
                                   _p(inv_count), _p(loss64), _p(lse_out), _p(dh), _ld(dh) if dh is not None else 0, _p(dE), _ld(dE), _p(dbias),
                                   _p(ws), ws.numel(), _stream()), "lce_fwd_bwd")


def clip_adam_l2(P, G, M, V, l2, clip, lr, b1, b2, eps, scal, grad_scale=1.0, n=None):
    n = P.numel() if n is None else n
    _lib.check(_lib.load().adt_clip_adam_l2(_p(P), _p(G), _p(M), _p(V), n, float(l2), float(clip), float(lr), float(b1), float(b2), float(eps),
                                            float(grad_scale), _p(scal), _stream()), "clip_adam_l2")


def score_rank_bias(F, ldf, E, bias, cand, B, C, want_rank=True):
    d = E.shape[1]
    logits = torch.empty(B, C, device=E.device, dtype=torch.float32)
    rank = torch.empty(B, device=E.device, dtype=torch.int32) if want_rank else None
    _lib.check(_lib.load().adt_score_rank_bias(_p(F), ldf, _p(E), _p(bias), _p(cand), B, C, d, _p(logits), _p(rank), _stream()), "score_rank_bias")
    return logits, rank


def wattn_fwd(Qm, Qc, Km, Kc, Vm, Vc, key_ids, B, H, L, p=0.0, seed=None, site=0, b_offset=0, prec=None):
    """prec: PREC_BF16 / PREC_F32 selects the matrix-core kernels (adt_wattn_mfma.cuh) where they cover the shape; None (or an
    uncovered shape) the exact vector-ALU kernels."""
    d = Qm.shape[1]
    hd = d // H
    Om = torch.empty(B * L, d, device=Qm.device, dtype=torch.float32)
    Oc = torch.empty_like(Om)
    LSE = torch.empty(B * H * L, device=Qm.device, dtype=torch.float32)
    if prec is not None:
        rc = _lib.load().adt_wattn_mfma_fwd(int(prec), _p(_f32(Qm)), _ld(Qm), _p(_f32(Qc)), _ld(Qc), _p(_f32(Km)), _ld(Km), _p(_f32(Kc)), _ld(Kc),
                                            _p(_f32(Vm)), _ld(Vm), _p(_f32(Vc)), _ld(Vc), _p(_i32(key_ids)), B, H, L, hd, float(p), _p(seed), site,
                                            b_offset, _p(Om), d, _p(Oc), d, _p(LSE), _stream())
        if rc != 1:
            _lib.check(rc, "wattn_mfma_fwd")
            return Om, Oc, LSE
    _lib.check(_lib.load().adt_wattn_fwd(_p(_f32(Qm)), _ld(Qm), _p(_f32(Qc)), _ld(Qc), _p(_f32(Km)), _ld(Km), _p(_f32(Kc)), _ld(Kc), _p(_f32(Vm)), _ld(Vm),
                                         _p(_f32(Vc)), _ld(Vc), _p(_i32(key_ids)), B, H, L, hd, float(p), _p(seed), site, b_offset, _p(Om), d, _p(Oc), d,
                                         _p(LSE), _stream()), "wattn_fwd")
    return Om, Oc, LSE


def wattn_bwd(Qm, Qc, Km, Kc, Vm, Vc, key_ids, Om, Oc, LSE, dOm, dOc, B, H, L, p=0.0, seed=None, site=0, b_offset=0, out=None, prec=None):
    """out: optional (dQm, dQc, dKm, dKc, dVm, dVc) views sharing one row stride.  prec: as wattn_fwd (use the same value)."""
    d = Qm.shape[1]
    hd = d // H
    outs = [torch.empty(B * L, d, device=Qm.device, dtype=torch.float32) for _ in range(6)] if out is None else list(out)
    ldd = _ld(outs[0])
    assert all(_ld(o) == ldd for o in outs)
    if prec is not None:
        rc = _lib.load().adt_wattn_mfma_bwd(int(prec), _p(_f32(Qm)), _ld(Qm), _p(_f32(Qc)), _ld(Qc), _p(_f32(Km)), _ld(Km), _p(_f32(Kc)), _ld(Kc),
                                            _p(_f32(Vm)), _ld(Vm), _p(_f32(Vc)), _ld(Vc), _p(_i32(key_ids)), _p(Om), _ld(Om), _p(Oc), _ld(Oc), _p(LSE),
                                            _p(_f32(dOm)), _ld(dOm), _p(_f32(dOc)), _ld(dOc), B, H, L, hd, float(p), _p(seed), site, b_offset,
                                            *[_p(o) for o in outs], ldd, _stream())
        if rc != 1:
            _lib.check(rc, "wattn_mfma_bwd")
            return outs
    _lib.check(_lib.load().adt_wattn_bwd(_p(_f32(Qm)), _ld(Qm), _p(_f32(Qc)), _ld(Qc), _p(_f32(Km)), _ld(Km), _p(_f32(Kc)), _ld(Kc), _p(_f32(Vm)), _ld(Vm),
                                         _p(_f32(Vc)), _ld(Vc), _p(_i32(key_ids)), _p(Om), _ld(Om), _p(Oc), _ld(Oc), _p(LSE), _p(_f32(dOm)), _ld(dOm),
                                         _p(_f32(dOc)), _ld(dOc), B, H, L, hd, float(p), _p(seed), site, b_offset, *[_p(o) for o in outs], ldd, _stream()),
               "wattn_bwd")
    return outs


def wdist_bpr(Sm, Sc, Em, Ec, pos, neg, pvn_weight, inv_count, dEm, dEc, loss3):
    T, d = Sm.shape
    dSm = torch.empty(T, d, device=Sm.device, dtype=torch.float32)
    dSc = torch.empty_like(dSm)
    _lib.check(_lib.load().adt_wdist_bpr(_p(_f32(Sm)), _p(_f32(Sc)), _ld(Sm), _p(Em), _p(Ec), _p(_i32(pos)), _p(_i32(neg)), T, d, float(pvn_weight),
                                         _p(inv_count), _p(dSm), _p(dSc), d, _p(dEm), _p(dEc), _p(loss3), _stream()), "wdist_bpr")
    return dSm, dSc


def wdist_full(Sm, Sc, Em, Ec, V):
    B, d = Sm.shape
    dist = torch.empty(B, V, device=Sm.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_wdist_full(_p(_f32(Sm)), _p(_f32(Sc)), _ld(Sm), _p(Em), _p(Ec), B, V, d, _p(dist), V, _stream()), "wdist_full")
    return dist


def topk_masked(dist, k, indptr=None, indices=None, want_val=False):
    """k smallest entries per row in ascending order after pushing the CSR-listed columns to 1e24; `dist` is consumed."""
    B, N = dist.shape
    idx = torch.empty(B, k, device=dist.device, dtype=torch.int32)
    val = torch.empty(B, k, device=dist.device, dtype=torch.float32) if want_val else None
    _lib.check(_lib.load().adt_topk_masked(_p(dist), dist.stride(0), B, N, _p(indptr), _p(indices), k, _p(idx), _p(val), _stream()), "topk_masked")
    return (idx, val) if want_val else idx


def axpy(dst, src, alpha=1.0, accumulate=True, mask_ids=None, d=0):
    """dst = (accumulate ? dst : 0) + alpha * src * rowmask."""
    _lib.check(_lib.load().adt_axpy(_p(dst), _p(_f32(src)), float(alpha), int(accumulate), src.numel(), _p(mask_ids), d, _stream()), "axpy")
    return dst


def log_softmax_fwd(X, H):
    Y = torch.empty_like(X)
    _lib.check(_lib.load().adt_log_softmax_fwd(_p(_f32(X)), X.numel() // H, H, _p(Y), _stream()), "log_softmax_fwd")
    return Y


def log_softmax_bwd(Y, dY, H, dX, accumulate):
    _lib.check(_lib.load().adt_log_softmax_bwd(_p(Y), _p(_f32(dY)), Y.numel() // H, H, _p(dX), int(accumulate), _stream()), "log_softmax_bwd")


def grad_sumsq(G, out64):
    _lib.check(_lib.load().adt_grad_sumsq(_p(G), G.numel(), _p(out64), _stream()), "grad_sumsq")


def adam_range(P, G, M, V, l2, clip, lr, b1, b2, eps, step, gn2_slots):
    _lib.check(_lib.load().adt_adam_range(_p(P), _p(G), _p(M), _p(V), P.numel(), float(l2), float(clip), float(lr), float(b1), float(b2), float(eps),
                                          float(step), _p(gn2_slots), _stream()), "adam_range")


def adamw_range(P, G, M, V, wd, clip, lr, b1, b2, eps, step, gn2_slots):
    """adam_range with torch.optim.AdamW's decoupled weight decay."""
    _lib.check(_lib.load().adt_adamw_range(_p(P), _p(G), _p(M), _p(V), P.numel(), float(wd), float(clip), float(lr), float(b1), float(b2), float(eps),
                                           float(step), _p(gn2_slots), _stream()), "adamw_range")


def dense_gradsrc(dY, act, U, p=0.0, seed=None, site=0, row_offset=0, mask_ids=None, t_dev=None):
    """G = dY * rowmask * dropmask * act'(U), materialised (see adt_dense_gradsrc)."""
    T, N = dY.shape
    G = torch.empty(T, N, device=dY.device, dtype=torch.float32)
    _lib.check(_lib.load().adt_dense_gradsrc(_p(_f32(dY)), _ld(dY), T, N, _p(mask_ids), float(p), _p(seed), site, row_offset, act, _p(U), _ld(U),
                                             _p(G), N, _p(t_dev), _stream()), "dense_gradsrc")
    return G


def dense_rows_enable(on):
    """bf16 dense layers: True = row-streaming kernels where the shape allows (default), False = always the tiled kernels.
    Returns the previous setting."""
    return bool(_lib.load().adt_dense_rows_enable(int(bool(on))))


# ---- deterministic item-table / positional-table gradient (include/adt_hip.h: adt_item_sort ...) -----------------------------------------
def _ptr_array(tensors, n):
    arr = (ctypes.c_void_p * n)()
    for i in range(n):
        t = tensors[i] if i < len(tensors) else None
        arr[i] = None if t is None else t.data_ptr()
    return arr


def item_sort(ids_list, V1, rows, coef, kind, row_offset=0):
    """Sorts the entries (src, t) of the int32 id tensors in ids_list (same length T each) by item and records the gather plan (rows / coef /
    kind per source: include/adt_hip.h); returns the work buffer for item_segsum."""
    lib = _lib.load()
    nsrc, T = len(ids_list), ids_list[0].numel()
    work = torch.empty(lib.adt_item_sort_work_ints(nsrc, T, V1) + 2, device=ids_list[0].device, dtype=torch.int32)
    kinds = (ctypes.c_int * 4)(*[int(kind[i]) if i < len(kind) else 0 for i in range(4)])
    _lib.check(lib.adt_item_sort(_ptr_array([_i32(t) for t in ids_list], 4), nsrc, T, V1, _ptr_array(rows, 4), _ptr_array(coef, 4), kinds,
                                 row_offset, _p(work), _stream()), "item_sort")
    work._adt_keep = (rows, coef)          # the plan holds their addresses
    return work


def item_segsum(work, nsrc, T, V1, src_mask, site, p, seed, emb_scale, dE, accumulate=False):
    """dE[item] (+)= the sorted entries of `item` from the sources in src_mask (adt_item_segsum)."""
    sites = (ctypes.c_uint32 * 4)(*[int(site[i]) if i < len(site) else 0 for i in range(4)])
    _lib.check(_lib.load().adt_item_segsum(_p(work), nsrc, T, V1, src_mask, sites, float(p), _p(seed), float(emb_scale), _p(_f32(dE)),
                                           int(bool(accumulate)), _stream()), "item_segsum")


def posemb_sum(ids_list, dX_list, sites, B, L, p, seed, row_offset, dP):
    n = len(ids_list)
    s = (ctypes.c_uint32 * 2)(*[int(sites[i]) if i < n else 0 for i in range(2)])
    _lib.check(_lib.load().adt_posemb_sum(_ptr_array([_i32(t) for t in ids_list], 2), _ptr_array(dX_list, 2), s, n, B, L, float(p), _p(seed),
                                          row_offset, _p(_f32(dP)), _stream()), "posemb_sum")
