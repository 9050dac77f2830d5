/* adt_hip.h -- C ABI of libadt_hip.so: the MI355X (gfx950) hot path of the Adaptive Disentangled
 * Transformer sequential recommender (reference: defineZYP/ADT, read-only at /root/reference).
 *
 * The reference has no FFI/operator layer: the hot path sits behind plain Python nn.Modules that call
 * ATen.  This header is the boundary the build inserts between those modules and the
 * device.  Every entry point below names the reference call site (file:line, relative to the reference
 * root) whose arithmetic it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch-ROCm caching allocator in our host
 *     code); nothing is allocated, freed or synchronised inside; every call only enqueues kernels on
 *     `stream` (a hipStream_t passed as void*), so calls may be captured into a hipGraph.
 *   - tensors are fp32, token-major: a (B, L, d) activation is B*L rows of d floats; "ld" = row stride in
 *     floats; head h of a projection lives in columns [h*hd, (h+1)*hd).  ids are int32.
 *   - prec: ADT_PREC_F32 (v_mfma_f32_16x16x4_f32, exact fp32) or ADT_PREC_BF16 (operands rounded to bf16,
 *     v_mfma_f32_16x16x32_bf16, fp32 accumulate).
 *   - dropout: stateless hash RNG (adt_common.cuh / oracle/rng.py).  `seed` points at a uint32 in DEVICE
 *     memory; p == 0 disables.  Element indices are GLOBAL (row_offset / b_offset shift a data-parallel
 *     shard), so an N-rank run draws the same masks as the 1-rank run.
 *   - return 0 on success, negative on error (message via adt_last_error()); no exceptions, re-entrant.
 */
#ifndef ADT_HIP_H
#define ADT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADT_PREC_F32 0
#define ADT_PREC_BF16 1

int adt_version(void);
const char* adt_last_error(void);
/* hash RNG known-answer helper (host): keep decision for (seed, site, idx, p) */
int adt_rng_keep(uint32_t seed, uint32_t site, uint32_t idx, float p);

/* ---- embedding: sasrec/model.py:34-41 and :53-59 ---------------------------------------------------
 * X[row] = dropout(E[ids[row]] * sqrt(d) + P[row % L]) * (ids[row] != 0) */
int adt_embed_fwd(const int32_t* ids, const float* E, const float* P, int T, int L, int d, float p,
                  const uint32_t* seed, uint32_t site, uint32_t row_offset, float* X, void* stream);
/* dE[ids] += dX*mask*sqrt(d), dP[l] += dX*mask (autograd of the lines above; dE/dP are accumulated) */
int adt_embed_bwd(const int32_t* ids, const float* dX, int T, int L, int d, float p, const uint32_t* seed,
                  uint32_t site, uint32_t row_offset, float* dE, float* dP, void* stream);

/* ---- LayerNorm over d (eps inside sqrt): torch.nn.LayerNorm at sasrec/modules.py:638,640,660, model.py:28 */
int adt_layernorm_fwd(const float* X, int ldx, const float* gamma, const float* beta, float eps, int T, int d,
                      float* Y, int ldy, void* stream);
int adt_layernorm_bwd(const float* dY, int lddy, const float* X, int ldx, const float* gamma, float eps, int T,
                      int d, float* dX, int lddx, int accumulate, float* dgamma, float* dbeta, void* stream);

/* ---- Linear d -> N: _in_projection_packed (sasrec/modules.py:84-137), out_proj (:519), Conv1d k=1
 * (:623-633).  Y = mask( R1 + R2 + relu?( dropout( X W^T + b ) ) ).  N multiple of 16. */
int adt_linear_fwd(int prec, const float* X, int ldx, const float* W, const float* b, int T, int K, int N,
                   float* Y, int ldy, float p, const uint32_t* seed, uint32_t site, uint32_t row_offset,
                   int relu, const float* R1, int ldr1, const float* R2, int ldr2, const int32_t* mask_ids,
                   void* stream);
/* dYp = dY * mask * dropmask * (U > 0);  dX = (beta ? dX : 0) + dYp W + Radd*mask(radd_ids);
 * dW += dYp^T X;  db += colsum(dYp). */
int adt_linear_bwd(int prec, const float* dY, int lddy, const float* X, int ldx, const float* W, int T, int K,
                   int N, const int32_t* mask_ids, float p, const uint32_t* seed, uint32_t site,
                   uint32_t row_offset, const float* U, int ldu, float* dX, int lddx, int beta,
                   const float* Radd, int ldradd, const int32_t* radd_ids, float* dW, float* db, void* stream);

/* ---- attention core: _scaled_dot_product_attention (sasrec/modules.py:21-64) incl. head split/merge
 * (:457-468,:517); also torch.nn.MultiheadAttention's core for the decoder (:661-662,:669-672).
 * O = dropout(softmax(Q K^T / sqrt(hd) + causal)) V per (b, h); LSE saved for the backward.
 * mask (optional, B*H*L x 8 uint32): the bf16 forward stores its dropout keep decisions as bits (query-major, bit
 * key % 32 of word key / 32) and the backward reads them instead of re-hashing; NULL = regenerate from the hash. */
int adt_attn_fwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int B,
                 int H, int L, int hd, int causal, float p, const uint32_t* seed, uint32_t site,
                 uint32_t b_offset, float* O, int ldo, float* LSE, uint32_t* mask, void* stream);
int adt_attn_bwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                 const float* O, int ldo, const float* LSE, const float* dO, int lddo, int B, int H, int L,
                 int hd, int causal, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset,
                 float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, const uint32_t* mask, void* stream);

/* ---- independence head classifier: SparseInputLinear + log_softmax (sasrec/modules.py:648-649,679-703).
 * rec rows are in the reference's order (row l*B + b = token (b, l); sasrec/modules.py:518). */
int adt_headcls_fwd(const float* O, int ldo, const float* Ws, const float* bs, int B, int L, int H, int hd,
                    float* rec, void* stream);
int adt_headcls_bwd(const float* O, int ldo, const float* Ws, const float* rec, const float* drec, int B, int L,
                    int H, int hd, float* dO, int lddo, float* dWs, float* dbs, void* stream);

/* ---- pos/neg logits: sasrec/model.py:72-76 */
int adt_logits_fwd(const float* F, int ldf, const float* E, const int32_t* pos, const int32_t* neg, int T, int d,
                   float* pos_logits, float* neg_logits, void* stream);
int adt_logits_bwd(const float* F, int ldf, const float* E, const int32_t* pos, const int32_t* neg,
                   const float* dpos, const float* dneg, int T, int d, float* dF, int lddf, float* dE,
                   void* stream);

/* Contention-relieved form of the item-table scatter-add used by the executor: `rep` holds nrep zeroed replicas of
 * the (V+1) x d table (rep_stride floats apart); rows are spread over the replicas.  adt_item_scatter adds
 * rowscale[row] * scale * dropmask * G[row] into rep[wave % nrep][ids[row]]; adt_replica_reduce adds all replicas to dE. */
int adt_item_scatter(const int32_t* ids, const float* G, int ldg, const float* rowscale, int T, int d, float scale, float p,
                     const uint32_t* seed, uint32_t site, uint32_t row_offset, float* rep, int nrep, int64_t rep_stride,
                     void* stream);
int adt_replica_reduce(float* dE, const float* rep, int64_t n, int nrep, int64_t rep_stride, void* stream);
/* The encoder embedding gradient, the decoder embedding gradient and the positive-logit rows (dpos[t] * F[t]) into the item-table replicas in one
 * pass (d = 64, T a multiple of L) -- sasrec/model.py:34-41, :53-59, :72-76 reversed.  The reference's sampler (sasrec/utils.py:288-307) makes
 * seq[b, l] == dec[b, l + 1] == pos[b, l - 1] for almost every token: those three rows go out as ONE atomic row-add; ids that do not line up
 * are added on their own (any ids are correct).  dP += the positional sums of both embeddings (scale sqrt(d) on the item rows only). */
int adt_embed_bwd3(const int32_t* seq, const int32_t* dec, const int32_t* pos, const float* dXs, const float* dXd, const float* F, const float* dpos,
                   int T, int L, float p, const uint32_t* seed, uint32_t site_seq, uint32_t site_dec, uint32_t row_offset, float* dP, float* rep,
                   int nrep, int64_t rep_stride, void* stream);
/* position-table half of adt_embed_bwd alone: dP[l] += sum_b dX[b, l] * dropmask * (ids != 0) */
int adt_posemb_bwd(const int32_t* ids, const float* dX, int T, int L, int d, float p, const uint32_t* seed,
                   uint32_t site, uint32_t row_offset, float* dP, void* stream);
/* dF = dpos * E[pos] + dneg * E[neg] alone (no item-table scatter) */
int adt_logits_bwd_df(const float* E, const int32_t* pos, const int32_t* neg, const float* dpos, const float* dneg, int T,
                      int d, float* dF, int lddf, void* stream);

/* ---- loss seeds: sasrec/main.py:151-153 (BCE), :155-158 (MSE), :160-169 (NLL).  norms = device
 * {n_bce, n_mse, n_nll} (global normalisers).  Every loss term is accumulated over 64 consecutive floats (sub-slots,
 * to avoid same-address atomic contention): bce: loss2[0..64) = pos term, [64..128) = neg term; the reader sums. */
int adt_bce_seed(const float* pos_logits, const float* neg_logits, const int32_t* pos, int T, const float* norms,
                 float* dpos, float* dneg, float* loss2, void* stream);
int adt_mse_seed(const float* A, const float* Bm, int64_t n, float lambda, const float* norms, float* GA,
                 int accumulate_a, float* GB, float* loss1, void* stream);
int adt_nll_seed(const float* rec, int n_rows, int H, float lambda2, const float* norms, float* drec,
                 float* loss1, void* stream);

/* ---- sasrec/main.py:170-173: grad += wd * E/||E||_F on the item table (flat offset 0, nE floats),
 * clip_grad_norm_(clip), Adam(lr, (b1, b2), eps).  scal = 192 device floats: [0] ||E||^2, [1] ||g||^2, [2] step
 * (incremented), [3] wd*||E||, [64..192) partial-sum slots zeroed inside. */
int adt_clip_adam(float* P, float* G, float* M, float* V, int64_t n, int64_t nE, float wd, float clip, float lr,
                  float b1, float b2, float eps, float grad_scale, float* scal, void* stream);
/* The same for a step opened by adt_sasrec_step_begin (which has already stored the partial sums of ||E||^2 in scal[64..128) and zeroed
 * scal[128..192)): two launches instead of four. */
int adt_clip_adam_pre(float* P, float* G, float* M, float* V, int64_t n, int64_t nE, float wd, float clip, float lr,
                      float b1, float b2, float eps, float grad_scale, float* scal, void* stream);

/* ---- SASRecADT.predict scoring (sasrec/model.py:89-96) + rank of evaluate_loader (sasrec/utils.py:410).
 * cand == NULL scores items 0..C-1 (full=True). rank may be NULL. */
int adt_score_rank(const float* F, int ldf, const float* E, const int32_t* cand, int B, int C, int d,
                   float* logits, int32_t* rank, void* stream);

/* ==== general ("wide") stage kernels: BERT4Rec-ADT, STOSA-ADT and d > 64 configurations ======================= */
#define ADT_ACT_NONE 0
#define ADT_ACT_RELU 1
#define ADT_ACT_GELU 2   /* nn.GELU(), erf form (bert4rec/model/modules.py:125) */
#define ADT_ACT_ELU 3    /* nn.ELU() (stosa/modules.py:477) */
#define ADT_ACT_ELU1 4   /* ELU(x) + 1 (stosa/modules.py:236-238) */

/* G[t][n] = dY[t][n] * (mask_ids[t] != 0) * dropmask * act'(U[t][n]): the gradient reaching the pre-activation of a dense layer
 * (the backward of the epilogue of adt_dense_fwd: bert4rec/model/modules.py:128-139 GELU + dropout, sasrec/modules.py:618-633 relu
 * + dropout), written once so that adt_dense_bwd can run on it with act = none, p = 0 (same arithmetic as its fused prologue). */
int adt_dense_gradsrc(const float* dY, int lddy, int T, int N, const int32_t* mask_ids, float p, const uint32_t* seed,
                      uint32_t site, uint32_t row_offset, int act, const float* U, int ldu, float* G, int ldg,
                      const int32_t* t_dev, void* stream);
/* ---- fused all-item logits + cross-entropy on the masked rows (bf16 operands; K = 128 or 256) ---------------------------------
 * Replaces, for BERT4Rec-ADT's output layer, `logits = h @ word_emb^T + bias` over all V (+100) items (bert4rec/model/bert.py:80-90)
 * followed by CrossEntropyLoss(ignore_index=0) (bert4rec/trainer.py:45,113-115) and their backward, without ever writing the
 * (rows x V) logits: an online log-sum-exp forward and a backward that recomputes each score tile on the matrix cores
 * (adt_amd/csrc/adt_lce.cuh).  h: (B*L, K) activations, row stride ldh; rows[m] (m < min(mcap, *m_dev)): the rows whose label is
 * non-zero, labels[m] their labels (1..V-1); E: (V, K) item table, bias: (V).  With n = 1 / *inv_count labelled rows:
 *   sum(loss64[0..64)) += sum_m (lse_m - logit_m[label_m]) / n  (64 partial slots);   lse_out[m] = lse_m (optional);
 *   dh[rows[m]] = sum_v (softmax_m[v] - [v == label_m]) / n * E[v]            (plain store; dh NULL = loss only)
 *   dE[v] += sum_m (softmax_m[v] - [v == label_m]) / n * h[rows[m]];  dbias[v] += sum_m (...)      (accumulated)
 * m_dev (optional) is a DEVICE count, so a captured graph replays with a different number of masked rows.  workspace: at least
 * adt_lce_workspace_bytes(mcap, V, K) bytes, 256-byte aligned, contents irrelevant on entry. */
int adt_lce_supported(int prec, int K);
/* Workgroups the passes are split over (default: the device's CU count); slots > 0 sets it (tests use a small grid so that every
 * workgroup walks several work items), returns the value in force.  The workspace size depends on it. */
int adt_lce_slots(int slots);
int64_t adt_lce_workspace_bytes(int mcap, int V, int K);
int adt_lce_fwd_bwd(const float* h, int ldh, const int32_t* rows, const int32_t* labels, int mcap, const int32_t* m_dev,
                    const float* E, int lde, const float* bias, int V, int K, const float* inv_count, float* loss64,
                    float* lse_out, float* dh, int lddh, float* dE, int lddE, float* dbias, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* Kernel selection for the dense layers in bf16 mode: 1 (default) = row-streaming kernels (adt_dense_rows.cuh) where the shape
 * allows (contraction 64/128/256 per chunk, N <= 1024), 0 = always the tiled kernels.  Returns the previous setting.  Results
 * agree to bf16 rounding either way; the switch exists for A/B measurements and tests. */
int adt_dense_rows_enable(int on);
/* Optional scratch for the dense backward: with at least 64 MiB registered (256 workgroups x 256 KiB), the weight gradient of a 256 x 256
 * layer in bf16 mode runs through private per-workgroup partials + a reduce instead of a 128-256 KiB atomic flush per workgroup.  The
 * library keeps the pointer (it allocates nothing itself); ws = NULL unregisters.  Results agree with the atomic path to summation order. */
int adt_dense_workspace(void* ws, int64_t bytes);
/* torch.nn.Linear with its surrounding elementwise ops, any K / N (bert4rec/model/modules.py:59-75,128-139,
 * bert.py:48-51,80-90; stosa/modules.py:199-212,477-487): Y = mask(R + dropout(act(X W^T + b))); W is N x K with row
 * stride ldw; U (optional) receives the pre-activation X W^T + b for the backward.  t_dev (optional, DEVICE int): only
 * rows < min(T, *t_dev) are computed -- batches of masked rows whose count changes per step under a captured graph. */
int adt_dense_fwd(int prec, const float* X, int ldx, const float* W, int ldw, const float* b, int T, int K, int N,
                  int act, float* U, int ldu, float p, const uint32_t* seed, uint32_t site, uint32_t row_offset,
                  const float* R, int ldr, const float* R2, int ldr2, const int32_t* mask_ids, float* Y, int ldy,
                  const int32_t* t_dev, void* stream);
/* G = dY * mask * dropmask * act'(U);  dX = (beta ? dX : 0) + G W (dX NULL = skip);  dW += G^T X, db += colsum(G)
 * (dW NULL = skip; db may be NULL). */
int adt_dense_bwd(int prec, const float* dY, int lddy, int T, int K, int N, const int32_t* mask_ids, float p,
                  const uint32_t* seed, uint32_t site, uint32_t row_offset, int act, const float* U, int ldu,
                  const float* X, int ldx, const float* W, int ldw, float* dX, int lddx, int beta, float* dW, int lddw,
                  float* db, const int32_t* t_dev, void* stream);

/* Masked attention core, bidirectional or causal, with key padding: bert4rec/model/modules.py:76-101.  Scores of masked
 * keys (key_ids[b*L + j] <= 0, or j > i when causal) are REPLACED by `fill` (-1e9 in the reference) and carry no
 * gradient; key_ids may be NULL. */
int adt_attn_masked_fwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int B, int H,
                        int L, int hd, int causal, const int32_t* key_ids, float fill, float p, const uint32_t* seed,
                        uint32_t site, uint32_t b_offset, float* O, int ldo, float* LSE, void* stream);
int adt_attn_masked_bwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O,
                        int ldo, const float* LSE, const float* dO, int lddo, int B, int H, int L, int hd, int causal,
                        const int32_t* key_ids, float fill, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset,
                        float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, void* stream);

/* X[row] = E[ids[row]] * scale + P[row % L] (+ S0): bert4rec/model/modules.py:42-46, stosa/models.py:183-210 */
int adt_embed_sum_fwd(const int32_t* ids, const float* E, const float* P, const float* S0, float scale, int T, int L,
                      int d, float* X, void* stream);
/* Y = act(dropout(X)) over n elements (index idx_offset + i); backward dX (+)= dY * act'(dropout(X)) * keep/(1-p) */
int adt_dropact_fwd(const float* X, int64_t n, float p, const uint32_t* seed, uint32_t site, uint32_t idx_offset, int act,
                    float* Y, void* stream);
int adt_dropact_bwd(const float* dY, const float* X, int64_t n, float p, const uint32_t* seed, uint32_t site,
                    uint32_t idx_offset, int act, float* dX, int accumulate, void* stream);
/* out[i] = F[rows[i]];  dF[rows[i]] (+)= G[i] (distinct rows): the masked positions of bert4rec/trainer.py:113-115.
 * m_dev (optional, DEVICE int) caps M like t_dev above. */
int adt_gather_rows(const float* F, int ldf, const int32_t* rows, int M, const int32_t* m_dev, int d, float* out, int ldo,
                    void* stream);
int adt_scatter_rows(const float* G, int ldg, const int32_t* rows, int M, const int32_t* m_dev, int d, float* dF, int lddf,
                     int accumulate, void* stream);
/* nn.CrossEntropyLoss(ignore_index=0) (bert4rec/trainer.py:45,113-115) on M rows of V logits: loss64[m & 63] += w *
 * (lse - z[label]); logits overwritten by w * (softmax - onehot); w = *inv_count; label 0 => zero row. */
int adt_ce_rows(float* logits, int ld, const int32_t* labels, int M, const int32_t* m_dev, int V, const float* inv_count,
                float* loss64, void* stream);
/* dst = (accumulate ? dst : 0) + alpha * src * (mask_ids == NULL || mask_ids[i / d] != 0): the candidate mixing of the
 * supernet (sasrec/super_modules.py:42-49, :79-83) and masked residual gradients */
int adt_axpy(float* dst, const float* src, float alpha, int accumulate, int64_t n, const int32_t* mask_ids, int d,
             void* stream);
/* log_softmax over rows of H <= 8 values (sasrec/super_modules.py:49) / dX (+)= dY - exp(Y) * sum(dY) */
int adt_log_softmax_fwd(const float* X, int64_t rows, int H, float* Y, void* stream);
int adt_log_softmax_bwd(const float* Y, const float* dY, int64_t rows, int H, float* dX, int accumulate, void* stream);
/* adt_clip_adam plus Adam's coupled weight_decay l2 (g += l2 * p after clipping): bert4rec/trainer.py:41,137-138 */
int adt_clip_adam_l2(float* P, float* G, float* M, float* V, int64_t n, float l2, float clip, float lr, float b1, float b2,
                     float eps, float grad_scale, float* scal, void* stream);
/* Optimizer pieces for the supernet (sasrec/evolution.py:109,314-316): the squared norm of ALL gradients into 64 partial
 * slots (clip_grad_norm_ is global), then Adam with coupled weight decay on one flat range with the caller's step count for
 * that range -- torch skips parameters whose grad is None (the candidate layers that were not mixed in), so every
 * candidate layer carries its own step count and moments. */
int adt_grad_sumsq(const float* G, int64_t n, float* out64, void* stream);
int adt_adam_range(float* P, float* G, float* M, float* V, int64_t n, float l2, float clip, float lr, float b1, float b2,
                   float eps, float step, const float* gn2_slots, void* stream);
/* the same with torch.optim.AdamW's DECOUPLED weight decay (p *= 1 - lr * wd, then the Adam update): the optimizer of the BERT4Rec-ADT
 * supernet warm-up (bert4rec/evolution.py:101) */
int adt_adamw_range(float* P, float* G, float* M, float* V, int64_t n, float wd, float clip, float lr, float b1, float b2,
                    float eps, float step, const float* gn2_slots, void* stream);
/* adt_score_rank with a per-item bias (bert4rec/model/bert.py:89,110-116) */
int adt_score_rank_bias(const float* F, int ldf, const float* E, const float* bias, const int32_t* cand, int B, int C,
                        int d, float* logits, int32_t* rank, void* stream);

/* ---- STOSA-ADT: Wasserstein attention (stosa/modules.py:30-43,222-275,311-361).  Qc/Kc/Vc are covariances (already
 * ELU(.)+1); key j is masked (additively, -2^32) when key_ids[b*L+j] <= 0 or j > i.  Om = Pd Vm, Oc = (Pd*Pd) Vc. */
int adt_wattn_fwd(const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                  const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, int B, int H, int L, int hd,
                  float p, const uint32_t* seed, uint32_t site, uint32_t b_offset, float* Om, int ldom, float* Oc, int ldoc,
                  float* LSE, void* stream);
int adt_wattn_bwd(const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                  const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, const float* Om, int ldom,
                  const float* Oc, int ldoc, const float* LSE, const float* dOm, int lddom, const float* dOc, int lddoc, int B,
                  int H, int L, int hd, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset, float* dQm, float* dQc,
                  float* dKm, float* dKc, float* dVm, float* dVc, int ldd, void* stream);
/* The same attention on the matrix cores (one 16x16x32 MFMA per score tile over the concatenation [mean | sqrt cov]); prec: ADT_PREC_BF16
 * (bf16 operands) or ADT_PREC_F32 (exact fp32 MFMA).  Return 1 = shape not covered (head size 16 / 32, L <= 128): use the functions above. */
int adt_wattn_mfma_fwd(int prec, const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                       const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, int B, int H, int L, int hd, float p,
                       const uint32_t* seed, uint32_t site, uint32_t b_offset, float* Om, int ldom, float* Oc, int ldoc, float* LSE,
                       void* stream);
int adt_wattn_mfma_bwd(int prec, const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                       const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, const float* Om, int ldom, const float* Oc,
                       int ldoc, const float* LSE, const float* dOm, int lddom, const float* dOc, int lddoc, int B, int H, int L, int hd, float p,
                       const uint32_t* seed, uint32_t site, uint32_t b_offset, float* dQm, float* dQc, float* dKm, float* dKc, float* dVm,
                       float* dVc, int ldd, void* stream);
/* bpr_optimization (stosa/trainer.py:358-391), loss and gradients in one pass: loss3 = 3 x 64 slots {bpr, pvn_weight *
 * pvn, auc}, all already divided by sum(istarget) (= 1 / *inv_count); dSm/dSc overwritten, dEm/dEc accumulated. */
int adt_wdist_bpr(const float* Sm, const float* Sc, int lds, const float* Em, const float* Ec, const int32_t* pos,
                  const int32_t* neg, int T, int d, float pvn_weight, const float* inv_count, float* dSm, float* dSc, int ldds,
                  float* dEm, float* dEc, float* loss3, void* stream);
/* dist_predict_full (stosa/trainer.py:464-479): dist[b][v] = W2(state b, item v), items 0..V-1 */
int adt_wdist_full(const float* Sm, const float* Sc, int lds, const float* Em, const float* Ec, int B, int V, int d,
                   float* dist, int ldo, void* stream);

/* Full-sort selection (stosa/trainer.py:598-612: rating_pred[train_matrix[users].toarray() > 0] = 1e+24, np.argpartition(.., 40),
 * np.argsort of the 40): per row b, set dist[b][indices[indptr[b]..indptr[b+1])] = 1e24 (CSR of the seen items; both NULL = no
 * mask), then write the k smallest entries in ascending order (ties: smaller id) to out_idx (B, k) / out_val (B, k, may be
 * NULL).  `dist` (B rows of N floats, row stride ld) is consumed. */
int adt_topk_masked(float* dist, int ld, int B, int N, const int32_t* indptr, const int32_t* indices, int k, int32_t* out_idx,
                    float* out_val, void* stream);

/* ==== model-level executor: SASRecADT (sasrec/model.py:8-97) + loop body (sasrec/main.py:146-173) ====== */
typedef struct adt_sasrec_cfg {
  int32_t item_num;     /* V; item table has V+1 rows                      */
  int32_t maxlen;       /* L                                               */
  int32_t hidden;       /* d (64 or 256)                                   */
  int32_t num_heads;    /* H                                               */
  int32_t num_layers;   /* encoder blocks == decoder blocks               */
  float dropout;        /* p                                               */
  int32_t prec;         /* ADT_PREC_*                                      */
} adt_sasrec_cfg;

/* Flat fp32 parameter buffer layout.  slots: [0]=item_emb [1]=pos_emb [2]=last_ln.w [3]=last_ln.b, then 14
 * per encoder layer {ln1.w, ln1.b, in_proj.w, in_proj.b, out_proj.w, out_proj.b, ln2.w, ln2.b, conv1.w,
 * conv1.b, conv2.w, conv2.b, sparse.w, sparse.b}, then 16 per decoder layer {ln.w, ln.b, slf.in_w, slf.in_b,
 * slf.out_w, slf.out_b, enc.in_w, enc.in_b, enc.out_w, enc.out_b, conv1.w, conv1.b, conv2.w, conv2.b,
 * unused_ln.w, unused_ln.b}.  offsets has 4 + 30*num_layers entries (in floats); returns total floats. */
int64_t adt_sasrec_param_layout(const adt_sasrec_cfg* cfg, int64_t* offsets);
/* workspace size (floats) for batch B, and the offset of a named activation inside it */
int64_t adt_sasrec_workspace_floats(const adt_sasrec_cfg* cfg, int B);
#define ADT_WS_ENC_X 0      /* layer i: input of encoder layer i (== enc_in[i]); i = num_layers: encoder out */
#define ADT_WS_DEC_X 1      /* layer i: input of decoder layer i; i+1: its output (dec_out_rev[j] = DEC_X[nl-j]) */
#define ADT_WS_REC 2        /* layer i: (L*B, H, H) log-probs in reference row order                         */
#define ADT_WS_POS_LOGITS 3
#define ADT_WS_NEG_LOGITS 4
#define ADT_WS_F 5          /* log_feats (post last_layernorm)                                                */
#define ADT_WS_G_ENC_X 6    /* gradient buffers, same indexing as ENC_X / DEC_X / REC / logits               */
#define ADT_WS_G_DEC_X 7
#define ADT_WS_G_REC 8
#define ADT_WS_G_POS 9
#define ADT_WS_G_NEG 10
#define ADT_WS_LOSS 11      /* (2 + 2*num_layers) x 64 floats: bce_pos, bce_neg, mse_i.., nll_l.. (64 sub-slots each) */
#define ADT_WS_NORMS 12     /* 3 floats: n_bce, n_mse, n_nll                                                   */
#define ADT_WS_SCAL 13      /* 192 floats for adt_clip_adam                                                    */
int64_t adt_sasrec_ws_offset(const adt_sasrec_cfg* cfg, int B, int what, int layer);

/* SASRecADT.forward (sasrec/model.py:67-81).  ids are device int32 (B*L).  training bit 0 enables dropout; bit 1 (value 2): the bf16
 * weight images of this step were already packed by adt_sasrec_step_begin / _ring (the forward then skips its own packing launch). */
int adt_sasrec_forward(const adt_sasrec_cfg* cfg, const float* params, float* ws, const int32_t* seq,
                       const int32_t* dec, const int32_t* pos, const int32_t* neg, int B, int training,
                       const uint32_t* seed, uint32_t b_offset, void* stream);
/* Measurement hook (bench.py `roofline`): launches ONLY the fused forward of decoder layer `layer` (one workgroup per sequence) on the
 * workspace of a completed adt_sasrec_forward of the same batch; bf16 mode, d = 64.  Not part of the reference's interface. */
int adt_sasrec_probe_dec_layer_fwd(const adt_sasrec_cfg* cfg, const float* params, float* ws, const int32_t* dec, int B, int training,
                                   const uint32_t* seed, uint32_t b_offset, int layer, void* stream);
/* ---- One EncoderLayer / DecoderLayer (sasrec/modules.py:644-655, :666-677) per call, on the per-sequence fused kernels: what the
 * supernet (sasrec/super_modules.py:35-50,74-85) runs for each of its four selected candidate layers.  bf16 operands, d = 64,
 * head size 16 / 32 / 64, L <= 224, L % 4 == 0 (adt_seq_layer_supported).  Pointers are device fp32 tensors of the layer (parameters, or
 * their gradient accumulators in the backward: gradients are ADDED).  wp_base / wp_img: the packed bf16 weight images of the 64x64
 * blocks (adt_pack_wimg over the same flat buffer: image of block W at wp_img + 6 * (W - wp_base) bf16 elements).  `save`: the layer's
 * saved activations (adt_seq_layer_save_floats floats), written by the forward, read by the backward.  y = (y_acc ? y : 0) + y_scale *
 * layer(x): the candidate-mixing epilogue; the backward takes the gradient of the MIXED output and the same scale. */
typedef struct { float *ln1_w, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_w, *ln2_b, *c1_w, *c1_b, *c2_w, *c2_b, *cls_w, *cls_b; } adt_enc_layer_ptrs;
typedef struct { float *ln_w, *ln_b, *sin_w, *sin_b, *so_w, *so_b, *ein_w, *ein_b, *eo_w, *eo_b, *c1_w, *c1_b, *c2_w, *c2_b; } adt_dec_layer_ptrs;
int adt_seq_layer_supported(int prec, int L, int d, int hd);
int64_t adt_seq_layer_save_floats(int B, int L, int H, int dec);
int adt_pack_wimg(const float* base, void* img, const int* offs, int n, void* stream);   /* n <= 256 blocks at base + offs[i] */
int adt_seq_enc_layer_fwd(int B, int L, int H, const int32_t* ids, const float* x, const adt_enc_layer_ptrs* P, const float* wp_base,
                          const void* wp_img, float p, const uint32_t* seed, uint32_t site_attn, uint32_t site_ffn1, uint32_t site_ffn2,
                          uint32_t b_offset, int training, float* save, float* y, float y_scale, int y_acc, float* rec, void* stream);
/* gy: gradient of the mixed output (B*L x 64); drec: gradient of this candidate's head-classifier log-probabilities (reference row order,
 * may be null); gx: gradient of the layer input (gx_acc: add); scratch: 2 * B*L*64 floats */
int adt_seq_enc_layer_bwd(int B, int L, int H, const int32_t* ids, const float* x, const adt_enc_layer_ptrs* P, const adt_enc_layer_ptrs* G,
                          const float* wp_base, const void* wp_img, float p, const uint32_t* seed, uint32_t site_attn, uint32_t site_ffn1,
                          uint32_t site_ffn2, uint32_t b_offset, const float* save, const float* gy, float gy_scale, const float* rec,
                          const float* drec, float* gx, int gx_acc, float* scratch, void* stream);
int adt_seq_dec_layer_fwd(int B, int L, int H, const int32_t* ids, const float* x, const float* feats, const adt_dec_layer_ptrs* P,
                          const float* wp_base, const void* wp_img, float p, const uint32_t* seed, uint32_t site_slf, uint32_t site_enc,
                          uint32_t site_ffn1, uint32_t site_ffn2, uint32_t b_offset, float* save, float* y, float y_scale, int y_acc,
                          void* stream);
/* gfeats: gradient of `feats` (ADDED); scratch: 4 * B*L*64 floats */
int adt_seq_dec_layer_bwd(int B, int L, int H, const int32_t* ids, const float* x, const float* feats, const adt_dec_layer_ptrs* P,
                          const adt_dec_layer_ptrs* G, const float* wp_base, const void* wp_img, float p, const uint32_t* seed,
                          uint32_t site_slf, uint32_t site_enc, uint32_t site_ffn1, uint32_t site_ffn2, uint32_t b_offset, const float* save,
                          const float* gy, float gy_scale, float* gx, int gx_acc, float* gfeats, float* scratch, void* stream);
/* loss assembly (sasrec/main.py:151-169): fills G_POS/G_NEG, G_ENC_X[0..nl-1], G_DEC_X[1..nl], G_REC and the
 * loss slots from the forward activations.  lambdas1/lambdas2: host arrays of num_layers floats; the NLL
 * weight is lambdas2[num_layers-1] for every layer (stale loop index, sasrec/main.py:169).  NORMS must
 * already hold the global normalisers. */
int adt_sasrec_loss_seed(const adt_sasrec_cfg* cfg, float* ws, const int32_t* pos, int B, const float* lambdas1,
                         const float* lambdas2, void* stream);
/* One launch for everything a training step (sasrec/main.py:146-173) does before its forward: grads[0..n) = 0 (optimizer.zero_grad), the loss
 * slots = 0, NORMS = norms_src[0..4), *seed += seed_inc (the per-step dropout stream), scal[128..192) = 0 and scal[64..128) = partial sums of
 * ||item table||^2 (the weight-decay term of adt_clip_adam_pre; the parameters do not change in between).  The same launch zeroes the
 * parameter-gradient replicas the backward chains flush into and, in bf16 mode, packs the step's weight images (adt_pack_wimg's work):
 * tell the forward (training | 2) and the backward (phase | 4) of the same step.  Pair it with adt_sasrec_loss_seed_nz
 * (adt_sasrec_loss_seed without its loss-slot fill) and adt_clip_adam_pre. */
int adt_sasrec_step_begin(const adt_sasrec_cfg* cfg, float* ws, int B, uint32_t* seed, uint32_t seed_inc, const float* norms_src,
                          const float* params, float* grads, int64_t n, float* scal, void* stream);
/* adt_sasrec_step_begin that also FETCHES the step's id batch (sasrec/main.py:144-145: the batch the DataLoader hands the loop becomes device
 * tensors at model.py:34,37,40,53).  `ring` holds nslots blocks of slot_ints int32 each, in pinned HOST memory (the kernel reads it over
 * PCIe: no copy-engine transfer, no cross-queue wait between "copy" and "step") or in HBM; one block is the packed batch
 * [seq | dec | pos | neg] (4 B L ids) + the three loss normalisers as float bits + a zero word.  Slot (state[0] % nslots) is copied to
 * ids_dst (4 B L + 4 int32) and its normalisers go to NORMS; then state[0] += 1 and, if `consumed` is not NULL, the new count is stored to
 * *consumed (a pinned host word the producer polls before it refills a slot: a slot written for step k is free once *consumed > k).
 * `state` = two zero-initialised device uint32 (step count, block ticket).  slot_ints a multiple of 4, >= 4 B L + 4. */
int adt_sasrec_step_begin_ring(const adt_sasrec_cfg* cfg, float* ws, int B, uint32_t* seed, uint32_t seed_inc, const int32_t* ring,
                               int64_t slot_ints, int nslots, int32_t* ids_dst, uint32_t* state, uint32_t* consumed, const float* params,
                               float* grads, int64_t n, float* scal, void* stream);
/* The same with a PREFETCH of the next batch: `state` = eight zero-initialised device uint32 (batches fetched, ticket, staged batch + 1, prefetch
 * ticket, producer count as seen, statistics: batches taken from `staging`), `staging` = device buffer of 4 B L + 4 int32, `produced` = pinned host word holding the number of batches
 * the producer has completely written (slot k is published by storing k + 1 AFTER its ids and normalisers).  This call samples *produced;
 * adt_sasrec_forward_loss_prefetch of the same step then copies the NEXT batch's slot into `staging` beside its loss-assembly pass if it
 * was published (and releases the slot through *consumed), and this call of the next step takes the batch from `staging` (HBM) instead of
 * the ring (PCIe: ~16 us at the flagship batch, in front of everything else of the step). */
int adt_sasrec_step_begin_ring_staged(const adt_sasrec_cfg* cfg, float* ws, int B, uint32_t* seed, uint32_t seed_inc, const int32_t* ring,
                                      int64_t slot_ints, int nslots, int32_t* ids_dst, uint32_t* state, uint32_t* consumed, const int32_t* staging,
                                      const uint32_t* produced, const float* params, float* grads, int64_t n, float* scal, void* stream);
int adt_sasrec_loss_seed_nz(const adt_sasrec_cfg* cfg, float* ws, const int32_t* pos, int B, const float* lambdas1,
                            const float* lambdas2, void* stream);
/* ---- Deterministic item-table / positional-table gradient (sasrec/model.py:34-41, :53-59, :72-76 reversed; d = 64) -----------------------
 * The reference's autograd adds the rows that hit one item in a fixed order; float-atomic scatters add in arrival order.  Here the step's id
 * arrays are sorted once (stable counting sort of the entries e = src * T + t by item: a pure function of the ids) and every item's rows are
 * summed by one owner in sorted order -- no float atomics, two runs give the same bits.  item_num + 1 <= 16,000 (adt_item_sort_supported).
 * work: adt_item_sort_work_ints(nsrc, T, item_num + 1) int32 of scratch, written by adt_item_sort, read by adt_item_segsum. */
int adt_item_sort_supported(int V1);
int64_t adt_item_sort_work_ints(int nsrc, int T, int V1);
/* Sorts the entries and records the gather plan: rows[s] (T x 64 floats) are the rows source s contributes, kind[s] = 0: the gradient of
 * an embedding layer's output, summed as rows * emb_scale * keep / (1 - p) with keep = the forward's dropout decision at site[s] (element
 * index (t + row_offset) * 64 + f) ; kind[s] = 1: rows[s] * coef[s][t] (the logits' item rows: log_feats * dlogit).  Only the POINTERS are
 * recorded: the values are read by adt_item_segsum.  work: 8-byte aligned. */
int adt_item_sort(const int32_t* const* ids, int nsrc, int T, int V1, const float* const* rows, const float* const* coef, const int* kind,
                  uint32_t row_offset, int32_t* work, void* stream);
/* dE[item] = (accumulate ? dE[item] : 0) + sum of the sorted entries of `item` from the sources in src_mask (bit s = id array s of
 * adt_item_sort); rows of items without entries are left alone (zero dE first).  accumulate = 0 is the fast form: plain stores. */
int adt_item_segsum(const int32_t* work, int nsrc, int T, int V1, uint32_t src_mask, const uint32_t* site, float p, const uint32_t* seed,
                    float emb_scale, float* dE, int accumulate, void* stream);
/* dP[l] += sum_b [ids[b, l] != 0] * keep / (1 - p) * dX[b, l] for nsrc (1 or 2) embedding layers, b ascending (one owner per position) */
int adt_posemb_sum(const int32_t* const* ids, const float* const* dX, const uint32_t* site, int nsrc, int B, int L, float p, const uint32_t* seed,
                   uint32_t row_offset, float* dP, void* stream);
/* adt_item_segsum + adt_posemb_sum (same p / seed) as one launch (+ the carry launch): the two are independent and each alone leaves most
 * of the chip idle */
int adt_item_segsum_posemb(const int32_t* work, int nsrc, int T, int V1, uint32_t src_mask, const uint32_t* site, float p, const uint32_t* seed,
                           float emb_scale, float* dE, int accumulate, const int32_t* const* pos_ids, const float* const* pos_dX,
                           const uint32_t* pos_site, int pos_nsrc, int B, int L, uint32_t row_offset, float* dP, void* stream);
/* adt_sasrec_forward + adt_sasrec_loss_seed_nz of one training step (sasrec/model.py:67-81 + sasrec/main.py:151-169) in one call.  When
 * adt_sasrec_bce_deferred(cfg) is 1 (bf16, d = 64, the lean per-sequence kernels cover the shape, <= 4 blocks) and training == 3 (dropout on,
 * weight images packed by adt_sasrec_step_begin* of this step), log_feats is written by the last encoder layer's own kernel and the pos / neg
 * logits + BCE seed + BCE loss terms are NOT formed here: adt_sasrec_backward of the same step must be called with phase bit 4 (+ 16) and
 * forms them in its first side kernel, which gathers the same rows.  Otherwise the two calls in sequence.  ADT_FWD_FUSED=0: never deferred. */
/* training bit 2 (value 4, with bits 0 and 1, deferred path only): the logits / BCE / item-row kernel is launched by THIS call on the library's
 * side stream beside the loss pass (adt_sasrec_step_begin* of the step zeroed the item-table replicas); adt_sasrec_backward of the step takes
 * phase bit 5 (+ 32) instead of bit 4 and joins it. */
int adt_sasrec_bce_deferred(const adt_sasrec_cfg* cfg);
int adt_sasrec_forward_loss(const adt_sasrec_cfg* cfg, const float* params, float* ws, const int32_t* seq, const int32_t* dec,
                            const int32_t* pos, const int32_t* neg, int B, int training, const uint32_t* seed, uint32_t b_offset,
                            const float* lambdas1, const float* lambdas2, void* stream);
/* adt_sasrec_forward_loss + the prefetch half of adt_sasrec_step_begin_ring_staged (ring == NULL: plain adt_sasrec_forward_loss) */
int adt_sasrec_forward_loss_prefetch(const adt_sasrec_cfg* cfg, const float* params, float* ws, const int32_t* seq, const int32_t* dec,
                                     const int32_t* pos, const int32_t* neg, int B, int training, const uint32_t* seed, uint32_t b_offset,
                                     const float* lambdas1, const float* lambdas2, const int32_t* ring, int64_t slot_ints, int nslots,
                                     uint32_t* state, uint32_t* consumed, int32_t* staging, void* stream);
/* reverse pass: consumes the G_* buffers (destroyed), accumulates into `grads` (same layout as params).
 * phase: 0 = everything; 1 = logits + decoder stack only; 2 = last LN + encoder stack + embeddings (lets the
 * host overlap the gradient all-reduce of the decoder bucket with phase 2).  + 4: the item-table and parameter-gradient replicas were already
 * zeroed by adt_sasrec_step_begin / _ring of this step (phase 2 zeroes the item-table replicas again for its own scatter).  With phase 0 the scatter / fold kernels run on a side stream of the library under the
 * chain kernels (joined before the call returns its last launch; ADT_SIDE_STREAM=0 keeps everything on `stream`).  + 8 (with phase 0 only):
 * the last fold of the gradient replicas into `grads` is left to adt_sasrec_fold_clip_adam, which must follow.  + 16: the step's forward was
 * adt_sasrec_forward_loss on the deferred path (adt_sasrec_bce_deferred): logits, BCE seed and BCE loss terms are formed here.  + 32 (instead
 * of + 16, with + 4, phase 0 or 1): that forward was called with training bit 2 and launched the kernel itself: only its join is left.
 * + 64 (phase 2 of a two-phase backward behind such a forward; implied by + 16 / + 32): the reconstruction seeds of the block inputs were not
 * materialised by the forward -- the attention-block backward forms each from its own input rows and the other stack's. */
int adt_sasrec_backward(const adt_sasrec_cfg* cfg, const float* params, float* grads, float* ws,
                        const int32_t* seq, const int32_t* dec, const int32_t* pos, const int32_t* neg, int B,
                        int training, const uint32_t* seed, uint32_t b_offset, int phase, void* stream);
/* The optimizer step of a single-GPU training step whose backward ran with phase 0 + 8 (and whose scal was prepared by
 * adt_sasrec_step_begin): the last fold of the gradient replicas, grad += wd * E / ||E||_F on the item table and the partial sums of
 * ||g||^2 in ONE pass over the gradient, then clip_grad_norm_(clip) + Adam (sasrec/main.py:170-173) -- adt_clip_adam_pre's result in two
 * launches instead of three.  m, v: Adam moments laid out like params. */
int adt_sasrec_fold_clip_adam(const adt_sasrec_cfg* cfg, float* ws, int B, float* params, float* grads, float* m, float* v, float wd,
                              float clip, float lr, float b1, float b2, float eps, float* scal, void* stream);
/* The fold half of adt_sasrec_fold_clip_adam alone: completes `grads` behind adt_sasrec_backward(phase | 8) -- no weight-decay term, no
 * norm, no optimizer step.  The data-parallel step runs it in front of the gradient all-reduce (sasrec/main.py:170 on every rank, then one
 * sum over the ranks) and adt_clip_adam_pre behind it. */
int adt_sasrec_fold_grads(const adt_sasrec_cfg* cfg, float* ws, int B, float* params, float* grads, float* scal, void* stream);
/* SASRecADT.predict (sasrec/model.py:83-97): encoder only, last position, candidate (cand != NULL, B x C) or
 * all-item (C = V+1) scores; optional rank of column 0. */
int adt_sasrec_predict(const adt_sasrec_cfg* cfg, const float* params, float* ws, const int32_t* seq,
                       const int32_t* cand, int B, int C, float* logits, int32_t* rank, void* stream);

#ifdef __cplusplus
}
#endif
#endif
