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

/* ---- SASRecADT.predict scoring (sasrec/model.py:89-96) + rank of evaluate_loader (sasrec/utils.py:410).
 * cand == NULL scores items 0..C-1 (full=True). rank may be NULL. */
int adt_score_rank(const float* F, int ldf, const float* E, const int32_t* cand, int B, int C, int d,
                   float* logits, int32_t* rank, void* stream);

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

/* SASRecADT.forward (sasrec/model.py:67-81).  ids are device int32 (B*L). training != 0 enables dropout. */
int adt_sasrec_forward(const adt_sasrec_cfg* cfg, const float* params, float* ws, const int32_t* seq,
                       const int32_t* dec, const int32_t* pos, const int32_t* neg, int B, int training,
                       const uint32_t* seed, uint32_t b_offset, void* stream);
/* loss assembly (sasrec/main.py:151-169): fills G_POS/G_NEG, G_ENC_X[0..nl-1], G_DEC_X[1..nl], G_REC and the
 * loss slots from the forward activations.  lambdas1/lambdas2: host arrays of num_layers floats; the NLL
 * weight is lambdas2[num_layers-1] for every layer (stale loop index, sasrec/main.py:169).  NORMS must
 * already hold the global normalisers. */
int adt_sasrec_loss_seed(const adt_sasrec_cfg* cfg, float* ws, const int32_t* pos, int B, const float* lambdas1,
                         const float* lambdas2, void* stream);
/* reverse pass: consumes the G_* buffers (destroyed), accumulates into `grads` (same layout as params).
 * phase: 0 = everything; 1 = logits + decoder stack only; 2 = last LN + encoder stack + embeddings (lets the
 * host overlap the gradient all-reduce of the decoder bucket with phase 2). */
int adt_sasrec_backward(const adt_sasrec_cfg* cfg, const float* params, float* grads, float* ws,
                        const int32_t* seq, const int32_t* dec, const int32_t* pos, const int32_t* neg, int B,
                        int training, const uint32_t* seed, uint32_t b_offset, int phase, void* stream);
/* SASRecADT.predict (sasrec/model.py:83-97): encoder only, last position, candidate (cand != NULL, B x C) or
 * all-item (C = V+1) scores; optional rank of column 0. */
int adt_sasrec_predict(const adt_sasrec_cfg* cfg, const float* params, float* ws, const int32_t* seq,
                       const int32_t* cand, int B, int C, float* logits, int32_t* rank, void* stream);

#ifdef __cplusplus
}
#endif
#endif
