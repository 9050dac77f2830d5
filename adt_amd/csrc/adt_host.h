// Host-side shared declarations for libadt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/adt_hip.h"
#include "adt_common.cuh"

int adt_set_error(const char* fmt, ...);
DropCfg adt_make_drop(float p, const uint32_t* seed, uint32_t site);

// dropout sites (identical to oracle/sasrec_oracle.py)
enum { SITE_EMB_SEQ = 1, SITE_EMB_DEC = 2 };
static inline uint32_t enc_site(int layer, int which) { return 16u + 8u * (uint32_t)layer + (uint32_t)which; }   // 0 attn 1 ffn1 2 ffn2
static inline uint32_t dec_site(int layer, int which) { return 128u + 8u * (uint32_t)layer + (uint32_t)which; }  // 0 slf 1 enc 2 ffn1 3 ffn2

// fused backward chains (adt_bwdchain.cuh); defined in adt_capi.hip.  which: 0 enc_post, 1 dec_post, 2 enc_pre, 3 dec_pre, 4 dec_mid
namespace adt { struct BwdChainArgs; }
int adt_launch_bwdchain(int prec, int which, const adt::BwdChainArgs& a, void* stream);

// wave-local forward chains (adt_fwdchain.cuh); which: 0 enc_pre, 1 dec_pre, 2 enc_post, 3 dec_mid, 4 dec_post, 5 final
namespace adt { struct FwdChainArgs; }
int adt_launch_fwdchain(int prec, int which, const adt::FwdChainArgs& a, void* stream);

// per-sequence fused layer kernels (adt_seqfwd.cuh); defined in adt_seq.hip
namespace adt { struct SeqFwdArgs; }
int adt_seq_supported(int prec, int L, int d, int hd);      // bf16 mode, d = 64, L <= 224; ADT_SEQ=0 in the environment turns them off
int adt_seq_lean(int prec, int L, int d, int hd);           // the forward may save bf16 tensors and skip LN(x) / qkv (see adt_seq.hip)
int adt_launch_seq_enc_fwd(int hd, const adt::SeqFwdArgs& a, void* stream);
int adt_launch_seq_dec_fwd(int hd, const adt::SeqFwdArgs& a, void* stream);
// adt_pack_wimg (pre-packed bf16 weight images) is part of the C ABI now: include/adt_hip.h; defined in adt_seq.hip
namespace adt { struct AttnArgs; }
int adt_launch_seq_attn_bwd(int hd, const adt::AttnArgs& a, void* stream);     // 0 launched, 1 shape not covered, < 0 error
// adt_attn_bwd with Q, K, V, O saved as bf16 rows (ld* of those four count bf16 elements); per-sequence kernel only: anything it does not
// cover is an error.  Defined in adt_capi.hip.
int adt_attn_bwd_saved_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo, const float* LSE,
                            const float* dO, int lddo, int B, int H, int L, int hd, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset,
                            float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, const uint32_t* mask, int out_bf16, void* stream);
// per-sequence backward of the token-wise chains (adt_seqpost_tt.cuh); enc: 1 encoder post chain, 0 decoder post chain; 0 launched, 1 not covered
int adt_launch_seq_post_bwd(int hd, int enc, const adt::BwdChainArgs& a, void* stream);
int adt_launch_seq_mid_bwd(int hd, const adt::BwdChainArgs& a, void* stream);      // dec_mid + kv chains in one launch
// private per-workgroup partials of the 64 x 64 weight gradients instead of atomics (adt_seqbwd_tt.cuh: sb_dw_tiles) and their sum
int adt_seq_partials(int prec, int L, int d, int hd);
int adt_dwpart_reduce(float* G, const float* part, size_t stride, int nwg, const int* slots, const int* offs, int nslots, void* stream);
// the same with a workgroup count per slot (nwg_slot[i] workgroups wrote slot i ; null: nwg everywhere)
int adt_dwpart_reduce_n(float* G, const float* part, size_t stride, int nwg, const int* nwg_slot, const int* slots, const int* offs, int nslots, void* stream);
namespace adt { struct SeqBwdArgs; }
int adt_launch_seq_attn_pre_bwd(int hd, int dec, const adt::SeqBwdArgs& a, void* stream);     // 0 launched, 1 not covered, < 0 error

// fused small kernels of the flagship backward (adt_misc.cuh; defined in adt_capi.hip): d log_feats + both item scatters in one pass,
// positional + item embedding gradient in one pass
extern "C" {
int adt_logits_bwd_scatter(const float* F, int ldf, const float* E, const int32_t* pos, const int32_t* neg, const float* dpos, const float* dneg,
                           int T, int d, float* dF, int lddf, float* rep, int nrep, int64_t rep_stride, void* stream);
// d = 64: logits, BCE seed + loss terms, d log_feats and the item rows of the positive / negative items in one pass (adt_misc.cuh)
int adt_logits_bce_scatter(const float* F, const float* E, const int32_t* pos, const int32_t* neg, const float* norms, int T, float* pos_logits,
                           float* neg_logits, float* dpos, float* dneg, float* loss_bce, float* dF, float* rep, int nrep, int64_t rep_stride, void* stream);
int adt_embed_bwd_rep(const int32_t* ids, const float* dX, int T, int L, int d, float p, const uint32_t* seed, uint32_t site, uint32_t row_offset,
                      float* dP, float* rep, int nrep, int64_t rep_stride, void* stream);
int adt_layernorm_bwd_rep(const float* dY, int lddy, const float* X, int ldx, const float* gamma, float eps, int T, int d, float* dX, int lddx,
                          int accumulate, float* dgamma, float* dbeta, int nrep, int64_t rep_stride, void* stream);
// Z / nz: a second range to zero (or null) ; pack_*: npack 64 x 64 blocks at pack_base + pack_offs[i] packed into pack_img (adt_pack_wimg's work) or null
int adt_step_begin_launch(uint32_t* seed, uint32_t inc, float* norms_dst, const float* norms_src, float* loss, int nloss, float* scal, float* G,
                          int64_t n, const float* E, int64_t nE, float* Z, int64_t nz, const float* pack_base, void* pack_img, const int* pack_offs,
                          int npack, void* stream);
int adt_step_begin_ring_launch(uint32_t* seed, uint32_t inc, float* norms_dst, float* loss, int nloss, float* scal, float* G, int64_t n, const float* E,
                               int64_t nE, const int32_t* ring, int64_t slot_ints, int nslots, int32_t* ids_dst, int64_t n_ints, uint32_t* state,
                               uint32_t* consumed, const int32_t* staging, const uint32_t* produced, float* Z, int64_t nz, const float* pack_base,
                               void* pack_img, const int* pack_offs, int npack, void* stream);
// adt_loss_seeds + the prefetch of the next step's id batch into `staging` as extra workgroups of the same launch (adt_misc.cuh: ring_prefetch_body)
void adt_loss_seeds_split_prefetch();
void adt_loss_seeds_attach_logits(const float* F, const float* E, const int32_t* pos, const int32_t* neg, const float* norms, int T, float* pos_logits,
                                  float* neg_logits, float* dpos, float* dneg, float* loss_bce, float* dF, float* rep, int nrep, int64_t rep_stride,
                                  int neg_only);
int adt_logits_bce_scatter_ex(const float* F, const float* E, const int32_t* pos, const int32_t* neg, const float* norms, int T, float* pos_logits,
                              float* neg_logits, float* dpos, float* dneg, float* loss_bce, float* dF, float* rep, int nrep, int64_t rep_stride,
                              int neg_only, void* stream);
int adt_loss_seeds_prefetch(const float* pos_logits, const float* neg_logits, const int32_t* pos, int T, const float* norms, float* dpos, float* dneg,
                            float* loss_bce, int nmse, const float* const* A, const float* const* Bm, int64_t n, const float* lambdas, float* const* GA,
                            int accumulate_a, float* const* GB, float* const* loss_mse, int nnll, const float* const* rec, int n_rows, int H, float lambda2,
                            float* const* drec, float* const* loss_nll, const int32_t* ring, int64_t slot_ints, int nslots, int64_t n_ints,
                            uint32_t* state, uint32_t* consumed, int32_t* staging, void* stream);
int adt_loss_seeds(const float* pos_logits, const float* neg_logits, const int32_t* pos, int T, const float* norms, float* dpos, float* dneg,
                   float* loss_bce, int nmse, const float* const* A, const float* const* Bm, int64_t n, const float* lambdas, float* const* GA,
                   int accumulate_a, float* const* GB, float* const* loss_mse, int nnll, const float* const* rec, int n_rows, int H, float lambda2,
                   float* const* drec, float* const* loss_nll, void* stream);
int adt_fold_clip_adam(float* P, float* G, float* M, float* V, int64_t n, float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1,
                       const float* r1, int64_t n1, int nrep1, int64_t s1, float wd, float clip, float lr, float b1, float b2, float eps, float* scal,
                       void* stream);
int adt_replica_reduce2(float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1, const float* r1, int64_t n1, int nrep1, int64_t s1,
                        void* stream);
int adt_fold_parts_clip_adam(float* P, float* G, float* M, float* V, int64_t n, float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1,
                             const float* r1, int64_t n1, int nrep1, int64_t s1, const float* part, int64_t part_stride, const int* nwg_slot, const int* slots,
                             const int* offs, int nslots, const float* vpart, const int* vsrc, const int* vnwg, const int* vstride, const int* voff, int nvec,
                             float* gn_part, float wd, float clip, float lr, float b1, float b2, float eps, float* scal, void* stream);
// its fold half alone (no weight-decay term, no optimizer step)
int adt_fold_parts(float* P, float* G, int64_t n, float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1, const float* r1, int64_t n1,
                   int nrep1, int64_t s1, const float* part, int64_t part_stride, const int* nwg_slot, const int* slots, const int* offs, int nslots,
                   const float* vpart, const int* vsrc, const int* vnwg, const int* vstride, const int* voff, int nvec, float* gn_part, float* scal,
                   void* stream);
int adt_layernorm_bwd_parts(const float* dY, int lddy, const float* X, int ldx, const float* gamma, float eps, int T, int d, float* dX, int lddx,
                            int accumulate, float* part, int max_blocks, void* stream);
}
