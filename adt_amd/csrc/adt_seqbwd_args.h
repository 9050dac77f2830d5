// Argument block of the per-sequence fused attention-block backward (adt_seqbwd_tt.cuh); shared with the host executor.
#pragma once
#include "adt_common.cuh"

namespace adt {

struct SeqBwdArgs {
  int L, B, H;
  const int* ids;                     // (B*L) padding mask of this stack
  DropCfg drop; uint32_t b_offset;    // attention dropout (site set by the host), global index of this shard's first sequence
  float ln_eps, scale;
  const float* x;                     // block input (B*L x 64): the encoder / decoder layer input
  const float* gamma; const float* beta;           // LayerNorm in front of the in-projection
  const float* Win; const float* bin;              // packed in-projection (3 x 64 x 64, 3 x 64)
  const float* dO; const float* o;                 // gradient wrt the attention output ; the forward's attention output
  const float* lse; const uint32_t* mask;          // (B*H*L) log-sum-exp ; (B*H*L x 8) dropout keep bits
  float dres_scale;                   // decoder only: the residual gradient is dres * dres_scale (0 means 1; the supernet's mixing weight)
  const float* dres;                  // encoder: gradient wrt LN(x) from the residual path ; decoder: gradient wrt the layer output (masked here)
  float* gx; int acc;                 // gradient wrt x (acc: add to what is there)
  const float* seed_other; const float* seed_coef; float* seed_loss; const float* seed_norms;
                                      // non-null: the reconstruction term's seed of this layer input is formed here instead of read from gx:
                                      // gx = dx + *seed_coef * (x - seed_other)  (sasrec/main.py:155-158: 2 lambda1 (a - b) / n, a = this block's
                                      // input, b = the other stack's tensor of the pair; *seed_coef is written by k_loss_seeds; acc must be 0).
                                      // seed_loss != nullptr: this launch also adds the pair's loss term sum (a - b)^2 / seed_norms[1] to its 64 sub-slots
  float* dWin; float* dbin; float* dgamma; float* dbeta;     // accumulators (global float atomics)
  float* part; size_t part_stride;    // non-null: private partials of dWin (3 x 4096 floats at part + workgroup * part_stride) instead of atomics
  int nrep; size_t rep_stride;        // parameter-gradient replicas (adt_bwdchain_args.h): workgroup b adds into replica b % nrep
  const float* wp_base; const void* wp_img;        // pre-packed weight images (slot-ordered: + 2 plain, + 3 transposed)
  int saved_bf16;                     // `o` holds bf16 rows (written by the transposed-chain forward in its lean mode)
  float* vpart;                       // non-null: dgamma[64] | dbeta[64] | dbin[192] of this workgroup stored at vpart + blockIdx * 512 (no atomics)
  int nsplit;                         // 2: two workgroups per sequence (grid 2 B; workgroup 2 b + part owns the tiles t % 2 == part; needs `part`)
  unsigned long long* stamps;
};

}  // namespace adt
