// Argument block of the wave-local forward row-chain kernels (adt_fwdchain.cuh); shared with the host executor.
#pragma once
#include "adt_common.cuh"

namespace adt {

struct FwdChainArgs {
  int T, L, B, H;
  const int* ids;                 // padding mask / gather ids of this stack (seq or dec)
  DropCfg drop; uint32_t site0, site1, site2; uint32_t row_offset;   // site0: embedding, site1/2: ffn1/ffn2
  float ln_eps;
  // inputs
  const float* x;                 // pre: layer input (null => gather) ; post/mid: attention output o ; final: encoder out
  const float* r0;                // enc_post: qn (residual) ; dec_post: dn (residual)
  const float* E; const float* P; float emb_scale;      // gather / logits
  const int* pos; const int* neg;
  // weights: up to 4 (64 x 64, row-major) + biases
  const float* W[4]; const float* b[4];
  const float* gamma; const float* beta;
  const float* Ws; const float* bs;                     // head classifier
  // outputs (null = do not store)
  float* o0; int ld0;             // pre: x (gathered input)      ; post: h / a2 ; mid: a1 ; final: f
  float* o1; int ld1;             // pre: xn (LayerNorm output)   ; post: u
  float* o2; int ld2;             // pre: q block (ld 192)        ; post: y      ; mid: q2 ; final: kv2 of decoder layer 0 (ld 128)
  float* o3; int ld3;             // final: kv2 of decoder layer 1 (ld 128)
  float* rec;                     // enc_post: (L*B, H, H) log-probabilities
  float* pos_logits; float* neg_logits;
  int nkv;                        // final: number of decoder layers served (1 or 2 per launch)
};

}  // namespace adt
