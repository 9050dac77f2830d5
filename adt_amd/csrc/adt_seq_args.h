// Argument blocks of the per-sequence fused layer kernels (adt_seqfwd.cuh); shared with the host executor.
#pragma once
#include "adt_common.cuh"

namespace adt {

// One workgroup = one user sequence = one whole EncoderLayer / DecoderLayer forward (sasrec/modules.py:644-655, :666-677).
struct SeqFwdArgs {
  int L, B, H;
  const int* ids;                      // (B*L) ids of this stack: padding mask, and the gather ids when x == nullptr
  DropCfg drop;                        // probability / scale / device seed; the site is set per use from the fields below
  uint32_t site_emb, site_attn, site_attn2, site1, site2;
  uint32_t b_offset;                   // global index of this shard's first sequence (dropout indices are global)
  float ln_eps, scale;                 // scale = 1/sqrt(head size)
  const float* x;                      // layer input (B*L x 64); nullptr => x = dropout(E[id]*emb_scale + P[l]) * (id != 0)
  const float* E; const float* P; float emb_scale;
  const float* gamma; const float* beta;           // LayerNorm in front of the in-projection
  const float* Win; const float* bin;              // packed in-projection of the (self) attention: 3 x (64 x 64), 3 x 64
  const float* Wo; const float* bo;                // its out_proj
  const float* gamma2; const float* beta2;         // encoder: forward_layernorm
  const float* Ws; const float* bs;                // encoder: head classifier (used when rec != nullptr)
  const float* f;                                  // decoder: log_feats (B*L x 64), keys / values of the cross attention
  const float* Win2; const float* bin2;            // decoder: enc_attn in-projection (3 x 64 x 64: q from a1, k / v from f)
  const float* Wo2; const float* bo2;              // decoder: enc_attn out_proj
  const float* W1; const float* b1; const float* W2; const float* b2;    // point-wise feed-forward
  // outputs and tensors saved for the backward (fp32, B*L rows unless noted; nullptr = not stored)
  float* x_out;                        // the gathered input (layer 0)
  float* xn;                           // LayerNorm output (encoder: Q~ ; decoder: D)
  float* qkv;                          // (B*L x 192) packed q, k, v
  float* o; float* lse; uint32_t* mask;            // attention output, (B*H*L) log-sum-exp, (B*H*L x 8) dropout keep bits
  float* h;                            // encoder: Q~ + out_proj(o) ; decoder: a2 = enc_attn.out_proj(o2)
  float* u;                            // relu(dropout1(conv1 .))
  float* y;                            // layer output
  float* rec;                          // encoder: (L*B, H, H) head-classifier log-probabilities, reference row order
  float* a1; float* q2; float* kv2;    // decoder: slf_attn.out_proj(o), cross query (B*L x 64), cross keys / values (B*L x 128)
  float* o2; float* lse2; uint32_t* mask2;         // decoder: cross attention
  float y_scale; int y_acc;            // layer output: y = (y_acc ? y : 0) + y_scale * layer(x)  (y_scale == 0 means 1: plain store) -- the supernet's
                                       // candidate mixing as an epilogue (sasrec/super_modules.py:42-49)
  int saved_bf16;                      // o, h, u, a1, q2, kv2, o2 are written as bf16 rows (first half of the same buffers); see adt_seq_lean
  const float* wp_base; const void* wp_img;        // pre-packed bf16 weight images (adt_wave.cuh: WPack); wp_img == nullptr: none
  unsigned long long* stamps;          // timing experiments only (ADT_SEQ_STAMPS): s_memtime per wave of workgroup 0 at phase ends
  int ablate;                          // timing experiments only (ADT_SEQ_ABLATE): 1 no saved-tensor stores, 2 no attention, 4 no ffn
  // tail of the LAST encoder layer (optional): log_feats = last_layernorm(y) -> f_out, formed on the output tile while it is in registers
  // (sasrec/model.py:48).  No loads: a global load this late in the kernel waits for every store the wave has issued before it (vmcnt is in order).
  const float* lnl_gamma; const float* lnl_beta; float* f_out;
  // Several workgroups per sequence (small batches: fewer sequences than CUs).  nsplit = S > 1: the grid is B * S workgroups, workgroup
  // (b, part) = (blockIdx / S, blockIdx % S) computes the key / value rows of EVERY tile (each workgroup needs them all: recomputed, not
  // exchanged) but queries, attention, out_proj, feed-forward and every store only for the tiles t with t % S == part.  0 / 1: one workgroup.
  int nsplit;
};

}  // namespace adt
