// Argument block of the fused backward row-chain kernels (adt_bwdchain.cuh); shared with the host executor.
#pragma once
#include "adt_common.cuh"

namespace adt {

struct BwdChainArgs {
  int T, L, B;
  const int* ids;
  DropCfg drop; uint32_t site1, site2; uint32_t row_offset;   // site1 = ffn1 (after conv1), site2 = ffn2
  float ln_eps;
  // activations (T x 64 unless noted)
  const float* gy;            // upstream gradient of the layer output (pre-mask)
  const float* u;             // relu(drop1(conv1 .))
  const float* xin;           // enc_post: h (LN2 input); dec_post: a2; *_pre: layer input x; dec_mid: a1
  const float* o;             // attention core output feeding the out_proj of this chain (o / o2 / o1)
  const float* dqkv; int lddqkv;   // *_pre: packed (T x 192) gradient of q,k,v ; dec_mid: dq2 (ld 64)
  const float* dkv2;          // dec_mid: (T x 128) gradient of cross k,v
  int grad_bf16;              // k_seqtt_mid_bwd: dqkv (ld 64) and dkv2 (ld 128) are bf16 rows in the saved-row order (AttnArgs::out_bf16)
  const float* dh;            // enc_pre: gradient wrt LN1 output from the residual path
  const float* f;             // dec_mid: log_feats
  // weights (64 x 64 blocks, row-major)
  const float* W0; const float* W1; const float* W2; const float* W3;
  const float* gamma; const float* beta;
  // outputs
  float* out0; int acc0;      // enc_post: dh ; enc_pre/dec_pre: gx (acc0: +=) ; dec_post: do2 ; dec_mid: do1
  float* out1; int acc1;      // enc_post: dO ; dec_mid: gf (+=)
  // gradient accumulators (global, atomics)
  float* dW0; float* dW1; float* dW2; float* dW3;
  float* db0; float* db1; float* db2; float* db3;
  float* dgamma; float* dbeta;
  // head classifier reverse fused into enc_post (sasrec/modules.py:648-649): rec/drec in reference row order (l*B+b)
  const float* rec; const float* drec; const float* Ws; float* dWs; float* dbs; int H;
  int ablate;                 // timing experiments only (ADT_BWD_ABLATE): 1 skip weight-gradient products, 2 skip flush
  // Parameter-gradient replicas: with nrep > 1 the accumulator pointers above address replica 0 of a zeroed replica area and
  // workgroup b flushes into replica b % nrep (rep_stride floats apart).  256 workgroups adding into the same 64 x 64 block is a
  // 256-deep chain of same-address float atomics (10.5 us per million adds measured, 5.7 us with 8 replicas, 4.5 us private);
  // the executor sums the replicas into the gradient buffer afterwards (adt_replica_reduce).
  int nrep; size_t rep_stride;
  float gy_scale;             // adt_seqpost_tt.cuh: the upstream gradient is gy * gy_scale (0 means 1): the supernet's mixing weight of this candidate
  float* part[4]; size_t part_stride;   // adt_seqpost_tt.cuh: non-null part[k] = private partials of dWk (4096 floats at part[k] + workgroup * part_stride)
  int saved_bf16;             // u / xin / o were saved as bf16 rows by the transposed-chain forward (post and mid chains)
  unsigned long long* stamps;  // timing experiments only (ADT_SEQ_STAMPS=3): s_memtime per wave of workgroup 0 (adt_seqpost_tt.cuh)
  const float* wp_base; const void* wp_img;      // pre-packed bf16 weight images (adt_wave.cuh: WPack); wp_img == nullptr: none
  float* vpart;               // adt_seqpost_tt.cuh: non-null: this workgroup's bias / LayerNorm / classifier gradient sums are STORED at
                              // vpart + blockIdx * 512 (the kernel's sRed layout) instead of added to the replicas with float atomics; they
                              // are summed over the workgroups in order by the optimizer's fold (k_fold_parts_gradnorm, job 3)
  const float* lnl_x; const float* lnl_gamma; float lnl_eps; float* vpart2; float* lnl_dgamma; float* lnl_dbeta;
                              // k_seqtt_post_bwd<ENC>: lnl_x != nullptr: gy is the gradient of the model's LAST LayerNorm's output; its input
                              // rows lnl_x and weight lnl_gamma: the LayerNorm is reversed per tile in front of the chain, its dgamma | dbeta
                              // sums of this workgroup are stored at vpart2 + blockIdx * 512 (128 floats), or (vpart2 == nullptr) added to
                              // lnl_dgamma / lnl_dbeta (replicas like dgamma / dbeta) with float atomics
  int nsplit;                 // adt_seqpost_tt.cuh: S > 1 workgroups per sequence (adt_seq_args.h): workgroup (blockIdx / S, blockIdx % S) runs the token
                              // chains of the tiles t with t % S == part; its weight-gradient partial (slot blockIdx) covers those tokens only
};

}  // namespace adt
