// Step descriptors of the fused forward chain kernel (adt_chain.cuh); shared with the host-side executor.
#pragma once
#include "adt_common.cuh"

namespace adt {

constexpr int CH_THREADS = 512;
constexpr int CH_RS = 68;
constexpr int CH_TILE = 64 * CH_RS;
constexpr int CH_NBUF = 3;
constexpr int CH_MAXSTEPS = 12;

enum { ST_END = 0, ST_LOAD, ST_GATHER, ST_LN, ST_GEMM, ST_CLS, ST_LOGITS };
enum { F_DROP = 1, F_RELU = 2, F_MASK = 4 };

struct ChainStep {
  int op;
  int src, dst;               // tile buffer indices
  int flags;
  const float* W;             // GEMM: 64 x 64 weight block (row-major, ld 64); LN: gamma
  const float* b;             // GEMM: bias (64) ; LN: beta
  const float* in_g; int ld_in;     // LOAD: source ; GEMM: optional global residual (added after relu)
  float* out_g; int ld_out;         // optional global store of the dst tile
  int add_buf;                // GEMM: optional LDS residual tile (-1 none)
  uint32_t site;              // dropout site
};

struct ChainArgs {
  ChainStep steps[CH_MAXSTEPS];
  int T, L, B;
  const int* ids;             // row mask / gather ids
  DropCfg drop;               // thr/scale/seed shared by all steps; site per step
  uint32_t row_offset;
  float ln_eps;
  // GATHER
  const float* E; const float* P; float emb_scale;
  // CLS (head classifier on the src tile)
  const float* Ws; const float* bs; int H; float* rec;
  // LOGITS (on the src tile)
  const int* pos; const int* neg; float* pos_logits; float* neg_logits;
};

}  // namespace adt
