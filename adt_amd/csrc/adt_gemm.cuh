// General dense layers on MFMA for the wide configurations (d = 256, inner = 1024, all-item logits): the
// Linear / Conv1d(k=1) forward, its input gradient and its weight gradient as three tiled GEMM kernels sharing
// one core.  Replaces torch.nn.Linear at bert4rec/model/modules.py:59-75,128-139 (q/k/v/out transfers, FFN),
// bert4rec/model/bert.py:48-51,80-90 (mask_trans_feat, all-item logits), stosa/modules.py:199-212,477-481
// (mean/cov projections, DistIntermediate) and the d = 256 template of sasrec/modules.py:84-137,618-633.
//
// Tiling: 128 x BN (128 | 64) output tile per 256-thread workgroup, BK = 32 contraction step, both operands in LDS
// as fp32 rows of 32 k-values (stride 36), next k-step's global loads issued before the current MFMAs.  Operands
// whose contraction index is the slow global index are transposed while they are written to LDS, so the MFMA
// loop is the same for all three products.  mma16<PREC>: exact-fp32 (v_mfma_f32_16x16x4_f32) or bf16 operands.
#pragma once
#include "adt_common.cuh"

namespace adt {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_ELU = 3, ACT_ELU1 = 4 };

ADT_DEVICE_INLINE float act_apply(int act, float u) {
  switch (act) {
    case ACT_RELU: return fmaxf(u, 0.f);
    case ACT_GELU: return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));   // nn.GELU() (erf form)
    case ACT_ELU: return u > 0.f ? u : expf(u) - 1.0f;                              // nn.ELU()
    case ACT_ELU1: return (u > 0.f ? u : expf(u) - 1.0f) + 1.0f;                    // ELU(x) + 1 (stosa covariances)
    default: return u;
  }
}
ADT_DEVICE_INLINE float act_grad(int act, float u) {
  switch (act) {
    case ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case ACT_GELU: return 0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
    case ACT_ELU:
    case ACT_ELU1: return u > 0.f ? 1.f : expf(u);
    default: return 1.f;
  }
}

constexpr int GBM = 128, GBK = 32, GRS = GBK + 4, GTH = 256;

// ---- operand sources: value(row, col..col+3) as a float4 of the logical ROW-MAJOR matrix the GEMM reads -----
struct PlainSrc {
  const float* p; int ld; int rows, cols;   // logical bounds
  ADT_DEVICE_INLINE float4 at(int r, int c) const {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows && c < cols) {
      const float* q = p + (size_t)r * ld + c;
      if (c + 3 < cols) v = *reinterpret_cast<const float4*>(q);
      else { v.x = q[0]; if (c + 1 < cols) v.y = q[1]; if (c + 2 < cols) v.z = q[2]; }
    }
    return v;
  }
};

// G[t][n] = dY[t][n] * (ids[t] != 0) * dropmask((t + row_offset) * N + n) * act'(U[t][n]) -- the upstream gradient
// of a dense layer pulled back through its epilogue (mask, dropout, activation), formed while the tile is loaded.
struct GradSrc {
  const float* dY; int lddy; int T, N;
  const float* U; int ldu; int act;
  DropCfg drop; uint32_t key; uint32_t row_offset;
  const int* ids;
  ADT_DEVICE_INLINE float4 at(int t, int n) const {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < T && n < N && !(ids && ids[t] == 0)) {
      const float* q = dY + (size_t)t * lddy + n;
      const int lim = N - n < 4 ? N - n : 4;
      if (lim == 4) *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(q);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < lim) v[j] = q[j];
      }
      if (drop.thr) {
        const uint32_t base = (uint32_t)(t + row_offset) * (uint32_t)N + (uint32_t)n;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = adt_keep(key, base + j, drop.thr) ? v[j] * drop.scale : 0.f;
      }
      if (act != ACT_NONE) {
        const float* u = U + (size_t)t * ldu + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < lim) v[j] *= act_grad(act, u[j]);
      }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

// ---- tile movers --------------------------------------------------------------------------------------------
// LDS tile: ROWS rows (output index) x 32 k.  Direct: tile[i][j] = src(r0 + i, k0 + j).  Transposed:
// tile[i][j] = src(k0 + j, r0 + i) (the contraction index is the source's row index).
template <int ROWS, bool TRANS, class Src>
ADT_DEVICE_INLINE void tile_fetch(float4 (&reg)[ROWS * 8 / GTH], const Src& s, int r0, int k0) {
#pragma unroll
  for (int it = 0; it < ROWS * 8 / GTH; ++it) {
    const int f = threadIdx.x + it * GTH;
    if constexpr (!TRANS) {
      const int i = f >> 3, j4 = (f & 7) * 4;
      reg[it] = s.at(r0 + i, k0 + j4);
    } else {
      const int j = f / (ROWS / 4), i4 = (f % (ROWS / 4)) * 4;
      reg[it] = s.at(k0 + j, r0 + i4);
    }
  }
}
template <int ROWS, bool TRANS>
ADT_DEVICE_INLINE void tile_commit(float* tile, const float4 (&reg)[ROWS * 8 / GTH]) {
#pragma unroll
  for (int it = 0; it < ROWS * 8 / GTH; ++it) {
    const int f = threadIdx.x + it * GTH;
    if constexpr (!TRANS) {
      const int i = f >> 3, j4 = (f & 7) * 4;
      *reinterpret_cast<float4*>(tile + i * GRS + j4) = reg[it];
    } else {
      const int j = f / (ROWS / 4), i4 = (f % (ROWS / 4)) * 4;
      tile[(i4 + 0) * GRS + j] = reg[it].x;
      tile[(i4 + 1) * GRS + j] = reg[it].y;
      tile[(i4 + 2) * GRS + j] = reg[it].z;
      tile[(i4 + 3) * GRS + j] = reg[it].w;
    }
  }
}

template <int BN> struct GemmShape {
  static constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN;     // wave grid
  static constexpr int TM = GBM / WM / 16, TN = BN / WN / 16;   // 16x16 tiles per wave: 4x4 or 2x4
};

// C tile (GBM x BN) += A (GBM x kdim) * B^T (BN x kdim); sums k over [k_begin, k_end).  Optional per-row sums of
// the A tile (rowsum: one float per thread < GBM) for the bias gradient.
template <int PREC, int BN, bool TA, bool TB, class SrcA, class SrcB, bool ROWSUM>
ADT_DEVICE_INLINE void gemm_core(f32x4 (&acc)[GemmShape<BN>::TM][GemmShape<BN>::TN], const SrcA& A, const SrcB& Bs, int m0, int n0,
                                 int k_begin, int k_end, float* sA, float* sB, float& rowsum) {
  using S = GemmShape<BN>;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int wm = w / S::WN, wn = w % S::WN;
  float4 ra[GBM * 8 / GTH], rb[BN * 8 / GTH];
  tile_fetch<GBM, TA>(ra, A, m0, k_begin);
  tile_fetch<BN, TB>(rb, Bs, n0, k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
    __syncthreads();   // previous step's readers done
    tile_commit<GBM, TA>(sA, ra);
    tile_commit<BN, TB>(sB, rb);
    __syncthreads();
    if (k0 + GBK < k_end) {
      tile_fetch<GBM, TA>(ra, A, m0, k0 + GBK);
      tile_fetch<BN, TB>(rb, Bs, n0, k0 + GBK);
    }
    if constexpr (ROWSUM) {
      if (threadIdx.x < GBM) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < GBK; j += 4) {
          const float4 v = *reinterpret_cast<const float4*>(sA + threadIdx.x * GRS + j);
          s += (v.x + v.y) + (v.z + v.w);
        }
        rowsum += s;
      }
    }
    Frag8 fa[S::TM], fb[S::TN];
#pragma unroll
    for (int i = 0; i < S::TM; ++i) fa[i] = frag_contig(sA + ((wm * S::TM + i) * 16 + c) * GRS + 8 * g);
#pragma unroll
    for (int j = 0; j < S::TN; ++j) fb[j] = frag_contig(sB + ((wn * S::TN + j) * 16 + c) * GRS + 8 * g);
#pragma unroll
    for (int i = 0; i < S::TM; ++i)
#pragma unroll
      for (int j = 0; j < S::TN; ++j) acc[i][j] = mma16<PREC>(acc[i][j], fa[i], fb[j]);
  }
}

// ---- forward: Y = mask(R + dropout(act(X W^T + b))) ----------------------------------------------------------
struct DenseFwdArgs {
  const float* X; int ldx;
  const float* W; int ldw; const float* b;
  int T, K, N;
  float* Y; int ldy;
  float* U; int ldu;          // optional: pre-activation X W^T + b, saved for the backward of gelu / elu / relu
  int act;
  DropCfg drop; uint32_t row_offset;   // idx = (row + row_offset) * N + col
  const float* R; int ldr;    // optional residual
  const float* R2; int ldr2;  // optional second residual (sasrec decoder: dec_input + (x + ffn(x)), modules.py:672-673)
  const int* ids;             // optional row mask
  const int* t_dev;           // optional DEVICE row count: only rows < min(T, *t_dev) are computed (masked-row batches
                              // whose size changes per step under a captured graph)
};

template <int PREC, int BN>
__global__ __launch_bounds__(GTH) void k_dense_fwd(DenseFwdArgs a) {
  using S = GemmShape<BN>;
  __shared__ __attribute__((aligned(16))) float sA[GBM * GRS];
  __shared__ __attribute__((aligned(16))) float sB[BN * GRS];
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * GBM;
  if (a.t_dev && a.T > *a.t_dev) a.T = *a.t_dev;
  if (m0 >= a.T) return;
  f32x4 acc[S::TM][S::TN] = {};
  const PlainSrc A{a.X, a.ldx, a.T, a.K};
  const PlainSrc Bw{a.W, a.ldw, a.N, a.K};
  float dummy = 0.f;
  gemm_core<PREC, BN, false, false, PlainSrc, PlainSrc, false>(acc, A, Bw, m0, n0, 0, a.K, sA, sB, dummy);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int wm = w / S::WN, wn = w % S::WN;
  const uint32_t key = drop_key(a.drop);
#pragma unroll
  for (int j = 0; j < S::TN; ++j) {
    const int col = n0 + (wn * S::TN + j) * 16 + c;
    if (col >= a.N) continue;
    const float bias = a.b ? a.b[col] : 0.f;
#pragma unroll
    for (int i = 0; i < S::TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * S::TM + i) * 16 + 4 * g + r;
        if (row >= a.T) continue;
        float v = acc[i][j][r] + bias;
        if (a.U) a.U[(size_t)row * a.ldu + col] = v;
        v = act_apply(a.act, v);
        if (a.drop.thr) v = adt_keep(key, (uint32_t)(row + a.row_offset) * (uint32_t)a.N + (uint32_t)col, a.drop.thr) ? v * a.drop.scale : 0.f;
        if (a.R) v += a.R[(size_t)row * a.ldr + col];
        if (a.R2) v += a.R2[(size_t)row * a.ldr2 + col];
        if (a.ids && a.ids[row] == 0) v = 0.f;
        a.Y[(size_t)row * a.ldy + col] = v;
      }
  }
}

// ---- input gradient: dX = (beta ? dX : 0) + G W ----------------------------------------------------------------
struct DenseBwdArgs {
  GradSrc G;                  // T x N
  const float* X; int ldx;    // T x K (weight gradient only)
  const float* W; int ldw;    // N x K
  int K;
  float* dX; int lddx; int beta;
  float* dW; int lddw; float* db;   // accumulated with atomics
  int t_chunk;                // rows of T per blockIdx.z (weight gradient)
  const int* t_dev;           // optional DEVICE row count (see DenseFwdArgs)
};

template <int PREC, int BN>
__global__ __launch_bounds__(GTH) void k_dense_bwd_dx(DenseBwdArgs a) {
  using S = GemmShape<BN>;
  __shared__ __attribute__((aligned(16))) float sA[GBM * GRS];
  __shared__ __attribute__((aligned(16))) float sB[BN * GRS];
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * GBM;   // n0 indexes K (the columns of dX)
  f32x4 acc[S::TM][S::TN] = {};
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  if (m0 >= G.T) return;
  G.key = drop_key(G.drop);
  const PlainSrc Bw{a.W, a.ldw, G.N, a.K};   // read transposed: tile[kcol][n] = W[n][kcol]
  float dummy = 0.f;
  const int kend = (G.N + GBK - 1) / GBK * GBK;
  gemm_core<PREC, BN, false, true, GradSrc, PlainSrc, false>(acc, G, Bw, m0, n0, 0, kend, sA, sB, dummy);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int wm = w / S::WN, wn = w % S::WN;
#pragma unroll
  for (int j = 0; j < S::TN; ++j) {
    const int col = n0 + (wn * S::TN + j) * 16 + c;
    if (col >= a.K) continue;
#pragma unroll
    for (int i = 0; i < S::TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * S::TM + i) * 16 + 4 * g + r;
        if (row >= G.T) continue;
        float* dst = a.dX + (size_t)row * a.lddx + col;
        *dst = (a.beta ? *dst : 0.f) + acc[i][j][r];
      }
  }
}

// ---- weight gradient: dW += G^T X, db += colsum(G); T split over blockIdx.z, partials added with atomics ------
template <int PREC, int BN>
__global__ __launch_bounds__(GTH) void k_dense_bwd_dw(DenseBwdArgs a) {
  using S = GemmShape<BN>;
  __shared__ __attribute__((aligned(16))) float sA[GBM * GRS];
  __shared__ __attribute__((aligned(16))) float sB[BN * GRS];
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * GBM;   // m0 indexes N (rows of dW), n0 indexes K (its columns)
  f32x4 acc[S::TM][S::TN] = {};
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const PlainSrc Xs{a.X, a.ldx, G.T, a.K};
  const int t0 = blockIdx.z * a.t_chunk;
  int t1 = t0 + a.t_chunk;
  if (t1 > G.T) t1 = G.T;
  if (t0 >= t1) return;
  const int kend = t0 + (t1 - t0 + GBK - 1) / GBK * GBK;
  // rows beyond t1 inside the last 32-step belong to the next chunk: bound both sources at t1
  GradSrc Gc = G; Gc.T = t1;
  PlainSrc Xc = Xs; Xc.rows = t1;
  float rowsum = 0.f;
  if (a.db && blockIdx.x == 0) gemm_core<PREC, BN, true, true, GradSrc, PlainSrc, true>(acc, Gc, Xc, m0, n0, t0, kend, sA, sB, rowsum);
  else gemm_core<PREC, BN, true, true, GradSrc, PlainSrc, false>(acc, Gc, Xc, m0, n0, t0, kend, sA, sB, rowsum);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int wm = w / S::WN, wn = w % S::WN;
#pragma unroll
  for (int j = 0; j < S::TN; ++j) {
    const int col = n0 + (wn * S::TN + j) * 16 + c;
    if (col >= a.K) continue;
#pragma unroll
    for (int i = 0; i < S::TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * S::TM + i) * 16 + 4 * g + r;
        if (row >= G.N) continue;
        atomicAdd(a.dW + (size_t)row * a.lddw + col, acc[i][j][r]);
      }
  }
  if (a.db && blockIdx.x == 0 && threadIdx.x < GBM && m0 + (int)threadIdx.x < G.N) atomicAdd(a.db + m0 + threadIdx.x, rowsum);
}

}  // namespace adt
