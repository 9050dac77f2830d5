// Token-parallel ("row") kernels: Linear forward/backward on 64-row tiles with MFMA, LayerNorm
// forward/backward.  Replaces the ATen sequences behind sasrec/modules.py:84-137 (_in_projection_packed),
// :519 (out_proj), :618-633 (PointWiseFeedForward), torch.nn.LayerNorm (:638,640,660; model.py:28).
#pragma once
#include "adt_common.cuh"

namespace adt {

constexpr int BM = 64;        // rows (tokens) per tile
constexpr int NTHREADS = 256; // 4 waves, wave w owns rows 16w..16w+15 of the tile

// ---------------------------------------------------------------------------------------------
// tile movers (all 256 threads; K = row length, multiple of 4; LDS row stride RS)
template <int K, int RS>
ADT_DEVICE_INLINE void load_rows(float* s, const float* g, int ld, int row0, int T) {
  constexpr int V = K / 4;
  for (int i = threadIdx.x; i < BM * V; i += NTHREADS) {
    const int r = i / V, c4 = i % V;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < T) v = *reinterpret_cast<const float4*>(g + (size_t)(row0 + r) * ld + 4 * c4);
    *reinterpret_cast<float4*>(s + r * RS + 4 * c4) = v;
  }
}

// weight rows [n0, n0+nrows) of W (N x K row-major) -> s[n][k]; rows beyond nrows zero-filled up to 64
template <int K, int RS>
ADT_DEVICE_INLINE void stage_w(float* s, const float* W, int n0, int nrows) {
  constexpr int V = K / 4;
  for (int i = threadIdx.x; i < 64 * V; i += NTHREADS) {
    const int r = i / V, c4 = i % V;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows) v = *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * K + 4 * c4);
    *reinterpret_cast<float4*>(s + r * RS + 4 * c4) = v;
  }
}

// C[16 rows of this wave][NT*16] += A[rows][K] * B[n][K]^T ; A, B in LDS with strides RSA, RSB
template <int PREC, int K, int NT, int RSA, int RSB>
ADT_DEVICE_INLINE void gemm_rows(f32x4 (&acc)[NT], const float* sA, const float* sB, int nt_valid) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int kb = 0; kb < K / 32; ++kb) {
    const Frag8 a = frag_contig(sA + (16 * w + c) * RSA + kb * 32 + 8 * g);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt < nt_valid) {
        const Frag8 b = frag_contig(sB + (16 * nt + c) * RSB + kb * 32 + 8 * g);
        acc[nt] = mma16<PREC>(acc[nt], a, b);
      }
    }
  }
}

// acc (C layout) -> LDS tile s[row][col], stride RS
template <int NT, int RS>
ADT_DEVICE_INLINE void acc_to_lds(const f32x4 (&acc)[NT], float* s) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[(16 * w + 4 * g + r) * RS + 16 * nt + c] = acc[nt][r];
}

// ---------------------------------------------------------------------------------------------
struct LinFwdArgs {
  const float* X; int ldx;         // T x K
  const float* W; const float* b;  // N x K row-major, bias N (may be null)
  int N;                           // multiple of 16
  float* Y; int ldy;               // T x N
  int T;
  DropCfg drop;                    // applied to (acc + b); idx = (row + row_offset) * N + col
  uint32_t row_offset;
  int relu;                        // after dropout (FFN1: relu(dropout1(conv1 x)), sasrec/modules.py:629)
  const float* R1; int ldr1;       // residual adds after relu
  const float* R2; int ldr2;
  const int* ids;                  // optional row mask (ids[row] != 0), applied last
};

template <int PREC, int K>
__global__ __launch_bounds__(NTHREADS) void k_linear_fwd(LinFwdArgs a) {
  constexpr int RS = K + 4;
  __shared__ __attribute__((aligned(16))) float sX[BM * RS];
  __shared__ __attribute__((aligned(16))) float sW[64 * RS];
  __shared__ __attribute__((aligned(16))) float sC[BM * 68];
  const int ntiles = (a.T + BM - 1) / BM;
  const int nchunks = (a.N + 63) / 64;
  const uint32_t key = drop_key(a.drop);
  if (nchunks == 1) {
    stage_w<K, RS>(sW, a.W, 0, a.N);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * BM;
    __syncthreads();  // previous tile's readers of sX / sC done
    load_rows<K, RS>(sX, a.X, a.ldx, row0, a.T);
    for (int ch = 0; ch < nchunks; ++ch) {
      const int n0 = ch * 64;
      const int ncols = min(64, a.N - n0);
      if (nchunks > 1) {
        __syncthreads();
        stage_w<K, RS>(sW, a.W, n0, ncols);
      }
      __syncthreads();
      f32x4 acc[4] = {};
      gemm_rows<PREC, K, 4, RS, RS>(acc, sX, sW, (ncols + 15) / 16);
      acc_to_lds<4, 68>(acc, sC);
      __syncthreads();
      // epilogue + coalesced store: thread -> (row, 4 cols)
      for (int i = threadIdx.x; i < BM * 16; i += NTHREADS) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        const int row = row0 + r;
        if (row >= a.T || c4 >= ncols) continue;
        float v[4];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(sC + r * 68 + c4);
        const int col = n0 + c4;
        if (a.b) {
          const float4 bb = *reinterpret_cast<const float4*>(a.b + col);
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        if (a.drop.thr) {
          const uint32_t base = (uint32_t)(row + a.row_offset) * (uint32_t)a.N + (uint32_t)col;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = adt_keep(key, base + j, a.drop.thr) ? v[j] * a.drop.scale : 0.f;
        }
        if (a.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (a.R1) {
          const float4 rr = *reinterpret_cast<const float4*>(a.R1 + (size_t)row * a.ldr1 + col);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (a.R2) {
          const float4 rr = *reinterpret_cast<const float4*>(a.R2 + (size_t)row * a.ldr2 + col);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (a.ids && a.ids[row] == 0) { v[0] = v[1] = v[2] = v[3] = 0.f; }
        *reinterpret_cast<float4*>(a.Y + (size_t)row * a.ldy + col) = *reinterpret_cast<float4*>(v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
struct LinBwdArgs {
  const float* dY; int lddy;       // T x N upstream gradient
  const float* X; int ldx;         // T x K forward input
  const float* W;                  // N x K
  int N;                           // multiple of 16, <= 64 * NCH
  int T;
  // prologue on dY (in this order): row mask, dropout mask, relu mask (U > 0)
  const int* ids;
  DropCfg drop; uint32_t row_offset;
  const float* U; int ldu;
  // outputs
  float* dX; int lddx; int beta;   // dX = (beta ? dX : 0) + dYp W (+ Radd * radd_mask); null = skip
  const float* Radd; int ldradd; const int* radd_ids;
  float* dW;                       // N x K, accumulated with atomics
  float* db;                       // N, accumulated with atomics (may be null)
};

template <int PREC, int K, int NCH>
__global__ __launch_bounds__(NTHREADS) void k_linear_bwd(LinBwdArgs a) {
  constexpr int RS = K + 4;
  constexpr int KT = K / 16;
  __shared__ __attribute__((aligned(16))) float sDY[BM * 68];
  __shared__ __attribute__((aligned(16))) float sX[BM * RS];
  __shared__ __attribute__((aligned(16))) float sW[64 * RS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int ntiles = (a.T + BM - 1) / BM;
  const uint32_t key = drop_key(a.drop);
  f32x4 accW[NCH][KT] = {};   // wave w owns weight rows n0 + 16w .. +15 of every chunk
  float dbacc[NCH] = {};      // threads 0..63 own one bias column per chunk
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * BM;
    __syncthreads();
    load_rows<K, RS>(sX, a.X, a.ldx, row0, a.T);
    f32x4 accX[KT] = {};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int n0 = ch * 64;
      const int ncols = min(64, a.N - n0);
      if (ncols <= 0) break;
      if (ch > 0) __syncthreads();
      // dY chunk with prologue
      for (int i = threadIdx.x; i < BM * 16; i += NTHREADS) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        const int row = row0 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < a.T && c4 < ncols && !(a.ids && a.ids[row] == 0)) {
          const int col = n0 + c4;
          *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(a.dY + (size_t)row * a.lddy + col);
          if (a.drop.thr) {
            const uint32_t base = (uint32_t)(row + a.row_offset) * (uint32_t)a.N + (uint32_t)col;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = adt_keep(key, base + j, a.drop.thr) ? v[j] * a.drop.scale : 0.f;
          }
          if (a.U) {
            const float4 u = *reinterpret_cast<const float4*>(a.U + (size_t)row * a.ldu + col);
            if (!(u.x > 0.f)) v[0] = 0.f;
            if (!(u.y > 0.f)) v[1] = 0.f;
            if (!(u.z > 0.f)) v[2] = 0.f;
            if (!(u.w > 0.f)) v[3] = 0.f;
          }
        }
        *reinterpret_cast<float4*>(sDY + r * 68 + c4) = *reinterpret_cast<float4*>(v);
      }
      if (a.dX) stage_w<K, RS>(sW, a.W, n0, ncols);
      __syncthreads();
      // dX[rows][k] += sum_n dYp[rows][n] * W[n][k]: A = sDY rows in slot order, B = sW columns (strided)
      if (a.dX) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const Frag8 fa = frag_slotc(sDY + (16 * w + c) * 68 + kb * 32, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            const Frag8 fb = frag_strided(sW + (kb * 32) * RS + 16 * kt + c, RS, g);
            accX[kt] = mma16<PREC>(accX[kt], fa, fb);
          }
        }
      }
      // dW[n0 + 16w + c'][k] += sum_r dYp[r][n] X[r][k]  (strided fragments, 2 k-steps of 32 rows)
      if (16 * w < ncols) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const Frag8 fa = frag_strided(sDY + (kb * 32) * 68 + 16 * w + c, 68, g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            const Frag8 fb = frag_strided(sX + (kb * 32) * RS + 16 * kt + c, RS, g);
            accW[ch][kt] = mma16<PREC>(accW[ch][kt], fa, fb);
          }
        }
      }
      if (a.db && threadIdx.x < 64) {
        float s = 0.f;
        for (int r = 0; r < BM; ++r) s += sDY[r * 68 + threadIdx.x];
        dbacc[ch] += s;
      }
    }
    if (a.dX) {
      __syncthreads();
      acc_to_lds<KT, RS>(accX, sX);  // sX no longer needed for this tile
      __syncthreads();
      constexpr int V = K / 4;
      for (int i = threadIdx.x; i < BM * V; i += NTHREADS) {
        const int r = i / V, c4 = (i % V) * 4;
        const int row = row0 + r;
        if (row >= a.T) continue;
        float4 v = *reinterpret_cast<const float4*>(sX + r * RS + c4);
        float* dst = a.dX + (size_t)row * a.lddx + c4;
        if (a.beta) {
          const float4 o = *reinterpret_cast<const float4*>(dst);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (a.Radd && !(a.radd_ids && a.radd_ids[row] == 0)) {
          const float4 o = *reinterpret_cast<const float4*>(a.Radd + (size_t)row * a.ldradd + c4);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(dst) = v;
      }
    }
  }
  // flush weight/bias gradient partials
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int n0 = ch * 64;
    if (n0 + 16 * w < a.N) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 16 * w + 4 * g + r;
          atomicAdd(a.dW + (size_t)n * K + 16 * kt + c, accW[ch][kt][r]);
        }
    }
    if (a.db && threadIdx.x < 64 && n0 + (int)threadIdx.x < a.N) atomicAdd(a.db + n0 + threadIdx.x, dbacc[ch]);
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over K (multiple of 64): 16 lanes per row, each lane K/16 contiguous floats (as float4s).
struct LnArgs {
  const float* X; int ldx;
  const float* gamma; const float* beta;
  float eps;
  float* Y; int ldy;
  int T;
  // backward only
  const float* dY; int lddy;
  float* dX; int lddx; int acc;   // acc: dX += result
  float* dgamma; float* dbeta;    // atomically accumulated
  int nrep; size_t rep_stride;    // nrep > 1: block b adds into replica b % nrep (rep_stride floats apart): shorter same-address chains
  int plain;                      // backward: block b STORES its sums at dgamma / dbeta + b * rep_stride (private slots, summed in order elsewhere)
};

template <int K>
__global__ __launch_bounds__(NTHREADS) void k_ln_fwd(LnArgs a) {
  constexpr int E = K / 16;  // elements per lane
  const int sub = threadIdx.x & 15;
  const int rows_per_block = NTHREADS / 16;
  for (int row = blockIdx.x * rows_per_block + (threadIdx.x >> 4); row < a.T; row += gridDim.x * rows_per_block) {
    float x[E];
#pragma unroll
    for (int e = 0; e < E; e += 4)
      *reinterpret_cast<float4*>(x + e) = *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + (e / 4) * 64 + 4 * sub);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += x[e];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s * (1.0f / K);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { x[e] -= mu; q += x[e] * x[e]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / K) + a.eps);
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const int col = (e / 4) * 64 + 4 * sub;
      const float4 gm = *reinterpret_cast<const float4*>(a.gamma + col);
      const float4 bt = *reinterpret_cast<const float4*>(a.beta + col);
      float4 y;
      y.x = x[e + 0] * rstd * gm.x + bt.x;
      y.y = x[e + 1] * rstd * gm.y + bt.y;
      y.z = x[e + 2] * rstd * gm.z + bt.z;
      y.w = x[e + 3] * rstd * gm.w + bt.w;
      *reinterpret_cast<float4*>(a.Y + (size_t)row * a.ldy + col) = y;
    }
  }
}

template <int K>
__global__ __launch_bounds__(NTHREADS) void k_ln_bwd(LnArgs a) {
  constexpr int E = K / 16;
  __shared__ float sred[2][NTHREADS / 16][K];
  const int sub = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int rows_per_block = NTHREADS / 16;
  float dg[E] = {}, dbt[E] = {};
  float gm[E];
#pragma unroll
  for (int e = 0; e < E; e += 4) *reinterpret_cast<float4*>(gm + e) = *reinterpret_cast<const float4*>(a.gamma + (e / 4) * 64 + 4 * sub);
  // K = 64: TWO rows per 16-lane group and iteration, all four row loads issued before the first reduction (the grid is capped: every block
  // ends with one atomic per column on the same addresses, so memory-level parallelism has to come from inside the wave).  At K = 256 the
  // same change costs registers and was slower (BERT 13.55 -> 13.77 ms per step): one row at a time there.
  constexpr int U = K <= 64 ? 2 : 1;
  const int stride = gridDim.x * rows_per_block;
  for (int row0 = blockIdx.x * rows_per_block + rg; row0 < a.T; row0 += U * stride) {
    float x[U][E], dy[U][E];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = row0 + u * stride;
      ok[u] = row < a.T;
#pragma unroll
      for (int e = 0; e < E; e += 4) {
        const int col = (e / 4) * 64 + 4 * sub;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(x[u] + e) = ok[u] ? *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + col) : z;
        *reinterpret_cast<float4*>(dy[u] + e) = ok[u] ? *reinterpret_cast<const float4*>(a.dY + (size_t)row * a.lddy + col) : z;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;                       // uniform over the 16-lane group; the shuffles below stay inside it
      const int row = row0 + u * stride;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s += x[u][e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mu = s * (1.0f / K);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) { x[u][e] -= mu; q += x[u][e] * x[u][e]; }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      const float rstd = 1.0f / sqrtf(q * (1.0f / K) + a.eps);
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        x[u][e] *= rstd;                 // xhat
        dg[e] += dy[u][e] * x[u][e];
        dbt[e] += dy[u][e];
        dy[u][e] *= gm[e];               // dxhat
        m1 += dy[u][e];
        m2 += dy[u][e] * x[u][e];
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
      m1 *= (1.0f / K);
      m2 *= (1.0f / K);
#pragma unroll
      for (int e = 0; e < E; e += 4) {
        const int col = (e / 4) * 64 + 4 * sub;
        float* dst = a.dX + (size_t)row * a.lddx + col;
        float4 r;
        r.x = rstd * (dy[u][e + 0] - m1 - x[u][e + 0] * m2);
        r.y = rstd * (dy[u][e + 1] - m1 - x[u][e + 1] * m2);
        r.z = rstd * (dy[u][e + 2] - m1 - x[u][e + 2] * m2);
        r.w = rstd * (dy[u][e + 3] - m1 - x[u][e + 3] * m2);
        if (a.acc) {
          const float4 o = *reinterpret_cast<const float4*>(dst);
          r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
        }
        *reinterpret_cast<float4*>(dst) = r;
      }
    }
  }
  // block reduction of dgamma / dbeta over the 16 row groups, then one atomic per column per block
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int col = (e / 4) * 64 + 4 * sub + (e & 3);
    sred[0][rg][col] = dg[e];
    sred[1][rg][col] = dbt[e];
  }
  __syncthreads();
  for (int col = threadIdx.x; col < K; col += NTHREADS) {
    float s0 = 0.f, s1 = 0.f;
    for (int r = 0; r < NTHREADS / 16; ++r) { s0 += sred[0][r][col]; s1 += sred[1][r][col]; }
    if (a.plain) {
      a.dgamma[(size_t)blockIdx.x * a.rep_stride + col] = s0;
      a.dbeta[(size_t)blockIdx.x * a.rep_stride + col] = s1;
      continue;
    }
    const size_t ro = a.nrep > 1 ? (size_t)(blockIdx.x % a.nrep) * a.rep_stride : 0;
    atomicAdd(a.dgamma + ro + col, s0);
    atomicAdd(a.dbeta + ro + col, s1);
  }
}

}  // namespace adt
