// General dense layers on MFMA for the wide configurations (d = 256, inner = 1024, all-item logits): the
// Linear / Conv1d(k=1) forward, its input gradient and its weight gradient as three tiled GEMM kernels sharing
// one core.  Replaces torch.nn.Linear at bert4rec/model/modules.py:59-75,128-139 (q/k/v/out transfers, FFN),
// bert4rec/model/bert.py:48-51,80-90 (mask_trans_feat, all-item logits), stosa/modules.py:199-212,477-481
// (mean/cov projections, DistIntermediate) and the d = 256 template of sasrec/modules.py:84-137,618-633.
//
// Tiling: 128 x BN (128 | 64) output tile per 256-thread workgroup, BK = 32 contraction step, both operands in LDS
// as rows of 32 k-values, next k-step's global loads issued before the current MFMAs.  Operands whose contraction
// index is the slow global index are transposed while they are written to LDS, so the MFMA loop is the same for all
// three products.  bf16 mode: tiles are converted to bf16 once, on the way into LDS (row stride 40 elements), and each
// wave runs 32x32x16 MFMAs on a 64x64 (or 32x64) sub-tile -- one ds_read_b128 per operand fragment, no conversion in
// the loop.  Exact mode: fp32 tiles (stride 36) and v_mfma_f32_16x16x4_f32 through mma16<PREC_F32>.
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

constexpr int GBM = 128, GBK = 32, GTH = 256;   // BK = 64 measured slower at d = 256 (fewer waves per SIMD), faster only at d = 64
constexpr int GPR = GBK / 4;   // float4 per tile row
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gbf16x4 __attribute__((ext_vector_type(4)));

template <int PREC> struct GemmLds;
template <> struct GemmLds<PREC_F32> { typedef float T; static constexpr int RS = GBK + 4; };
template <> struct GemmLds<PREC_BF16> { typedef __bf16 T; static constexpr int RS = GBK + 8; };

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
  int idx_ld, idx_off;          // dropout index = (t + row_offset) * idx_ld + idx_off + n (a column chunk of a wider layer keeps the
                                // layer's own indices); the host sets idx_ld = N, idx_off = 0 for a whole layer
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
        const uint32_t base = (uint32_t)(t + row_offset) * (uint32_t)idx_ld + (uint32_t)(idx_off + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = adt_keep(key, base + j, drop.thr) ? v[j] * drop.scale : 0.f;
      }
      if (act != ACT_NONE) {
        const float* u = U + (size_t)t * ldu + n;
        if (lim == 4 && ((ldu | n) & 3) == 0 && (reinterpret_cast<uintptr_t>(U) & 15) == 0) {      // one 16-byte load instead of four scalar ones
          const float4 uv = *reinterpret_cast<const float4*>(u);
          v[0] *= act_grad(act, uv.x); v[1] *= act_grad(act, uv.y); v[2] *= act_grad(act, uv.z); v[3] *= act_grad(act, uv.w);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (j < lim) v[j] *= act_grad(act, u[j]);
        }
      }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

// ---- tile movers --------------------------------------------------------------------------------------------
// LDS tile: ROWS rows (output index) x GBK k.  Direct: tile[i][j] = src(r0 + i, k0 + j).  Transposed:
// tile[i][j] = src(k0 + j, r0 + i) (the contraction index is the source's row index).
// Interior tiles of a plain matrix (every row and column in range -- all but the edge workgroups / the last k-step): unconditional
// 16-byte loads from one per-thread base pointer, no per-load bounds arithmetic (the generic path spent more vector instructions
// on index checks than on the MFMAs' operands).
template <int ROWS, bool TRANS>
ADT_DEVICE_INLINE bool tile_fetch_fast(float4 (&reg)[ROWS * GPR / GTH], const PlainSrc& s, int r0, int k0) {
  const bool interior = TRANS ? (k0 + GBK <= s.rows && r0 + ROWS <= s.cols) : (r0 + ROWS <= s.rows && k0 + GBK <= s.cols);
  if (!interior) return false;
  if constexpr (!TRANS) {
    const float* base = s.p + (size_t)(r0 + threadIdx.x / GPR) * s.ld + k0 + (threadIdx.x % GPR) * 4;
    const size_t step = (size_t)(GTH / GPR) * s.ld;
#pragma unroll
    for (int it = 0; it < ROWS * GPR / GTH; ++it) reg[it] = *reinterpret_cast<const float4*>(base + it * step);
  } else {
    constexpr int PER = ROWS / 4;                       // float4 per source row
    const float* base = s.p + (size_t)(k0 + threadIdx.x / PER) * s.ld + r0 + (threadIdx.x % PER) * 4;
    const size_t step = (size_t)(GTH / PER) * s.ld;
#pragma unroll
    for (int it = 0; it < ROWS * GPR / GTH; ++it) reg[it] = *reinterpret_cast<const float4*>(base + it * step);
  }
  return true;
}
template <int ROWS, bool TRANS, class Src>
ADT_DEVICE_INLINE void tile_fetch(float4 (&reg)[ROWS * GPR / GTH], const Src& s, int r0, int k0) {
  if constexpr (__is_same(Src, PlainSrc)) {
    if (tile_fetch_fast<ROWS, TRANS>(reg, s, r0, k0)) return;
  }
#pragma unroll
  for (int it = 0; it < ROWS * GPR / GTH; ++it) {
    const int f = threadIdx.x + it * GTH;
    if constexpr (!TRANS) {
      const int i = f / GPR, j4 = (f % GPR) * 4;
      reg[it] = s.at(r0 + i, k0 + j4);
    } else {
      const int j = f / (ROWS / 4), i4 = (f % (ROWS / 4)) * 4;
      reg[it] = s.at(k0 + j, r0 + i4);
    }
  }
}
template <int PREC, int ROWS, bool TRANS>
ADT_DEVICE_INLINE void tile_commit(typename GemmLds<PREC>::T* tile, const float4 (&reg)[ROWS * GPR / GTH]) {
  typedef typename GemmLds<PREC>::T T;
  constexpr int RS = GemmLds<PREC>::RS;
#pragma unroll
  for (int it = 0; it < ROWS * GPR / GTH; ++it) {
    const int f = threadIdx.x + it * GTH;
    if constexpr (!TRANS) {
      const int i = f / GPR, j4 = (f % GPR) * 4;
      if constexpr (PREC == PREC_BF16) {
        gbf16x4 v;
        v[0] = (__bf16)reg[it].x; v[1] = (__bf16)reg[it].y; v[2] = (__bf16)reg[it].z; v[3] = (__bf16)reg[it].w;
        *reinterpret_cast<gbf16x4*>(tile + i * RS + j4) = v;
      } else {
        *reinterpret_cast<float4*>(tile + i * RS + j4) = reg[it];
      }
    } else {
      const int j = f / (ROWS / 4), i4 = (f % (ROWS / 4)) * 4;
      tile[(i4 + 0) * RS + j] = (T)reg[it].x;
      tile[(i4 + 1) * RS + j] = (T)reg[it].y;
      tile[(i4 + 2) * RS + j] = (T)reg[it].z;
      tile[(i4 + 3) * RS + j] = (T)reg[it].w;
    }
  }
}

template <int PREC, int BN> constexpr size_t gemm_lds_bytes() { return (size_t)(GBM + BN) * GemmLds<PREC>::RS * sizeof(typename GemmLds<PREC>::T); }

template <int BN> struct GemmShape {
  static constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN;     // wave grid
  static constexpr int TM = GBM / WM / 16, TN = BN / WN / 16;   // 16x16 tiles per wave: 4x4 or 2x4
};

// Accumulators of one wave's sub-tile and the (row, col) of every element: 16x16 tiles (col = c, row = 4g + r) in the exact
// mode, 32x32 tiles (col = lane % 32, row = 8 (e / 4) + 4 (lane / 32) + e % 4) in the bf16 mode.
template <int PREC, int BN>
struct GemmAcc {
  using S = GemmShape<BN>;
  static constexpr bool B16 = PREC == PREC_BF16;
  static constexpr int TM = B16 ? S::TM / 2 : S::TM, TN = B16 ? S::TN / 2 : S::TN;
  f32x4 a16[B16 ? 1 : TM][B16 ? 1 : TN];
  f32x16 a32[B16 ? TM : 1][B16 ? TN : 1];

  ADT_DEVICE_INLINE void zero() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (B16) {
#pragma unroll
          for (int e = 0; e < 16; ++e) a32[i][j][e] = 0.f;
        } else {
          a16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
  }

  // One call per MFMA tile with its NE (16 or 4) elements: rows[e], the common column and the values.  The epilogues use it to
  // issue all the global loads of a tile (residuals, the old dX) BEFORE the first store: a load placed after a store through
  // a possibly-aliasing pointer cannot be hoisted by the compiler, and one exposed memory latency per element made the
  // epilogue the longest part of these kernels.
  static constexpr int NE = B16 ? 16 : 4;
  template <class F>
  ADT_DEVICE_INLINE void foreach_tile(int m0, int n0, F f) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w / S::WN, wn = w % S::WN;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int rows[NE];
        float vals[NE];
        int col;
        if constexpr (B16) {
          col = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
          for (int e = 0; e < 16; ++e) { rows[e] = m0 + (wm * TM + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3); vals[e] = a32[i][j][e]; }
        } else {
          col = n0 + (wn * TN + j) * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) { rows[r] = m0 + (wm * TM + i) * 16 + 4 * (lane >> 4) + r; vals[r] = a16[i][j][r]; }
        }
        f(rows, col, vals);
      }
  }

  template <class F>
  ADT_DEVICE_INLINE void foreach(int m0, int n0, F f) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w / S::WN, wn = w % S::WN;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (B16) {
          const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
          for (int e = 0; e < 16; ++e) f(m0 + (wm * TM + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3), col, a32[i][j][e]);
        } else {
          const int col = n0 + (wn * TN + j) * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) f(m0 + (wm * TM + i) * 16 + 4 * (lane >> 4) + r, col, a16[i][j][r]);
        }
      }
  }
};

// C tile (GBM x BN) += A (GBM x kdim) * B^T (BN x kdim); sums k over [k_begin, k_end).  Optional per-row sums of
// the A tile (rowsum: one float per thread < GBM) for the bias gradient.
template <int PREC, int BN, bool TA, bool TB, class SrcA, class SrcB, bool ROWSUM>
ADT_DEVICE_INLINE void gemm_core(GemmAcc<PREC, BN>& acc, const SrcA& A, const SrcB& Bs, int m0, int n0, int k_begin, int k_end,
                                 typename GemmLds<PREC>::T* sA, typename GemmLds<PREC>::T* sB, float& rowsum) {
  using S = GemmShape<BN>;
  using AC = GemmAcc<PREC, BN>;
  constexpr int RS = GemmLds<PREC>::RS;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w / S::WN, wn = w % S::WN;
  float4 ra[GBM * GPR / GTH], rb[BN * GPR / GTH];
  tile_fetch<GBM, TA>(ra, A, m0, k_begin);
  tile_fetch<BN, TB>(rb, Bs, n0, k_begin);
  auto compute = [&]() {
    if constexpr (ROWSUM) {
      if (threadIdx.x < GBM) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < GBK; ++j) s += (float)sA[threadIdx.x * RS + j];
        rowsum += s;
      }
    }
    if constexpr (PREC == PREC_BF16) {
      const int r32 = lane & 31, kg = lane >> 5;
#pragma unroll
      for (int ks = 0; ks < GBK / 16; ++ks) {
        bf16x8 fa[AC::TM], fb[AC::TN];
#pragma unroll
        for (int i = 0; i < AC::TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sA + ((wm * AC::TM + i) * 32 + r32) * RS + ks * 16 + 8 * kg);
#pragma unroll
        for (int j = 0; j < AC::TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sB + ((wn * AC::TN + j) * 32 + r32) * RS + ks * 16 + 8 * kg);
#pragma unroll
        for (int i = 0; i < AC::TM; ++i)
#pragma unroll
          for (int j = 0; j < AC::TN; ++j) acc.a32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc.a32[i][j], 0, 0, 0);
      }
    } else {
      const int c = lane & 15, g = lane >> 4;
#pragma unroll
      for (int kb = 0; kb < GBK / 32; ++kb) {
        Frag8 fa[S::TM], fb[S::TN];
#pragma unroll
        for (int i = 0; i < S::TM; ++i) fa[i] = frag_contig(sA + ((wm * S::TM + i) * 16 + c) * RS + kb * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < S::TN; ++j) fb[j] = frag_contig(sB + ((wn * S::TN + j) * 16 + c) * RS + kb * 32 + 8 * g);
#pragma unroll
        for (int i = 0; i < S::TM; ++i)
#pragma unroll
          for (int j = 0; j < S::TN; ++j) acc.a16[i][j] = mma16<PREC_F32>(acc.a16[i][j], fa[i], fb[j]);
      }
    }
  };
  // (a second register stage -- two k-steps of loads in flight -- was measured slower: the extra 32 VGPRs cost a wave per SIMD)
  for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
    __syncthreads();   // previous step's readers done
    tile_commit<PREC, GBM, TA>(sA, ra);
    tile_commit<PREC, BN, TB>(sB, rb);
    __syncthreads();
    if (k0 + GBK < k_end) {
      tile_fetch<GBM, TA>(ra, A, m0, k0 + GBK);
      tile_fetch<BN, TB>(rb, Bs, n0, k0 + GBK);
    }
    compute();
  }
}

// XCD-aware block -> tile mapping.  Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own L2:
// the tiles that re-read the same operand rows (all column tiles of one row tile; all output tiles of one split-k chunk) are
// given ids that are congruent mod 8, so they run on ONE XCD and the shared operand is fetched into one L2 once instead of
// into eight.  grid = xcd_grid(n_outer, n_inner) blocks; (outer, inner) = the tile pair of this block, outer >= n_outer: idle.
ADT_DEVICE_INLINE void xcd_tile(int n_outer, int n_inner, int& outer, int& inner) {
  const int bid = blockIdx.x;
  if (n_outer < 16) {      // too few groups to spread over 8 XCDs: plain row-major mapping (grid = n_outer * n_inner)
    outer = bid / n_inner;
    inner = bid % n_inner;
    return;
  }
  const int xcd = bid & 7, local = bid >> 3;
  outer = (local / n_inner) * 8 + xcd;
  inner = local % n_inner;
}
static inline int xcd_grid(int n_outer, int n_inner) { return n_outer < 16 ? n_outer * n_inner : (n_outer + 7) / 8 * 8 * n_inner; }

// ---- forward: Y = mask(R + R2 + dropout(act(X W^T + b))) -----------------------------------------------------
struct DenseFwdArgs {
  const float* X; int ldx;
  const float* W; int ldw; const float* b;
  int T, K, N;
  float* Y; int ldy;
  float* U; int ldu;          // optional: pre-activation X W^T + b, saved for the backward of gelu / elu / relu
  int act;
  int nt_n, nt_m;             // tile counts along N and T (1-D launch, see xcd_tile)
  DropCfg drop; uint32_t row_offset;   // idx = (row + row_offset) * N + col
  const float* R; int ldr;    // optional residual
  const float* R2; int ldr2;  // optional second residual (sasrec decoder: dec_input + (x + ffn(x)), modules.py:672-673)
  const int* ids;             // optional row mask
  const int* t_dev;           // optional DEVICE row count: only rows < min(T, *t_dev) are computed (masked-row batches
                              // whose size changes per step under a captured graph)
};

template <int PREC, int BN>
__global__ __launch_bounds__(GTH) void k_dense_fwd(DenseFwdArgs a) {
  adt_prefetch_kernargs<sizeof(DenseFwdArgs) <= 512 ? sizeof(DenseFwdArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef typename GemmLds<PREC>::T LT;
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  LT* sA = reinterpret_cast<LT*>(gemm_smem);
  LT* sB = sA + GBM * GemmLds<PREC>::RS;
  int tm, tn;
  xcd_tile(a.nt_m, a.nt_n, tm, tn);
  const int n0 = tn * BN, m0 = tm * GBM;
  if (a.t_dev && a.T > *a.t_dev) a.T = *a.t_dev;
  if (m0 >= a.T) return;
  GemmAcc<PREC, BN> acc;
  acc.zero();
  const PlainSrc A{a.X, a.ldx, a.T, a.K};
  const PlainSrc Bw{a.W, a.ldw, a.N, a.K};
  float dummy = 0.f;
  gemm_core<PREC, BN, false, false, PlainSrc, PlainSrc, false>(acc, A, Bw, m0, n0, 0, a.K, sA, sB, dummy);
  const uint32_t key = drop_key(a.drop);
  constexpr int NE = GemmAcc<PREC, BN>::NE;
  acc.foreach_tile(m0, n0, [&](const int (&rows)[NE], int col, const float (&vals)[NE]) {
    if (col >= a.N) return;
    const float bias = a.b ? a.b[col] : 0.f;
    float r1[NE], r2[NE];
    int keep[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {        // every load of the tile first ...
      const bool ok = rows[e] < a.T;
      r1[e] = (ok && a.R) ? a.R[(size_t)rows[e] * a.ldr + col] : 0.f;
      r2[e] = (ok && a.R2) ? a.R2[(size_t)rows[e] * a.ldr2 + col] : 0.f;
      keep[e] = (ok && a.ids) ? a.ids[rows[e]] : 1;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {        // ... then the arithmetic and the stores
      const int row = rows[e];
      if (row >= a.T) continue;
      float v = vals[e] + bias;
      if (a.U) a.U[(size_t)row * a.ldu + col] = v;
      v = act_apply(a.act, v);
      if (a.drop.thr) v = adt_keep(key, (uint32_t)(row + a.row_offset) * (uint32_t)a.N + (uint32_t)col, a.drop.thr) ? v * a.drop.scale : 0.f;
      v += r1[e] + r2[e];
      if (keep[e] == 0) v = 0.f;
      a.Y[(size_t)row * a.ldy + col] = v;
    }
  });
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
  int nt_a, nt_b, nt_z;       // tile counts of the 1-D launches (see xcd_tile): dx: K tiles, T tiles, N splits; dw: K tiles, N tiles, T splits
  int n_chunk;                // columns of N per blockIdx.z (input gradient; gridDim.z > 1: partials are added with atomics)
  const int* t_dev;           // optional DEVICE row count (see DenseFwdArgs)
};

template <int PREC, int BN>
__global__ __launch_bounds__(GTH) void k_dense_bwd_dx(DenseBwdArgs a) {
  adt_prefetch_kernargs<sizeof(DenseBwdArgs) <= 512 ? sizeof(DenseBwdArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef typename GemmLds<PREC>::T LT;
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  LT* sA = reinterpret_cast<LT*>(gemm_smem);
  LT* sB = sA + GBM * GemmLds<PREC>::RS;
  int tm, inner;
  xcd_tile(a.nt_b, a.nt_a * a.nt_z, tm, inner);                    // all K tiles and N splits of one T tile share its G rows
  const int zz = inner / a.nt_a;
  const int n0 = (inner % a.nt_a) * BN, m0 = tm * GBM;     // n0 indexes K (the columns of dX)
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  if (m0 >= G.T) return;
  GemmAcc<PREC, BN> acc;
  acc.zero();
  G.key = drop_key(G.drop);
  const PlainSrc Bw{a.W, a.ldw, G.N, a.K};   // read transposed: tile[kcol][n] = W[n][kcol]
  float dummy = 0.f;
  const int kall = (G.N + GBK - 1) / GBK * GBK;
  const int kbeg = a.nt_z > 1 ? zz * a.n_chunk : 0;
  const int kend = a.nt_z > 1 ? (kbeg + a.n_chunk < kall ? kbeg + a.n_chunk : kall) : kall;
  if (kbeg >= kend) return;
  gemm_core<PREC, BN, false, true, GradSrc, PlainSrc, false>(acc, G, Bw, m0, n0, kbeg, kend, sA, sB, dummy);
  const bool split = a.nt_z > 1;     // long contraction (all-item logits): the host zeroes dX first unless beta
  constexpr int NE = GemmAcc<PREC, BN>::NE;
  acc.foreach_tile(m0, n0, [&](const int (&rows)[NE], int col, const float (&vals)[NE]) {
    if (col >= a.K) return;
    float old[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) old[e] = (a.beta && !split && rows[e] < G.T) ? a.dX[(size_t)rows[e] * a.lddx + col] : 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (rows[e] >= G.T) continue;
      float* dst = a.dX + (size_t)rows[e] * a.lddx + col;
      if (split) atomicAdd(dst, vals[e]);
      else *dst = old[e] + vals[e];
    }
  });
}

// ---- weight gradient of a 64 x 64 layer (bf16 operands): dW += G^T X, db += colsum(G) -------------------------------------------------
// The tiled kernel below runs a 128 x 64 output tile in 32-row k-steps with a barrier pair per step; on a 64 x 64 layer it is one tile, so the
// only parallelism is the T split, and 400-512 workgroups of two k-steps each spent their time on launch latency and on 4,096 atomics apiece
// (33 us per call, 14 calls per STOSA-ADT step = 21 % of it).  Here a workgroup takes 128 rows per stage: G (with the epilogue's gradient
// applied by GradSrc::at) and X go to two natural-order bf16 images, wave w computes output rows 16 w .. 16 w + 15 (four 16 x 16 tiles,
// the G fragment shared) with both operands read through ds_read_b64_tr_b16, and the chunk count is chosen so that ~200 workgroups flush.
// Layers of up to four 64 x 64 blocks (64 <-> 256 feed-forward layers) run one block per blockIdx.y.
constexpr int DW64_ROWS = 128, DW64_RS = 72, DW64_NTH = 256;
typedef short dw64_s4 __attribute__((ext_vector_type(4)));
typedef __bf16 dw64_b4 __attribute__((ext_vector_type(4)));
ADT_DEVICE_INLINE bf16x8 dw64_trfrag(const __bf16* img, int row0, int col0, int c, int g) {     // feature col0 + c on the lane, 8 rows in slot order
  const __bf16* p = img + (row0 + 4 * g + (c >> 2)) * DW64_RS + col0 + 4 * (c & 3);
  union { struct { dw64_s4 a, b; } s; bf16x8 v; } u;
  u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(p));
  u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(p + 16 * DW64_RS));
  return u.v;
}
// part != nullptr: the workgroup's 64 x 64 product (+ its 64 bias sums) is STORED at part + (blockIdx.y * gridDim.x + blockIdx.x) * 4160 in
// register order and k_dense_dw64_reduce adds the partials up: with ~200 workgroups flushing 4,096 float atomics each onto the same 4,096
// addresses the flush was most of the kernel's 27.5 us at 25,600 tokens (STOSA-ADT runs it twenty times per step).
constexpr int DW64_PART = 4096 + 64;
__global__ __launch_bounds__(DW64_NTH) void k_dense_dw64(DenseBwdArgs a, float* part) {
  __shared__ __attribute__((aligned(16))) __bf16 sG[DW64_ROWS * DW64_RS];
  __shared__ __attribute__((aligned(16))) __bf16 sX[DW64_ROWS * DW64_RS];
  __shared__ float sB[16][64];
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const int t0 = blockIdx.x * a.t_chunk;
  const int t1 = t0 + a.t_chunk < G.T ? t0 + a.t_chunk : G.T;
  if (t0 >= t1) return;
  G.T = t1;                                                // rows of the next chunk read as zeros
  const int kblocks = a.K / 64, n0 = 64 * (blockIdx.y / kblocks), k0 = 64 * (blockIdx.y % kblocks);      // the 64 x 64 block of a wider layer
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int col4 = (threadIdx.x & 15) * 4, rsub = threadIdx.x >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = t0; s0 < t1; s0 += DW64_ROWS) {
    float4 gv[DW64_ROWS / 16], xv[DW64_ROWS / 16];
#pragma unroll
    for (int i = 0; i < DW64_ROWS / 16; ++i) {
      const int row = s0 + rsub + 16 * i;
      xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < t1) xv[i] = *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + k0 + col4);
    }
#pragma unroll
    for (int i = 0; i < DW64_ROWS / 16; ++i) gv[i] = G.at(s0 + rsub + 16 * i, n0 + col4);
#pragma unroll
    for (int i = 0; i < DW64_ROWS / 16; ++i) {
      const int r = rsub + 16 * i;
      bs[0] += gv[i].x; bs[1] += gv[i].y; bs[2] += gv[i].z; bs[3] += gv[i].w;
      *reinterpret_cast<dw64_b4*>(sG + r * DW64_RS + col4) = dw64_b4{(__bf16)gv[i].x, (__bf16)gv[i].y, (__bf16)gv[i].z, (__bf16)gv[i].w};
      *reinterpret_cast<dw64_b4*>(sX + r * DW64_RS + col4) = dw64_b4{(__bf16)xv[i].x, (__bf16)xv[i].y, (__bf16)xv[i].z, (__bf16)xv[i].w};
    }
    __syncthreads();
#pragma unroll
    for (int kp = 0; kp < DW64_ROWS / 32; ++kp) {
      const bf16x8 fg = dw64_trfrag(sG, kp * 32, 16 * w, c, g);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg, dw64_trfrag(sX, kp * 32, 16 * kt, c, g), acc[kt], 0, 0, 0);
    }
    __syncthreads();
  }
  float* const mine = part ? part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * DW64_PART : nullptr;
  if (mine) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) *reinterpret_cast<f32x4*>(mine + ((w * 4 + kt) * 64 + lane) * 4) = acc[kt];      // element ((w, kt, lane), r)
  } else {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(a.dW + (size_t)(n0 + 16 * w + 4 * g + r) * a.lddw + k0 + 16 * kt + c, acc[kt][r]);
  }
  if (a.db && k0 == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sB[rsub][col4 + j] = bs[j];
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += sB[i][threadIdx.x];
      if (mine) mine[4096 + threadIdx.x] = t;
      else atomicAdd(a.db + n0 + threadIdx.x, t);
    }
  }
}
// dW (+ db) += the partials of k_dense_dw64, workgroups in order: block (x, y) takes 64 float4 slots (x < 16) or the bias sums (x == 16) of
// 64 x 64 block y; sixteen waves split the workgroups (wave j: j, j + 16, ..), joined through LDS in wave order.  Workgroups whose token
// chunk was empty (t_dev) left their partial unwritten: nwg_dev = number of chunks that ran.
__global__ __launch_bounds__(1024) void k_dense_dw64_reduce(const float* part, int nwg, const int* t_dev, int T, int t_chunk, float* dW, int lddw,
                                                            int kblocks, float* db) {
  __shared__ float4 sp[16][64];
  int tt = T;
  if (t_dev && tt > *t_dev) tt = *t_dev;
  const int nrun = (tt + t_chunk - 1) / t_chunk < nwg ? (tt + t_chunk - 1) / t_chunk : nwg;
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6, y = blockIdx.y;
  const float* base = part + (size_t)y * nwg * DW64_PART;
  const bool bias = blockIdx.x == 16;
  if (bias && (!db || (y % kblocks) != 0)) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!bias) {
    const int slot = blockIdx.x * 64 + lane;
    for (int z0 = j; z0 < nrun; z0 += 16 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int z = z0 + 16 * u;
        v[u] = z < nrun ? reinterpret_cast<const float4*>(base + (size_t)z * DW64_PART)[slot] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  } else {
    for (int z = j; z < nrun; z += 16) s.x += base[(size_t)z * DW64_PART + 4096 + lane];
  }
  sp[j][lane] = s;
  __syncthreads();
  if (j != 0) return;
  float4 t = sp[0][lane];
#pragma unroll
  for (int k = 1; k < 16; ++k) { t.x += sp[k][lane].x; t.y += sp[k][lane].y; t.z += sp[k][lane].z; t.w += sp[k][lane].w; }
  const int n0 = 64 * (y / kblocks), k0 = 64 * (y % kblocks);
  if (bias) { db[n0 + lane] += t.x; return; }
  const int slot = blockIdx.x * 64 + lane;              // (w, kt, lane'): slot = (w * 4 + kt) * 64 + lane'
  const int lp = slot & 63, kt = (slot >> 6) & 3, w = slot >> 8, c = lp & 15, g = lp >> 4;
  float* dst = dW + (size_t)(n0 + 16 * w + 4 * g) * lddw + k0 + 16 * kt + c;
  dst[0] += t.x; dst[(size_t)lddw] += t.y; dst[2 * (size_t)lddw] += t.z; dst[3 * (size_t)lddw] += t.w;
}

// ---- weight gradient of a 256 x 256 layer through private partials (bf16 operands) -------------------------------------------------------
// At 256 x 256 the tiled kernel (and the 256 x 128-block one) is bound by its flush: every workgroup adds a 128 KB (256 KB) block to dW with
// float atomics.  Here a workgroup owns the WHOLE 256 x 256 product of its token chunk: eight waves x (32 rows x 256 columns) = 128
// accumulator registers per lane, 32-token stages, G and X as two bf16 LDS images in the dual-use layout of adt_lce.cuh (both operands of
// v_mfma_f32_32x32x16_bf16 are read through ds_read_b64_tr_b16: the contraction runs over tokens), the next stage's rows requested before the
// current one is multiplied.  The partial goes to the workspace in register order with plain 16-byte stores; k_dense_dw256_reduce folds the
// partials into dW (32 per block, then one atomic per element: an 8-deep chain instead of a 256-deep one).  Layers of up to four 256 x 256
// blocks (the 256 <-> 1024 feed-forward layers) run one block per blockIdx.y with the workgroups divided among the blocks.
constexpr int DWP_TS = 32, DWP_NTH = 512, DWP_IMG = 32 * 256 * 2;
ADT_DEVICE_INLINE int dwp_off(int row, int ch) {          // byte offset of 16-byte chunk ch of row `row` (adt_lce.cuh: lce_off<256>)
  return 4096 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
__global__ __launch_bounds__(DWP_NTH) void k_dense_dw256(DenseBwdArgs a, float* part) {
  adt_prefetch_kernargs<sizeof(DenseBwdArgs) <= 512 ? sizeof(DenseBwdArgs) : 512>();
  __shared__ __attribute__((aligned(1024))) unsigned char sG[DWP_IMG];
  __shared__ __attribute__((aligned(1024))) unsigned char sX[DWP_IMG];
  __shared__ float sB[8][256];
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const int t0 = blockIdx.x * a.t_chunk;
  const int t1 = t0 + a.t_chunk < G.T ? t0 + a.t_chunk : G.T;
  G.T = t1 > t0 ? t1 : t0;
  const int kblocks = a.K / 256, n0 = 256 * (blockIdx.y / kblocks), k0 = 256 * (blockIdx.y % kblocks);      // the 256 x 256 block of a wider layer
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5;
  const int col4 = (threadIdx.x & 63) * 4, rsub = threadIdx.x >> 6;       // staging: 64 float4 per row, rows rsub + 8 i
  int tb[2];
  {
    const int li = lane & 15, q = li >> 2, p = li & 3, gi = (lane >> 4) & 1;
    tb[0] = 64 * (4 * h + q) + 16 * ((2 * gi + (p >> 1)) ^ h) + 8 * (p & 1);
    tb[1] = 64 * (4 * h + q) + 16 * ((2 * gi + (p >> 1)) ^ (2 + h)) + 8 * (p & 1);
  }
  typedef float f32x16d __attribute__((ext_vector_type(16)));
  f32x16d acc[8];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[kt][e] = 0.f;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  float4 gv[4], xv[4];
  auto request = [&](int s0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = s0 + rsub + 8 * i;
      xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < t1) xv[i] = *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + k0 + col4);
      gv[i] = G.at(row, n0 + col4);
    }
  };
  if (t0 < t1) request(t0);
  for (int s0 = t0; s0 < t1; s0 += DWP_TS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rsub + 8 * i, o = dwp_off(r, col4 >> 3) + 8 * ((col4 >> 2) & 1);
      bs[0] += gv[i].x; bs[1] += gv[i].y; bs[2] += gv[i].z; bs[3] += gv[i].w;
      *reinterpret_cast<dw64_b4*>(sG + o) = dw64_b4{(__bf16)gv[i].x, (__bf16)gv[i].y, (__bf16)gv[i].z, (__bf16)gv[i].w};
      *reinterpret_cast<dw64_b4*>(sX + o) = dw64_b4{(__bf16)xv[i].x, (__bf16)xv[i].y, (__bf16)xv[i].z, (__bf16)xv[i].w};
    }
    __syncthreads();
    if (s0 + DWP_TS < t1) request(s0 + DWP_TS);           // in flight while this stage is multiplied
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      union { struct { dw64_s4 lo, hi; } p; bf16x8 v; } fa;
      fa.p.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(sG + tb[0] + 4096 * (2 * ks) + 512 * w));
      fa.p.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(sG + tb[1] + 4096 * (2 * ks + 1) + 512 * w));
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        union { struct { dw64_s4 lo, hi; } p; bf16x8 v; } fb;
        fb.p.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(sX + tb[0] + 4096 * (2 * ks) + 512 * kt));
        fb.p.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(sX + tb[1] + 4096 * (2 * ks + 1) + 512 * kt));
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc[kt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // partial in register order: float4 slot ((w * 8 + kt) * 4 + i) * 64 + lane of workgroup blockIdx.x
  float4* dp = reinterpret_cast<float4*>(part) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16384 + (size_t)w * 8 * 4 * 64 + lane;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) dp[(kt * 4 + i) * 64] = make_float4(acc[kt][4 * i], acc[kt][4 * i + 1], acc[kt][4 * i + 2], acc[kt][4 * i + 3]);
  if (a.db && k0 == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sB[rsub][col4 + j] = bs[j];
    __syncthreads();
    if (threadIdx.x < 256) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += sB[i][threadIdx.x];
      if (t != 0.f) atomicAdd(a.db + n0 + threadIdx.x, t);
    }
  }
}
// dW[n][k] += sum over the workgroups' partials: block (slice, z) sums partials z * per .. of 256 float4 slots and adds its result with atomics
// (blockIdx.z = the 256 x 256 block of a wider layer: its partials start at part + z * nwg * 65536, its corner of dW is (256 (z / kblocks), 256 (z % kblocks)))
__global__ __launch_bounds__(256) void k_dense_dw256_reduce(const float* part, int nwg, int per, float* dW, int lddw, int kblocks) {
  part += (size_t)blockIdx.z * nwg * 65536;
  dW += (size_t)(256 * (blockIdx.z / kblocks)) * lddw + 256 * (blockIdx.z % kblocks);
  const int slot = blockIdx.x * 256 + threadIdx.x;          // 0 .. 16383: ((w * 8 + kt) * 4 + i) * 64 + lane
  const int z0 = blockIdx.y * per, z1 = z0 + per < nwg ? z0 + per : nwg;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int z = z0;
  for (; z + 8 <= z1; z += 8) {          // eight loads in flight (one per iteration was one round trip per partial: 8-10 us per launch)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(part)[(size_t)(z + u) * 16384 + slot];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; z < z1; ++z) {
    const float4 v = reinterpret_cast<const float4*>(part)[(size_t)z * 16384 + slot];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int lane = slot & 63, i = (slot >> 6) & 3, kt = (slot >> 8) & 7, w = slot >> 11;
  const int n = 32 * w + 8 * i + 4 * (lane >> 5), k = 32 * kt + (lane & 31);
  atomicAdd(dW + (size_t)(n + 0) * lddw + k, s.x);
  atomicAdd(dW + (size_t)(n + 1) * lddw + k, s.y);
  atomicAdd(dW + (size_t)(n + 2) * lddw + k, s.z);
  atomicAdd(dW + (size_t)(n + 3) * lddw + k, s.w);
}

// ---- forward of a layer with K = 256, N a multiple of 256 (bf16 operands): Y = mask(R + R2 + dropout(act(X W^T + b))) -------------------------
// The skeleton of k_dense_dx256 below: wave w of column block blockIdx.y keeps W[n0 + 32 w .. + 31][:] as sixteen B fragments (staged through
// LDS in eight coalesced passes, pass p holds wave p's rows), the activations stream through LDS in 32-token stages, the output tile is
// transposed (n on the lane, token on the accumulator rows) and the epilogue of k_dense_fwd runs on it -- same order of operations, same
// dropout indices.  The epilogue is specialised at compile time (EPI bit 0: residuals / row mask, bit 1: dropout, bit 2: activation / U): the
// generic form held 234 registers and per-element branches and was slower than the row-streaming kernel.  Y, U, R, R2: 16-byte aligned rows.
template <int EPI>
__global__ __launch_bounds__(DWP_NTH) void k_dense_fwd256(DenseFwdArgs a, int t_chunk) {
  adt_prefetch_kernargs<sizeof(DenseFwdArgs) <= 512 ? sizeof(DenseFwdArgs) : 512>();
  __shared__ __attribute__((aligned(1024))) unsigned char sX[DWP_IMG];
  __shared__ __attribute__((aligned(16))) float sT[32 * 260];
  const int T = (a.t_dev && *a.t_dev < a.T) ? *a.t_dev : a.T;
  const int t0 = blockIdx.x * t_chunk;
  const int t1 = t0 + t_chunk < T ? t0 + t_chunk : T;
  if (t0 >= t1) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int col4 = (threadIdx.x & 63) * 4, rsub = threadIdx.x >> 6;
  const int col = 256 * blockIdx.y + 32 * w + r;           // this lane's output column
  int rb[2];
  {
    const int v = (r >> 2) & 3;
    rb[0] = 4096 * (r >> 3) + 64 * (r & 7) + 16 * (h ^ v);
    rb[1] = 4096 * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ v);
  }
  bf16x8 wf[16];
  for (int pass = 0; pass < 8; ++pass) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = rsub + 8 * i;
      const float4 wv = *reinterpret_cast<const float4*>(a.W + (size_t)(256 * blockIdx.y + 32 * pass + rr) * a.ldw + col4);
      *reinterpret_cast<dw64_b4*>(sX + dwp_off(rr, col4 >> 3) + 8 * ((col4 >> 2) & 1)) = dw64_b4{(__bf16)wv.x, (__bf16)wv.y, (__bf16)wv.z, (__bf16)wv.w};
    }
    __syncthreads();
    if (pass == w) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(sX + rb[ks & 1] + 512 * (ks >> 1));
    }
    __syncthreads();
  }
  float4 xv[4];
  auto request = [&](int s0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = s0 + rsub + 8 * i;
      xv[i] = *reinterpret_cast<const float4*>(a.X + (size_t)(row < t1 ? row : t1 - 1) * a.ldx + col4);      // rows past the chunk: never stored
    }
  };
  request(t0);
  const float bias = a.b ? a.b[col] : 0.f;
  const uint32_t key = (EPI & 2) ? drop_key(a.drop) : 0u;
  typedef float f32x16y __attribute__((ext_vector_type(16)));
  for (int s0 = t0; s0 < t1; s0 += DWP_TS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = rsub + 8 * i;
      *reinterpret_cast<dw64_b4*>(sX + dwp_off(rr, col4 >> 3) + 8 * ((col4 >> 2) & 1)) = dw64_b4{(__bf16)xv[i].x, (__bf16)xv[i].y, (__bf16)xv[i].z, (__bf16)xv[i].w};
    }
    __syncthreads();
    if (s0 + DWP_TS < t1) request(s0 + DWP_TS);
    f32x16y acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = bias;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sX + rb[ks & 1] + 512 * (ks >> 1)), wf[ks], acc, 0, 0, 0);
    if constexpr (EPI == 0) {                              // plain linear layer: straight from the accumulators (two 128-byte row segments per store)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = s0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < t1) a.Y[(size_t)row * a.ldy + col] = acc[e];
      }
    } else {
    // the tile crosses LDS once (fp32 [32][260]) so that the epilogue runs on ROWS: 16-byte loads of the residuals, one dropout hash per four
    // columns (adt_keep4), 16-byte stores of U and Y -- on the transposed accumulators it was 4-byte accesses and one hash per element
#pragma unroll
    for (int e = 0; e < 16; ++e) sT[((e & 3) + 8 * (e >> 2) + 4 * h) * 260 + 32 * w + r] = acc[e];
    __syncthreads();
    const int colg = 256 * blockIdx.y + col4;
    float4 rv[4], r2v[4];
    int kp[4];
    if constexpr ((EPI & 1) != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = s0 + rsub + 8 * i, rc = row < t1 ? row : t1 - 1;
        rv[i] = a.R ? *reinterpret_cast<const float4*>(a.R + (size_t)rc * a.ldr + colg) : make_float4(0.f, 0.f, 0.f, 0.f);
        r2v[i] = a.R2 ? *reinterpret_cast<const float4*>(a.R2 + (size_t)rc * a.ldr2 + colg) : make_float4(0.f, 0.f, 0.f, 0.f);
        kp[i] = a.ids ? a.ids[rc] : 1;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = rsub + 8 * i, row = s0 + rl;
      if (row >= t1) continue;
      float4 v = *reinterpret_cast<const float4*>(sT + rl * 260 + col4);
      if constexpr ((EPI & 4) != 0) {
        if (a.U) *reinterpret_cast<float4*>(a.U + (size_t)row * a.ldu + colg) = v;
        v.x = act_apply(a.act, v.x); v.y = act_apply(a.act, v.y); v.z = act_apply(a.act, v.z); v.w = act_apply(a.act, v.w);
      }
      if constexpr ((EPI & 2) != 0) {
        const uint32_t bits = adt_keep4(key, (uint32_t)(row + a.row_offset) * (uint32_t)a.N + (uint32_t)colg, a.drop.thr);
        v.x = (bits & 1u) ? v.x * a.drop.scale : 0.f; v.y = (bits & 2u) ? v.y * a.drop.scale : 0.f;
        v.z = (bits & 4u) ? v.z * a.drop.scale : 0.f; v.w = (bits & 8u) ? v.w * a.drop.scale : 0.f;
      }
      if constexpr ((EPI & 1) != 0) {
        v.x += rv[i].x + r2v[i].x; v.y += rv[i].y + r2v[i].y; v.z += rv[i].z + r2v[i].z; v.w += rv[i].w + r2v[i].w;
        if (kp[i] == 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4*>(a.Y + (size_t)row * a.ldy + colg) = v;
    }
    }
    __syncthreads();
  }
}

// ---- input gradient of a 256 x 256 layer (bf16 operands): dX (+)= G W ------------------------------------------------------------------------
// The LCE / k_dense_dw256 skeleton with the roles turned: the WEIGHT is the register-resident operand (wave w keeps W[:, 32 w .. 32 w + 31] as
// sixteen B fragments, read once through ds_read_b64_tr_b16 from a staged image), the gradient streams through LDS in 32-token stages (one
// bf16 image in the dual-use layout, row reads), the output tile is transposed (k on the lane, token on the accumulator rows: every store
// instruction writes two 128-byte row segments).  The row-streaming kernel (adt_dense_rows.cuh) re-stages the 256 KB fp32 weight panel per
// workgroup and holds two waves per SIMD with no overlap between loads, MFMAs and stores: 52 us per layer; this one is bound by its HBM bytes.
template <int NB>      // N = 256 NB (the contraction: NB <= 3 keeps the 16 NB weight fragments in registers); K = 256 gridDim.y
__global__ __launch_bounds__(DWP_NTH) void k_dense_dx256(DenseBwdArgs a) {
  adt_prefetch_kernargs<sizeof(DenseBwdArgs) <= 512 ? sizeof(DenseBwdArgs) : 512>();
  extern __shared__ __attribute__((aligned(1024))) unsigned char dx_smem[];      // NB gradient images; image 0 doubles as the weight staging image
  unsigned char* sG = dx_smem;
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const int t0 = blockIdx.x * a.t_chunk;
  const int t1 = t0 + a.t_chunk < G.T ? t0 + a.t_chunk : G.T;
  if (t0 >= t1) return;
  G.T = t1;
  const int k0 = 256 * blockIdx.y;                         // this workgroup's 256 output columns
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int col4 = (threadIdx.x & 63) * 4, rsub = threadIdx.x >> 6;
  float4 gv[NB][4];
  auto request = [&](int s0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) gv[nb][i] = G.at(s0 + rsub + 8 * i, 256 * nb + col4);
  };
  // W^T fragments: element j of lane (k = k0 + 32 w + r, h) of step ks is W[16 ks + 8 h + j][k] -- the natural slot order of the row-read A operand
  bf16x8 wf[16 * NB];
  {
    const int li = lane & 15, q = li >> 2, p = li & 3, gi = (lane >> 4) & 1;
    int tbn[2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) tbn[j2] = 4096 * h + 64 * (4 * j2 + q) + 16 * ((2 * gi + (p >> 1)) ^ ((2 * h + j2) & 3)) + 8 * (p & 1);
#pragma unroll
    for (int pass = 0; pass < 8 * NB; ++pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = rsub + 8 * i;
        const float4 wv = *reinterpret_cast<const float4*>(a.W + (size_t)(32 * pass + rr) * a.ldw + k0 + col4);
        *reinterpret_cast<dw64_b4*>(sG + dwp_off(rr, col4 >> 3) + 8 * ((col4 >> 2) & 1)) = dw64_b4{(__bf16)wv.x, (__bf16)wv.y, (__bf16)wv.z, (__bf16)wv.w};
      }
      __syncthreads();
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        union { struct { dw64_s4 lo, hi; } p; bf16x8 v; } f;
        f.p.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(sG + tbn[0] + 8192 * sk + 512 * w));
        f.p.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dw64_s4 __attribute__((address_space(3)))*)(sG + tbn[1] + 8192 * sk + 512 * w));
        wf[2 * pass + sk] = f.v;
      }
      __syncthreads();
    }
  }
  request(t0);
  int rb[2];
  {
    const int v = (r >> 2) & 3;
    rb[0] = 4096 * (r >> 3) + 64 * (r & 7) + 16 * (h ^ v);
    rb[1] = 4096 * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ v);
  }
  typedef float f32x16x __attribute__((ext_vector_type(16)));
  for (int s0 = t0; s0 < t1; s0 += DWP_TS) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = rsub + 8 * i;
        *reinterpret_cast<dw64_b4*>(sG + DWP_IMG * nb + dwp_off(rr, col4 >> 3) + 8 * ((col4 >> 2) & 1)) =
            dw64_b4{(__bf16)gv[nb][i].x, (__bf16)gv[nb][i].y, (__bf16)gv[nb][i].z, (__bf16)gv[nb][i].w};
      }
    __syncthreads();
    if (s0 + DWP_TS < t1) request(s0 + DWP_TS);
    f32x16x acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sG + DWP_IMG * nb + rb[ks & 1] + 512 * (ks >> 1)), wf[16 * nb + ks], acc, 0, 0, 0);
    if (a.beta) {                                          // all sixteen old values first: interleaved with the stores they would be sixteen
      float old[16];                                       // serial round trips (the compiler cannot prove that the rows do not alias)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = s0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        old[e] = t < t1 ? a.dX[(size_t)t * a.lddx + k0 + 32 * w + r] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] += old[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int t = s0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (t < t1) a.dX[(size_t)t * a.lddx + k0 + 32 * w + r] = acc[e];
    }
    __syncthreads();
  }
}

// ---- weight gradient: dW += G^T X, db += colsum(G); T split over blockIdx.z, partials added with atomics ------
template <int PREC, int BN>
__global__ __launch_bounds__(GTH) void k_dense_bwd_dw(DenseBwdArgs a) {
  adt_prefetch_kernargs<sizeof(DenseBwdArgs) <= 512 ? sizeof(DenseBwdArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef typename GemmLds<PREC>::T LT;
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  LT* sA = reinterpret_cast<LT*>(gemm_smem);
  LT* sB = sA + GBM * GemmLds<PREC>::RS;
  int zz, inner;
  xcd_tile(a.nt_z, a.nt_a * a.nt_b, zz, inner);                    // all output tiles of one T chunk share its G and X rows
  const int tk = inner % a.nt_a;
  const int n0 = tk * BN, m0 = (inner / a.nt_a) * GBM;     // m0 indexes N (rows of dW), n0 indexes K (its columns)
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const PlainSrc Xs{a.X, a.ldx, G.T, a.K};
  const int t0 = zz * a.t_chunk;
  int t1 = t0 + a.t_chunk;
  if (t1 > G.T) t1 = G.T;
  if (t0 >= t1) return;
  GemmAcc<PREC, BN> acc;
  acc.zero();
  const int kend = t0 + (t1 - t0 + GBK - 1) / GBK * GBK;
  // rows beyond t1 inside the last 32-step belong to the next chunk: bound both sources at t1
  GradSrc Gc = G; Gc.T = t1;
  PlainSrc Xc = Xs; Xc.rows = t1;
  float rowsum = 0.f;
  if (a.db && tk == 0) gemm_core<PREC, BN, true, true, GradSrc, PlainSrc, true>(acc, Gc, Xc, m0, n0, t0, kend, sA, sB, rowsum);
  else gemm_core<PREC, BN, true, true, GradSrc, PlainSrc, false>(acc, Gc, Xc, m0, n0, t0, kend, sA, sB, rowsum);
  acc.foreach(m0, n0, [&](int row, int col, float v) {
    if (row >= G.N || col >= a.K) return;
    atomicAdd(a.dW + (size_t)row * a.lddw + col, v);
  });
  if (a.db && tk == 0 && threadIdx.x < GBM && m0 + (int)threadIdx.x < G.N) atomicAdd(a.db + m0 + threadIdx.x, rowsum);
}

}  // namespace adt
