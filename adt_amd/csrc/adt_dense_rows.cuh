// Row-streaming dense layers for bf16 operands and short contractions (<= 256): the weight panel is the ONLY thing in LDS.
//
// The tiled kernels of adt_gemm.cuh move both operands through LDS in 32-wide k-steps with a barrier pair per step, and every
// column tile re-reads, re-converts and (in the backward) re-derives its activation rows.  At d = 256 those GEMMs are bound by
// HBM (fp32 activations: 104 MB per 51,200 x 256 x 256 layer against 6.7 GFLOP), not by MFMA, so here the activations never
// touch LDS: a wave owns 16 rows, loads them from global memory directly in MFMA operand layout (lane (c, g) of row c takes
// k = 32 kb + {4g..4g+3, 16+4g..16+4g+3}: two 16-byte loads, 64 contiguous bytes per row and instruction), converts them to
// bf16 once, and sweeps the whole weight panel (<= 256 output columns, bf16, k permuted to the same slot order, staged once per
// workgroup) with one ds_read_b128 per MFMA.  No barrier after the panel is staged; the next tile's rows are in flight while
// the current one is multiplied; the epilogue / prologue arithmetic (bias, activation, dropout hash, residuals, act') runs
// once per element instead of once per column tile.  Output orientation: acc[r] = Y[row0 + c][n0 + 16 nt + 4g + r], i.e. one
// 16-byte store per lane and tile.
//
// Replaces the same reference lines as adt_gemm.cuh (nn.Linear / Conv1d(k=1) forward and input gradient); the weight gradient
// stays on the tiled split-T kernel.
#pragma once
#include "adt_gemm.cuh"

namespace adt {

constexpr int ROWS_NW = 8;                  // waves per workgroup
constexpr int ROWS_PC = 256;                // output columns per weight panel

// position of contraction index k (0..31 inside its 32-block) in the permuted panel row: lane g reads 8 consecutive elements
ADT_DEVICE_INLINE int rows_slot_pos(int k32) { return k32 < 16 ? ((k32 >> 2) * 8 + (k32 & 3)) : (((k32 - 16) >> 2) * 8 + 4 + (k32 & 3)); }

template <int KB> struct RowsRaw { float4 lo[KB], hi[KB]; };

ADT_DEVICE_INLINE bf16x8 rows_pack(const float4& lo, const float4& hi) {
  bf16x8 o;
  o[0] = (__bf16)lo.x; o[1] = (__bf16)lo.y; o[2] = (__bf16)lo.z; o[3] = (__bf16)lo.w;
  o[4] = (__bf16)hi.x; o[5] = (__bf16)hi.y; o[6] = (__bf16)hi.z; o[7] = (__bf16)hi.w;
  return o;
}

static inline size_t rows_lds_bytes(int contraction, int pc) { return (size_t)pc * (contraction + 8) * sizeof(__bf16) + (size_t)pc * sizeof(float); }

// ---- forward: Y = mask(R + R2 + dropout(act(X W^T + b))) ------------------------------------------------------------------
// panel: rows n0..n0+pc of W (N x K) -> sW[pc][K + 8], k permuted; rows >= N are zero
template <int KB, int NTH>
ADT_DEVICE_INLINE void rows_stage_w(__bf16* sW, const float* W, int ldw, int n0, int N, int pc) {
  constexpr int K = KB * 32, RS = K + 8, V4 = K / 4, UN = 8;
  const int total = pc * V4;
  const int rot = (int)((blockIdx.x * 37u) % (unsigned)pc);   // each workgroup starts at a different panel row: spreads the L2 requests
  // UN independent 16-byte loads in flight per thread before the first conversion (one L2 latency per batch, not per element)
  for (int base = threadIdx.x; base < total; base += NTH * UN) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = base + u * NTH;
      const int r = (i / V4 + rot) % pc, k = (i % V4) * 4;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < total && n0 + r < N) v[u] = *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * ldw + k);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = base + u * NTH;
      if (i >= total) continue;
      const int r = (i / V4 + rot) % pc, k = (i % V4) * 4;
      gbf16x4 b;
      b[0] = (__bf16)v[u].x; b[1] = (__bf16)v[u].y; b[2] = (__bf16)v[u].z; b[3] = (__bf16)v[u].w;
      *reinterpret_cast<gbf16x4*>(sW + r * RS + (k & ~31) + rows_slot_pos(k & 31)) = b;
    }
  }
}

template <int KB>
ADT_DEVICE_INLINE RowsRaw<KB> rows_load_x(const float* X, int ldx, int row, int T, int g) {
  RowsRaw<KB> x;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    x.lo[kb] = make_float4(0.f, 0.f, 0.f, 0.f);
    x.hi[kb] = x.lo[kb];
  }
  if (row < T) {
    const float* p = X + (size_t)row * ldx + 4 * g;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      x.lo[kb] = *reinterpret_cast<const float4*>(p + kb * 32);
      x.hi[kb] = *reinterpret_cast<const float4*>(p + kb * 32 + 16);
    }
  }
  return x;
}

// Residual operands of CH consecutive 16-column tiles of one row (lane (c, g): row c, columns 4g..4g+3 of each tile).
// gfx9 returns vector-memory operations in order and counts stores in vmcnt: a load issued after a store cannot be waited for
// without draining that store.  So every load of the loop (next rows of X, next chunk's residuals, next tile's row mask) is
// issued BEFORE the stores of the chunk that is being computed, one stage ahead, and the bias sits in LDS.
template <int CH> struct RowsRes { float4 r1[CH]; float4 r2[CH == 4 ? CH : 1]; };    // two residuals only in the CH = 4 build

template <int CH>
ADT_DEVICE_INLINE void rows_load_res(RowsRes<CH>& o, const DenseFwdArgs& a, int row, int col_base) {
  constexpr bool HAS_R2 = CH == 4;
  const bool row_ok = row < a.T;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int col = col_base + j * 16;
    const bool ok = row_ok && col < a.N;
    o.r1[j] = (ok && a.R) ? *reinterpret_cast<const float4*>(a.R + (size_t)row * a.ldr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_R2) o.r2[HAS_R2 ? j : 0] = (ok && a.R2) ? *reinterpret_cast<const float4*>(a.R2 + (size_t)row * a.ldr2 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int KB, int CH>
__global__ __launch_bounds__(ROWS_NW * 64) void k_dense_fwd_rows(DenseFwdArgs a, int n_panels, int pc) {
  adt_prefetch_kernargs<sizeof(DenseFwdArgs) <= 512 ? sizeof(DenseFwdArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  constexpr int K = KB * 32, RS = K + 8, NW = ROWS_NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char rows_smem[];
  __bf16* sW = reinterpret_cast<__bf16*>(rows_smem);
  float* sBias = reinterpret_cast<float*>(rows_smem + (size_t)pc * RS * sizeof(__bf16));     // [pc]
  const int panel = blockIdx.x % n_panels, rg = blockIdx.x / n_panels, nrg = gridDim.x / n_panels;
  const int n0 = panel * pc;
  if (a.t_dev && a.T > *a.t_dev) a.T = *a.t_dev;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int ntiles = (a.T + 15) / 16, stride = nrg * NW;
  int tile = rg * NW + w;
  RowsRaw<KB> raw = rows_load_x<KB>(a.X, a.ldx, tile * 16 + c, a.T, g);     // in flight while the panel is staged
  RowsRes<CH> cur, nxt;
  rows_load_res<CH>(cur, a, tile * 16 + c, n0 + 4 * g);
  int keep_next = 1;
  if (a.ids && tile * 16 + c < a.T) keep_next = a.ids[tile * 16 + c];
  rows_stage_w<KB, NW * 64>(sW, a.W, a.ldw, n0, a.N, pc);
  for (int i = threadIdx.x; i < pc; i += NW * 64) sBias[i] = (a.b && n0 + i < a.N) ? a.b[n0 + i] : 0.f;
  __syncthreads();
  const int ncols = a.N - n0 < pc ? a.N - n0 : pc;
  const int ntn = (ncols + 15) / 16, nch = (ntn + CH - 1) / CH;
  const uint32_t key = drop_key(a.drop);
  for (; tile < ntiles; tile += stride) {
    bf16x8 fx[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) fx[kb] = rows_pack(raw.lo[kb], raw.hi[kb]);
    const int row = tile * 16 + c, row_n = row + stride * 16;
    raw = rows_load_x<KB>(a.X, a.ldx, row_n, a.T, g);
    const bool keep = keep_next != 0;
    keep_next = 1;
    if (a.ids && row_n < a.T) keep_next = a.ids[row_n];
    const bool row_ok = row < a.T;
    const uint32_t idx_row = (uint32_t)(row + a.row_offset) * (uint32_t)a.N;
    for (int ch = 0; ch < nch; ++ch) {
      if (a.R || a.R2) {
        if (ch + 1 < nch) rows_load_res<CH>(nxt, a, row, n0 + (ch + 1) * CH * 16 + 4 * g);
        else rows_load_res<CH>(nxt, a, row_n, n0 + 4 * g);
      }
#pragma unroll
      for (int j = 0; j < CH; j += 2) {
        const int nt = ch * CH + j;
        if (nt >= ntn) break;
        const bool two = nt + 1 < ntn;
        const int col0 = n0 + nt * 16 + 4 * g, col1 = col0 + 16;
        const bool ok0 = row_ok && col0 < a.N, ok1 = two && row_ok && col1 < a.N;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const __bf16* w0 = sW + (nt * 16 + c) * RS + 8 * g;
        const __bf16* w1 = w0 + 16 * RS;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(w0 + kb * 32), fx[kb], acc0, 0, 0, 0);
          if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(w1 + kb * 32), fx[kb], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (!(t ? ok1 : ok0)) continue;
          const int col = t ? col1 : col0;
          const f32x4 acc = t ? acc1 : acc0;
          const float4 bias = *reinterpret_cast<const float4*>(sBias + (col - n0));
          float v[4] = {acc[0] + bias.x, acc[1] + bias.y, acc[2] + bias.z, acc[3] + bias.w};
          if (a.U) *reinterpret_cast<float4*>(a.U + (size_t)row * a.ldu + col) = make_float4(v[0], v[1], v[2], v[3]);
          float4 q1 = cur.r1[j + t];
          if (CH == 4) {
            const float4 q2 = cur.r2[CH == 4 ? j + t : 0];
            q1.x += q2.x; q1.y += q2.y; q1.z += q2.z; q1.w += q2.w;
          }
          const float rr[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float y = act_apply(a.act, v[r]);
            if (a.drop.thr) y = adt_keep(key, idx_row + (uint32_t)(col + r), a.drop.thr) ? y * a.drop.scale : 0.f;
            y += rr[r];
            v[r] = keep ? y : 0.f;
          }
          *reinterpret_cast<float4*>(a.Y + (size_t)row * a.ldy + col) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      if (a.R || a.R2) cur = nxt;
    }
  }
}

// ---- input gradient: dX = (beta ? dX : 0) + G W, contraction over N <= 256 ------------------------------------------------------
// panel: columns k0..k0+pc of W (N x K), transposed -> sWT[pc][N + 8] with n permuted; columns >= K are zero
template <int NB, int NTH>
ADT_DEVICE_INLINE void rows_stage_wt(__bf16* sWT, const float* W, int ldw, int k0, int K, int pc, int n_valid) {
  constexpr int N = NB * 32, RS = N + 8, UN = 8;
  const int v4 = pc / 4, total = N * v4;
  for (int base = threadIdx.x; base < total; base += NTH * UN) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = base + u * NTH;
      const int n = i / v4, k = (i % v4) * 4;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < total && n < n_valid && k0 + k < K) v[u] = *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k0 + k);     // K % 4 == 0
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = base + u * NTH;
      if (i >= total) continue;
      const int n = i / v4, k = (i % v4) * 4;
      __bf16* dst = sWT + k * RS + (n & ~31) + rows_slot_pos(n & 31);
      dst[0] = (__bf16)v[u].x; dst[RS] = (__bf16)v[u].y; dst[2 * RS] = (__bf16)v[u].z; dst[3 * RS] = (__bf16)v[u].w;
    }
  }
}

// Raw operands of G for one row: dY (and the saved pre-activation U when the layer has an activation), in fragment order.
// The prologue arithmetic (row mask, dropout hash, act') is applied when the tile is converted, one tile after the loads
// were issued, so the loads of the next tile fly while the current one is multiplied.
template <int NB, bool HAS_U> struct RowsRawG { RowsRaw<NB> dy; RowsRaw<HAS_U ? NB : 1> u; int id; };

template <int NB, bool HAS_U>
ADT_DEVICE_INLINE void rows_load_g(RowsRawG<NB, HAS_U>& x, const GradSrc& G, int row, int g) {
  x.dy = rows_load_x<NB>(G.dY, G.lddy, row, G.T, g);
  if (HAS_U) x.u = rows_load_x<HAS_U ? NB : 1>(G.U, G.ldu, row, G.T, g);
  x.id = 1;
  if (G.ids && row < G.T) x.id = G.ids[row];
}

template <int NB, bool HAS_U>
ADT_DEVICE_INLINE void rows_g_frags(bf16x8 (&fg)[NB], const RowsRawG<NB, HAS_U>& x, const GradSrc& G, int row, int g) {
  const bool live = row < G.T && x.id != 0;
  const uint32_t base = (uint32_t)(row + G.row_offset) * (uint32_t)G.idx_ld + (uint32_t)(G.idx_off + 4 * g);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    float v[8] = {x.dy.lo[nb].x, x.dy.lo[nb].y, x.dy.lo[nb].z, x.dy.lo[nb].w, x.dy.hi[nb].x, x.dy.hi[nb].y, x.dy.hi[nb].z, x.dy.hi[nb].w};
    if (G.drop.thr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t n = (uint32_t)(nb * 32 + (j < 4 ? j : 12 + j));        // 4g + j | 16 + 4g + (j - 4), the 4g is in base
        v[j] = adt_keep(G.key, base + n, G.drop.thr) ? v[j] * G.drop.scale : 0.f;
      }
    }
    if (HAS_U) {
      const int ub = HAS_U ? nb : 0;
      const float u[8] = {x.u.lo[ub].x, x.u.lo[ub].y, x.u.lo[ub].z, x.u.lo[ub].w, x.u.hi[ub].x, x.u.hi[ub].y, x.u.hi[ub].z, x.u.hi[ub].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= act_grad(G.act, u[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) fg[nb][j] = (__bf16)(live ? v[j] : 0.f);
  }
}

template <int CH> struct RowsOld { float4 v[CH]; };

template <int CH>
ADT_DEVICE_INLINE void rows_load_old(RowsOld<CH>& o, const DenseBwdArgs& a, int row, int T, int col_base) {
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int col = col_base + j * 16;
    o.v[j] = (row < T && col < a.K) ? *reinterpret_cast<const float4*>(a.dX + (size_t)row * a.lddx + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int NB, bool HAS_U>
__global__ __launch_bounds__(ROWS_NW * 64) void k_dense_dx_rows(DenseBwdArgs a, int n_panels, int pc) {
  adt_prefetch_kernargs<sizeof(DenseBwdArgs) <= 512 ? sizeof(DenseBwdArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  constexpr int N = NB * 32, RS = N + 8, NW = ROWS_NW, CH = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char rows_smem[];
  __bf16* sWT = reinterpret_cast<__bf16*>(rows_smem);
  const int panel = blockIdx.x % n_panels, rg = blockIdx.x / n_panels, nrg = gridDim.x / n_panels;
  const int k0 = panel * pc;
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int ntiles = (G.T + 15) / 16, stride = nrg * NW;
  int tile = rg * NW + w;
  RowsRawG<NB, HAS_U> raw;
  rows_load_g<NB, HAS_U>(raw, G, tile * 16 + c, g);
  RowsOld<CH> cur, nxt;
#pragma unroll
  for (int j = 0; j < CH; ++j) cur.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.beta) rows_load_old<CH>(cur, a, tile * 16 + c, G.T, k0 + 4 * g);
  rows_stage_wt<NB, NW * 64>(sWT, a.W, a.ldw, k0, a.K, pc, G.N);
  __syncthreads();
  const int ncols = a.K - k0 < pc ? a.K - k0 : pc;
  const int ntn = (ncols + 15) / 16, nch = (ntn + CH - 1) / CH;
  for (; tile < ntiles; tile += stride) {
    const int row = tile * 16 + c, row_n = row + stride * 16;
    bf16x8 fg[NB];
    rows_g_frags<NB, HAS_U>(fg, raw, G, row, g);
    rows_load_g<NB, HAS_U>(raw, G, row_n, g);
    const bool row_ok = row < G.T;
    for (int ch = 0; ch < nch; ++ch) {
      if (a.beta) {
        if (ch + 1 < nch) rows_load_old<CH>(nxt, a, row, G.T, k0 + (ch + 1) * CH * 16 + 4 * g);
        else rows_load_old<CH>(nxt, a, row_n, G.T, k0 + 4 * g);
      }
#pragma unroll
      for (int j = 0; j < CH; j += 2) {
        const int nt = ch * CH + j;
        if (nt >= ntn) break;
        const bool two = nt + 1 < ntn;
        const int col0 = k0 + nt * 16 + 4 * g, col1 = col0 + 16;
        const bool ok0 = row_ok && col0 < a.K, ok1 = two && row_ok && col1 < a.K;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const __bf16* w0 = sWT + (nt * 16 + c) * RS + 8 * g;
        const __bf16* w1 = w0 + 16 * RS;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(w0 + nb * 32), fg[nb], acc0, 0, 0, 0);
          if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(w1 + nb * 32), fg[nb], acc1, 0, 0, 0);
        }
        const float4 o0 = cur.v[j], o1 = cur.v[j + 1];
        if (ok0) *reinterpret_cast<float4*>(a.dX + (size_t)row * a.lddx + col0) = make_float4(o0.x + acc0[0], o0.y + acc0[1], o0.z + acc0[2], o0.w + acc0[3]);
        if (ok1) *reinterpret_cast<float4*>(a.dX + (size_t)row * a.lddx + col1) = make_float4(o1.x + acc1[0], o1.y + acc1[1], o1.z + acc1[2], o1.w + acc1[3]);
      }
      if (a.beta) cur = nxt;
    }
  }
}


// ---- weight gradient: dW += G^T X, db += colsum(G); T split over workgroups, partials added with atomics ----------------------------
// Output block 256 (n) x 128 (k) per 512-thread workgroup, 64 rows of T per stage, double-buffered LDS, ONE barrier per stage.
// Both operands are contracted over their slow (row) index, so they go to LDS "octet-interleaved": element (t, col) of a stage
// lives at ((t / 8) * COLS + col) * 8 + t % 8 -- the eight t of an octet are the eight k-slots of one MFMA lane, i.e. one
// ds_read_b128 per operand fragment and no transposition anywhere: the staging thread that owns (octet, 4 columns) loads its
// 8 rows x 16 bytes with coalesced 16-byte loads and writes four packed 16-byte words (the X side: 4 rows, 8-byte words).
// acc[r] = dW[n0 + 16 nt + 4g + r][k0 + 16 kt + c].
constexpr int DW_BN = 256, DW_BK = 128, DW_TS = 64, DW_NTH = 512;
constexpr size_t DW_LDS_BYTES = 2 * (size_t)(8 * DW_BN * 8 + 8 * DW_BK * 8) * sizeof(__bf16);     // 96 KB

struct DwStage { float4 g[8]; float4 x[4]; };

__global__ __launch_bounds__(DW_NTH) void k_dense_dw_rows(DenseBwdArgs a, int n_blocks, int k_blocks) {
  adt_prefetch_kernargs<sizeof(DenseBwdArgs) <= 512 ? sizeof(DenseBwdArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  extern __shared__ __attribute__((aligned(16))) unsigned char rows_smem[];
  __bf16* sG = reinterpret_cast<__bf16*>(rows_smem);                    // [2][8][DW_BN][8]
  __bf16* sX = sG + 2 * 8 * DW_BN * 8;                                  // [2][8][DW_BK][8]
  const int tiles = n_blocks * k_blocks;
  // all output blocks of one T chunk run on ONE XCD (xcd_tile): the chunk's G and X rows are fetched into that L2 once instead
  // of once per block (N = 1024, K = 256: 8 blocks, 630 MB of reads without the mapping against 262 MB of distinct data)
  int split, tile;
  xcd_tile(a.nt_z, tiles, split, tile);
  if (split >= a.nt_z) return;
  const int n0 = (tile / k_blocks) * DW_BN, k0 = (tile % k_blocks) * DW_BK;
  GradSrc G = a.G;
  if (a.t_dev && G.T > *a.t_dev) G.T = *a.t_dev;
  G.key = drop_key(G.drop);
  const int t_begin = split * a.t_chunk;
  int t_end = t_begin + a.t_chunk;
  if (t_end > G.T) t_end = G.T;
  if (t_begin >= t_end) return;
  G.T = t_end;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
  // staging roles
  const int go = tid >> 6, gn = (tid & 63) * 4;                          // G: octet go, columns gn..gn+3, rows 8 go + 0..7
  const int xo = tid >> 6, xh = (tid >> 5) & 1, xc = (tid & 31) * 4;     // X: octet xo, rows 8 xo + 4 xh + 0..3, columns xc..xc+3
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_db = a.db != nullptr && k0 == 0;
  auto load_stage = [&](DwStage& s, int t0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s.g[j] = G.at(t0 + 8 * go + j, n0 + gn);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + 8 * xo + 4 * xh + j, k = k0 + xc;
      s.x[j] = (t < t_end && k < a.K) ? *reinterpret_cast<const float4*>(a.X + (size_t)t * a.ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_stage = [&](const DwStage& s, int buf) {
    __bf16* dg = sG + ((size_t)(buf * 8 + go) * DW_BN + gn) * 8;
    const float* f = reinterpret_cast<const float*>(s.g);
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // column gn + q: the 8 rows of the octet
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[4 * j + q];
      *reinterpret_cast<bf16x8*>(dg + q * 8) = v;
      if (want_db) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[q] += f[4 * j + q];
      }
    }
    __bf16* dx = sX + ((size_t)(buf * 8 + xo) * DW_BK + xc) * 8 + 4 * xh;
    const float* fx = reinterpret_cast<const float*>(s.x);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gbf16x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (__bf16)fx[4 * j + q];
      *reinterpret_cast<gbf16x4*>(dx + q * 8) = v;
    }
  };
  const int wn = (w & 3) * 64, wk = (w >> 2) * 64;     // this wave's 64 x 64 sub-block
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  DwStage st;
  load_stage(st, t_begin);
  store_stage(st, 0);
  __syncthreads();
  const int nstages = (t_end - t_begin + DW_TS - 1) / DW_TS;
  for (int s = 0; s < nstages; ++s) {
    const bool more = s + 1 < nstages;
    if (more) load_stage(st, t_begin + (s + 1) * DW_TS);
    const __bf16* bg = sG + (size_t)((s & 1) * 8) * DW_BN * 8;
    const __bf16* bx = sX + (size_t)((s & 1) * 8) * DW_BK * 8;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(bg + ((size_t)(kb * 4 + g) * DW_BN + wn + i * 16 + c) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bx + ((size_t)(kb * 4 + g) * DW_BK + wk + j * 16 + c) * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_stage(st, (s + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + wn + i * 16 + 4 * g + r, col = k0 + wk + j * 16 + c;
        if (row < a.G.N && col < a.K) atomicAdd(a.dW + (size_t)row * a.lddw + col, acc[i][j][r]);
      }
  if (want_db) {
    // the eight octet-threads of a column meet in LDS first: ONE global atomic per column and workgroup (hundreds of workgroups
    // hitting the same 256 addresses eight times each cost more than the whole product)
    float* sdb = reinterpret_cast<float*>(rows_smem);          // the stage buffers are free after the last barrier
    if (tid < DW_BN) sdb[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(sdb + gn + q, bsum[q]);
    __syncthreads();
    if (tid < DW_BN && n0 + tid < a.G.N) atomicAdd(a.db + n0 + tid, sdb[tid]);
  }
}

}  // namespace adt
