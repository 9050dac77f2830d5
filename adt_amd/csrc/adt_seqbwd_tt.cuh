// Per-sequence fused BACKWARD of one attention block on the transposed chains of adt_tt.cuh (bf16 MFMA operands):
//   LayerNorm + packed in-projection (recomputed) -> causal attention backward -> in-projection weight gradients -> gradient of the
//   block input.  One workgroup = one user sequence, replacing per (layer, block) the staged k_attn_bwd / k_seq_attn_bwd + k_pre_bwd pair
//   (sasrec/modules.py:644-647, :666-670 and their autograd) and the trip of q, k, v, dq, dk, dv through HBM between them.
//
// q, k, v are not read back: the layer input x is (fp32, 256 B per token), LayerNorm and the three 64x64 products are recomputed into
// registers (24 MFMAs per 16-token tile), and k, v go straight into the LDS row images the attention sweeps.  dO / O come from the
// out_proj backward.  The two passes of adt_seqattn.cuh run on two image slots: K, V for pass A (the wave's own q and dO rows are
// MFMA operands in registers), then Q, dO for pass B (own k, v rows in registers).  dq, dk, dv stay in registers as transposed tiles:
//   * weight gradients dW = G^T X contract over ALL tokens of the sequence: G and X rows go to the two image slots and every wave
//     computes two of the 16 output tiles through ds_read_b64_tr_b16 fragments (bias gradients: per-lane column sums);
//   * the input gradient is two / three more transposed-chain products and the LayerNorm backward, all in registers.
#pragma once
#include "adt_seqattn.cuh"
#include "adt_seqbwd_args.h"
#include "adt_seqfwd_tt.cuh"
#include "adt_tt.cuh"

namespace adt {

constexpr int SB_NW = 8;
constexpr int SB_R = 224;                          // image rows: L <= 224

template <int H>
struct SeqBwdLds {
  static constexpr size_t wbytes = 6 * (size_t)TT_WIMG * 2, ibytes = (size_t)SB_R * TT_RS * 2;
  static constexpr size_t mbytes = (size_t)H * SB_R * 8 * 4, sbytes = 2 * (size_t)H * SB_R * 4, rbytes = (320 + 320) * 4;         // sRed: dgamma, dbeta, dbin ; sVec: gamma, beta, bin
  // the three plain weight images (dead behind P1) lie at the END of the allocation: a third token image overlays them in P4
  static constexpr size_t pbytes = ibytes + 64 > wbytes / 2 ? ibytes + 64 : wbytes / 2;
  static constexpr size_t poff = wbytes / 2 + 2 * ibytes + 64 /* spill of the last row's ones-column read */ + mbytes + sbytes + rbytes;
  static constexpr size_t bytes = poff + pbytes;
};

// operands of every head from a transposed tile, in the slot order sab_rowfrag reads image rows in: f[h * KB + kb]
template <int HD>
ADT_DEVICE_INLINE void sb_frags(const TT& t, float mul, bf16x8 (&f)[(64 / HD) * ((HD + 31) / 32)]) {
  if constexpr (HD >= 32) {
    f[0] = tt_pack(t.v[0], t.v[1], mul);
    f[1] = tt_pack(t.v[2], t.v[3], mul);
  } else {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 4; ++h) f[h] = tt_pack(t.v[h], z, mul);
  }
}

// rows of a natural-order image from such operands (the inverse of sab_rowfrag)
template <int HD>
ADT_DEVICE_INLINE void sb_put_frags(__bf16* img, int token, const bf16x8 (&f)[(64 / HD) * ((HD + 31) / 32)], bool valid, int g) {
  constexpr int NF = (64 / HD) * ((HD + 31) / 32);
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    bf16x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) { lo[j] = valid ? f[i][j] : (__bf16)0.f; hi[j] = valid ? f[i][4 + j] : (__bf16)0.f; }
    if constexpr (HD >= 32) {
      *reinterpret_cast<bf16x4*>(img + token * TT_RS + 32 * i + 4 * g) = lo;
      *reinterpret_cast<bf16x4*>(img + token * TT_RS + 32 * i + 16 + 4 * g) = hi;
    } else {
      *reinterpret_cast<bf16x4*>(img + token * TT_RS + 16 * i + 4 * g) = lo;
    }
  }
}

// LayerNorm of a transposed tile, keeping what its backward needs
struct TTLn { TT xhat; float rstd; };
ADT_DEVICE_INLINE TTLn tt_ln_stats(const TT& x, float eps) {
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) s += (x.v[nt][0] + x.v[nt][1]) + (x.v[nt][2] + x.v[nt][3]);
  const float mu = tt_colsum(s) * (1.0f / 64);
  TTLn o;
  float q = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float t = x.v[nt][r] - mu; o.xhat.v[nt][r] = t; q += t * t; }
  o.rstd = 1.0f / sqrtf(tt_colsum(q) * (1.0f / 64) + eps);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) o.xhat.v[nt] *= o.rstd;
  return o;
}
ADT_DEVICE_INLINE TT tt_ln_apply(const TT& xhat, const float* gamma, const float* beta, int g) {
  const TTV gm = tt_vec(gamma, g), bt = tt_vec(beta, g);
  TT y;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) y.v[nt][r] = xhat.v[nt][r] * gm.v[nt][r] + bt.v[nt][r];
  return y;
}
// dx = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)), dxh = dy * gamma ; dgamma += dy * xhat, dbeta += dy (per-lane partials)
ADT_DEVICE_INLINE TT tt_ln_bwd(const TT& dy, const TTLn& st, const float* gamma, TT& dgm, TT& dbt, int g) {
  const TTV gm = tt_vec(gamma, g);
  TT dx;
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dgm.v[nt][r] += dy.v[nt][r] * st.xhat.v[nt][r];
      dbt.v[nt][r] += dy.v[nt][r];
      const float dxh = dy.v[nt][r] * gm.v[nt][r];
      dx.v[nt][r] = dxh;
      m1 += dxh;
      m2 += dxh * st.xhat.v[nt][r];
    }
  m1 = tt_colsum(m1) * (1.0f / 64);
  m2 = tt_colsum(m2) * (1.0f / 64);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dx.v[nt][r] = st.rstd * (dx.v[nt][r] - m1 - st.xhat.v[nt][r] * m2);
  return dx;
}

// One weight-gradient product over all tokens of the sequence: dW[n][k] += sum_t G[t][n] X[t][k], from two natural-order LDS row images
// through ds_read_b64_tr_b16 fragments; every wave calls (no barrier inside).
// 16 output tiles, exactly two per wave -- (nt0, kt) and (nt0 + 2, kt), which share the X
// fragments -- and the token loop unrolled (NP pairs of 16-token tiles; image rows beyond the sequence are zero).  A 20-tile form with a ones column in the
// X image for the bias gradient gave four waves a third tile: 10k cycles per product on the critical path against 3k for the two-tile waves (profiles/r02_stamps_post.txt).
//
// Where the result goes.  part == nullptr: float atomics into dW (a replica of the gradient).  They execute at the memory side at
// ~1.3 TB/s chip-wide (MI355X_MICROARCH.md, Global float atomics): 16 KB per product and workgroup is ~8k cycles during which the wave's
// next vector-memory instruction cannot issue (profiles/r02_stamps_post.txt), ~100 us per training step over the 32 products of a
// 2-layer model.  part != nullptr: this workgroup's PRIVATE 4,096-float partial of the block, written with plain 256-byte stores in
// register order -- element ((tile * 4 + r) * 64 + lane), tile = 4 nt + kt -- and summed over the workgroups, in a fixed order, by
// k_dwpart_reduce (adt_seq.hip).  Nothing is zeroed and the sum does not depend on timing.
// Bias gradient.  bred != nullptr: the column sums of G over the tokens (the bias gradient of the same layer) come out of the same
// fragments: the two waves with kt == 0 multiply their G fragments by a fragment of ones as well (14 more MFMAs on two waves) and store the
// 64 sums to bred[0..64) (LDS) -- every column of such a product holds sum_t G[t][n].  The sums are those of the bf16 image rows, like the
// weight gradient beside them.  (Per-lane fp32 sums + a 16-lane DPP reduction of 16 registers + LDS atomics, sb_colsum_flush, cost every
// wave ~1.1k cycles per layer on the critical path in front of the product's barrier: profiles/r03_stamps_attn_pre_bwd.txt.)
// wave w computes output tile w (nt = w >> 2, kt = w & 3) and, when `two` (wave-uniform), also the tile (nt + step, kt) that shares its X fragments:
// NW = 8: every wave, step 2 (tiles w and w + 8); NW = 12: waves 0 .. 3, step 3 (tiles w and w + 12); NW = 16: none.
template <int NP, int STEP, bool BIAS>
ADT_DEVICE_INLINE void sb_dw_accumulate(const __bf16* sG, const __bf16* sX, int kt, int nt0, bool two, int c, int g, f32x4& acc0, f32x4& acc1, f32x4& b0, f32x4& b1) {
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
#pragma unroll
  for (int kp = 0; kp < NP; ++kp) {
    const bf16x8 fx = tt_trfrag(sX, kp * 32, 16 * kt, c, g);
    const bf16x8 g0 = tt_trfrag(sG, kp * 32, 16 * nt0, c, g);
    acc0 = mfma_bf16(acc0, g0, fx);
    if (BIAS) b0 = mfma_bf16(b0, g0, ones);
    if (STEP > 0 && two) {
      const bf16x8 g1 = tt_trfrag(sG, kp * 32, 16 * (nt0 + STEP), c, g);
      acc1 = mfma_bf16(acc1, g1, fx);
      if (BIAS) b1 = mfma_bf16(b1, g1, ones);
    }
  }
}
template <int NP, int NW = SB_NW>
ADT_DEVICE_INLINE void sb_dw_tiles(const __bf16* sG, const __bf16* sX, float* dW, float* part, float* bred, int w, int c, int g) {
  constexpr int STEP = NW == 8 ? 2 : (NW == 12 ? 3 : 0);      // second tile of a wave: + STEP n-tiles = + NW tiles
  const int kt = w & 3, nt0 = w >> 2;
  const bool two = NW == 8 || (NW == 12 && w < 4);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
  if (bred != nullptr && kt == 0) {
    sb_dw_accumulate<NP, STEP, true>(sG, sX, kt, nt0, two, c, g, acc0, acc1, b0, b1);
    if (c == 0) {
      *reinterpret_cast<f32x4*>(bred + 16 * nt0 + 4 * g) = b0;             // rows 4 g + r of the tile, any column
      if (two) *reinterpret_cast<f32x4*>(bred + 16 * (nt0 + STEP) + 4 * g) = b1;
    }
  } else {
    sb_dw_accumulate<NP, STEP, false>(sG, sX, kt, nt0, two, c, g, acc0, acc1, b0, b1);
  }
  if (part) {
    // private partial of this workgroup, bf16: element (tile, lane, r) at (tile * 64 + lane) * 4 + r of the slot -- a lane's four accumulator
    // registers are one 8-byte store.  The sum over the workgroups runs in fp32 (k_fold_parts_gradnorm / k_dwpart_reduce); rounding a
    // per-sequence partial to bf16 (2^-9 relative) is far below what the bf16 operands of the product already cost, and halves the
    // 268 MB per step the partials used to move.
    const int lane = 16 * g + c;
    __bf16* pb = reinterpret_cast<__bf16*>(part);
    bf16x4 v0, v1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { v0[r] = (__bf16)acc0[r]; v1[r] = (__bf16)acc1[r]; }
    *reinterpret_cast<bf16x4*>(pb + ((size_t)w * 64 + lane) * 4) = v0;              // tile 4 nt0 + kt == w
    if (two) *reinterpret_cast<bf16x4*>(pb + ((size_t)(w + NW) * 64 + lane) * 4) = v1;
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    atomicAdd(dW + (16 * nt0 + 4 * g + r) * 64 + 16 * kt + c, acc0[r]);
    if (two) atomicAdd(dW + (16 * (nt0 + STEP) + 4 * g + r) * 64 + 16 * kt + c, acc1[r]);
  }
}
template <int NW = SB_NW>
ADT_DEVICE_INLINE void sb_dw_product16(const __bf16* sG, const __bf16* sX, int npair, float* dW, float* part, float* bred, int w, int c, int g) {
  static_assert(NW == 8 || NW == 12 || NW == 16, "8 waves (two output tiles each), 12 (waves 0..3 two) or 16 (one each)");
  if (npair <= 4) sb_dw_tiles<4, NW>(sG, sX, dW, part, bred, w, c, g);
  else sb_dw_tiles<SB_R / 32, NW>(sG, sX, dW, part, bred, w, c, g);
}
// bias gradients (column sums of G over the tokens): per-lane sums of the tiles a wave holds, reduced over the 16 tokens of a lane row,
// added to a 64-float LDS vector; the workgroup adds the vector to the global accumulator once, at its end
ADT_DEVICE_INLINE void sb_colsum_flush(float* red, const TT& acc, int c, int g) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sum = tt_rowsum16(acc.v[nt][r]);
      if (c == 0) atomicAdd(red + 16 * nt + 4 * g + r, sum);
    }
}

// N weight images global -> LDS in two steps: all the global loads (2 x 16 B per thread and image), later the LDS stores -- the loads are
// the first vector-memory instructions of the kernel, the phase-1 activation loads queue behind them, and both stream during the zero-fills
template <int N> struct SbImgRegs { uint4 r0[N], r1[N]; };
template <int N>
ADT_DEVICE_INLINE SbImgRegs<N> sb_img_load(const __bf16* const (&src)[N]) {
  constexpr int CH = TT_WIMG * 2 / 16;
  SbImgRegs<N> t;
  const int i1 = threadIdx.x + SB_NW * 64;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint4* p = reinterpret_cast<const uint4*>(src[k]);
    t.r0[k] = p[threadIdx.x];
    t.r1[k] = p[i1 < CH ? i1 : 0];
  }
  return t;
}
template <int N>
ADT_DEVICE_INLINE void sb_img_store(__bf16* wimg, const SbImgRegs<N>& t) {
  constexpr int CH = TT_WIMG * 2 / 16;
  const int i1 = threadIdx.x + SB_NW * 64;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    uint4* dst = reinterpret_cast<uint4*>(wimg + k * TT_WIMG);
    dst[threadIdx.x] = t.r0[k];
    if (i1 < CH) dst[i1] = t.r1[k];
  }
}

// Key tile of slot s of wave w in pass B.  Pass A costs a query tile t its (t + 1) / 2 key-tile pairs, pass B costs a key tile t its
// (13 - t) / 2 query-tile pairs: with ONE tile map (tq_tile) the SIMD with four tiles carried 18 of pass B's 49 pairs per head against
// 10-11 on the others (pass A: 12-13 everywhere).  Pass B therefore has its own map, 13 / 12 / 12 / 12 pairs per SIMD -- its k / v operands
// are read back from the K / V images (sab_rowfrag: the same bf16 values the own-tile registers held), and P5 reads dk / dv of its own
// tile back from the token images P4 wrote (the same bf16 values tt_bfrags made of the registers).
ADT_DEVICE_INLINE int sb_tile_b(int s, int w, int ntiles) {
  if (ntiles != 13) return tq_tile(s, w, ntiles);
  const int t0 = (0x0 << 0) | (0x1 << 4) | (0x2 << 8) | (0x3 << 12) | (0x7 << 16) | (0x5 << 20) | (0x6 << 24) | (0x4 << 28);      // slot 0 of waves 0 .. 7
  const int t1 = (0xC << 12) | (0x8 << 16) | (0x9 << 20) | (0xA << 24) | (0xB << 28);                                             // slot 1 of waves 3 .. 7
  if (s == 0) return (int)(((unsigned)t0 >> (4 * w)) & 15u);
  return w >= 3 ? (int)(((unsigned)t1 >> (4 * w)) & 15u) : -1;
}

#define SB_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[w * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

// DEC = false: encoder block (q from LN(x), k / v from the raw x; the residual gradient joins at the LayerNorm output);
// DEC = true : decoder self-attention block (q, k, v from LN(x); the masked layer-output gradient joins at the LayerNorm output).
template <int HD, int MODE, bool DEC>
__global__ __launch_bounds__(SB_NW * 64) void k_seqtt_attn_pre_bwd(SeqBwdArgs a) {
  adt_prefetch_kernargs<(sizeof(SeqBwdArgs) + 63) / 64 * 64 <= 512 ? sizeof(SeqBwdArgs) : 512>();      // adt_common.cuh
  constexpr int H = 64 / HD, NT = HD / 16, KB = (HD + 31) / 32, NF = H * KB, NW = SB_NW, R = SB_R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef SeqBwdLds<H> Lds;
  __bf16* wtr = reinterpret_cast<__bf16*>(smem_raw);                        // Wq, Wk, Wv transposed (P5)
  __bf16* wpl = reinterpret_cast<__bf16*>(smem_raw + Lds::poff);             // Wq, Wk, Wv plain (P1) ; behind P1: the third token image
  __bf16* img2 = wpl;
  __bf16* img0 = reinterpret_cast<__bf16*>(smem_raw + Lds::wbytes / 2);
  __bf16* img1 = reinterpret_cast<__bf16*>(smem_raw + Lds::wbytes / 2 + Lds::ibytes);
  uint32_t* sM = reinterpret_cast<uint32_t*>(smem_raw + Lds::wbytes / 2 + 2 * Lds::ibytes + 64);
  float* sLse = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(sM) + Lds::mbytes);
  float* sDelta = sLse + H * R;
  float* sRed = sDelta + H * R;                                              // dgamma, dbeta, dbin[192] of this workgroup
  float* sVec = sRed + 320;                                                  // gamma, beta, packed in-projection bias (LDS copies)
  const float *vgamma = sVec, *vbeta = sVec + 64, *vbin = sVec + 128;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L, ntiles = (L + 15) / 16, npair = (L + 31) / 32;
  // Two workgroups per sequence (a.nsplit == 2: small batches, adt_seq_args.h): workgroup `part` OWNS the tiles t % 2 == part -- pass A for
  // them as query tiles, pass B for them as key tiles (a tile costs t + 1 there and 13 - t here: every wave carries the same sum), their
  // rows of the weight-gradient products and of the input gradient -- and recomputes q / k / v / delta of ALL tiles, which both passes
  // read (slot 0 of wave w: own tile wpart + 2 w ; slot 1: the other workgroup's tile 1 - wpart + 2 w, operands only).  The G images of
  // the weight-gradient products hold zeros in the rows of the other workgroup's tiles.
  const bool split = a.nsplit == 2;
  const int b = split ? blockIdx.x >> 1 : blockIdx.x, wpart = split ? (blockIdx.x & 1) : 0;
  auto TM = [&](int s) {      // tile whose operands slot s of this wave prepares (P1 / P3 / the X images)
    if (!split) return tq_tile(s, w, ntiles);
    const int t = (s == 0 ? wpart : 1 - wpart) + 2 * w;
    return t < ntiles ? t : -1;
  };
  auto TA = [&](int s) { return (split && s == 1) ? -1 : TM(s); };                                     // ... it runs pass A / P5 for
  auto TB = [&](int s) { return split ? TA(s) : sb_tile_b(s, w, ntiles); };                            // ... it runs pass B for
  if (a.nrep > 1) {
    const size_t off = (size_t)(blockIdx.x % a.nrep) * a.rep_stride;
    a.dWin += off; a.dbin += off; a.dgamma += off; a.dbeta += off;
  }
  float* const part = a.part ? a.part + (size_t)blockIdx.x * a.part_stride : nullptr;      // three consecutive blocks: Wq, Wk, Wv
  SB_STAMP(0);
  // The weight images are requested first and P1's activations right behind them: every workgroup of the launch runs the same phase at
  // the same time, so HBM streams the activations while the prologue runs instead of idling through it (adt_seqpost_tt.cuh)
  // Vector-memory results return in issue order, so the SMALL tables (keep bits, log-sum-exp, LayerNorm / bias vectors) are requested
  // first: their LDS stores then wait for nothing but themselves, while the weight images and the activations stream behind them.
  // (Requested after the big loads, their stores waited for every byte in front of them: the prologue took 21.7k cycles.)
  constexpr int MI = (H * R * 2 + NW * 64 - 1) / (NW * 64), LI = (H * R + NW * 64 - 1) / (NW * 64);
  tt_u4 mreg[MI];
  float lreg[LI];
  // every load of the prologue is UNCONDITIONAL (lanes without an element read a block of zeros / a clamped index): `if (valid) x = *p`
  // compiles to a branch with the wait for the load inside it, and six such loads were six serial memory round trips before the first
  // activation load was even issued (12k of the 18.6k prologue cycles, profiles/r03_stamps_attn_pre_bwd.txt)
  typedef const tt_u4 __attribute__((address_space(1))) * gu4;
  typedef const float __attribute__((address_space(1))) * gf1;
  if constexpr (MODE == 1) {
    adt_static_for<MI>([&](auto k) {
      const int i = threadIdx.x + k * NW * 64, hr = i >> 1, h = hr / R, r = hr - h * R;
      const bool ok = i < H * R * 2 && r < L;
      const gu4 p = ok ? (gu4)(a.mask + ((size_t)(b * H + h) * L + r) * 8) + (i & 1) : (gu4)tt_zero_row;
      mreg[k] = *p;
    });
  }
  adt_static_for<LI>([&](auto k) {
    const int i = threadIdx.x + k * NW * 64, h = i / R, r = i - h * R;
    const bool ok = i < H * R && r < L;
    const float v = *(ok ? (gf1)(a.lse + (size_t)(b * H + h) * L + r) : (gf1)tt_zero_row);
    lreg[k] = ok ? v * 1.4426950408889634f : INFINITY;
  });
  float vreg = 0.f;
  {
    const int t = threadIdx.x;
    const gf1 p = t < 64 ? (gf1)(a.gamma + t) : t < 128 ? (gf1)(a.beta + (t - 64)) : t < 320 ? (gf1)(a.bin + (t - 128)) : (gf1)tt_zero_row;
    vreg = *p;
  }
  const __bf16* wsrc = reinterpret_cast<const __bf16*>(a.wp_img) + 6 * (a.Win - a.wp_base);
  const __bf16* const src6[6] = {wsrc + 2 * WPACK_IMG, wsrc + 6 * 4096 + 2 * WPACK_IMG, wsrc + 12 * 4096 + 2 * WPACK_IMG,
                                 wsrc + 3 * WPACK_IMG, wsrc + 6 * 4096 + 3 * WPACK_IMG, wsrc + 12 * 4096 + 3 * WPACK_IMG};
  // Weight images (9 KiB each, plain copies of the pre-packed images) by LDS-DMA: no staging registers (the register-staged form held 48
  // VGPRs through the prologue and forced the activation loads into their branching form).  The prologue is a BANDWIDTH burst -- all 256
  // workgroups pull their inputs at once, ~11 B per cycle and CU (MI355X_MICROARCH.md; 200 KB per workgroup took the 18.6k cycles of
  // profiles/r03_stamps_attn_pre_bwd.txt) -- so only what phase P1 needs is requested in front of the first barrier: the small tables, the
  // three plain images and x.  dO, O and the three transposed images (needed from the end of P1 / in P5) are requested behind the barrier
  // and stream while P1 computes.
#pragma unroll
  for (int k = 0; k < 3; ++k) adt_glds_block<NW>(src6[k], wpl + k * TT_WIMG, TT_WIMG * 2);
  TT xa[2], doa[2], oa[2];
  int idv[2] = {1, 1};                           // decoder block: the pad mask of the residual-path gradient in P5 (read there it was an exposed round trip)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TM(s);
    const int l = tile * 16 + c, row = b * L + l;
    xa[s] = tt_load(a.x + (size_t)row * 64, tile >= 0 && l < L, g);               // unconditional (address-select) loads: all in flight together
    if (DEC) idv[s] = tt_load_id(a.ids, row, tile >= 0 && l < L);
  }
  SB_STAMP(11);
  // ---- P0: zeroed token images, then the tables and the weight images land in LDS ------------------------------------------------
  {
    uint4* z = reinterpret_cast<uint4*>(img0);
    for (int i = threadIdx.x; i < (int)((2 * Lds::ibytes + 64) / 16); i += NW * 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x < 320) sRed[threadIdx.x] = 0.f;
    SB_STAMP(12);
    if constexpr (MODE == 1) {
      adt_static_for<MI>([&](auto k) {
        const int i = threadIdx.x + k * NW * 64;
        if (i < H * R * 2) reinterpret_cast<tt_u4*>(sM + (size_t)(i >> 1) * 8)[i & 1] = mreg[k];
      });
    }
    adt_static_for<LI>([&](auto k) {
      const int i = threadIdx.x + k * NW * 64;
      if (i < H * R) { sLse[i] = lreg[k]; sDelta[i] = 0.f; }
    });
    if (threadIdx.x < 320) sVec[threadIdx.x] = vreg;
    SB_STAMP(13);
    adt_wait_vm0();          // the LDS-DMA of the weight images (and everything requested before it) has landed
    SB_STAMP(14);
  }
  __syncthreads();
  SB_STAMP(1);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TM(s);
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = tile >= 0 && l < L;
    doa[s] = tt_load(a.dO + (size_t)row * 64, valid, g);
    oa[s] = tt_load_saved(a.o, row, valid, g, a.saved_bf16);
  }
#pragma unroll
  for (int k = 3; k < 6; ++k) adt_glds_block<NW>(src6[k], wtr + (k - 3) * TT_WIMG, TT_WIMG * 2);      // waited for in front of the barrier behind pass A
  const float qmul = a.scale * 1.4426950408889634f;
  const uint32_t key_rng = drop_key(a.drop);
  // ---- P1: recompute LN + in-projection; operands to registers, K / V images ------------------------------------------------------
  bf16x8 fq[2][NF], fdo[2][NF];
  float delta[2][H];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TM(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c;
    const bool valid = l < L;
    const TT& x = xa[s];
    const TT xn = tt_ln_apply(tt_ln_stats(x, a.ln_eps).xhat, vgamma, vbeta, g);
    const TTB bn = tt_bfrags(xn);
    TTB bx;
    if (!DEC) bx = tt_bfrags(x);
    TT q = tt_gemm(bn, wpl, c, g);
    tt_add_vec(q, vbin, g);
    sb_frags<HD>(q, qmul, fq[s]);
    TT k = tt_gemm(DEC ? bn : bx, wpl + TT_WIMG, c, g);
    tt_add_vec(k, vbin + 64, g);
    tt_put_rows(img0, l, k, valid, g);
    TT v = tt_gemm(DEC ? bn : bx, wpl + 2 * TT_WIMG, c, g);
    tt_add_vec(v, vbin + 128, g);
    tt_put_rows(img1, l, v, valid, g);
    const TT& dO = doa[s];
    const TT& o = oa[s];
    sb_frags<HD>(dO, a.drop.scale, fdo[s]);          // the dO operands / image carry 1 / (1 - p) (adt_seqattn.cuh: sab_pass_a); delta below does not
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float part = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part += dO.v[h * NT + nt][r] * o.v[h * NT + nt][r];
      delta[s][h] = tt_colsum(part);
      if (g == 0 && valid) sDelta[h * R + l] = delta[s][h];
    }
  }
  __syncthreads();
  SB_STAMP(2);
  {                                              // the plain weight images are dead: their place becomes the third token image of P4 (rows beyond L stay zero)
    uint4* z = reinterpret_cast<uint4*>(img2);
    for (int i = threadIdx.x; i < (int)((Lds::ibytes + 64) / 16); i += NW * 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  // ---- P2: pass A (dQ): keys / values from the images, own query and dO rows from registers ----------------------------------------
  TT dq[2], dk[2], dv[2];
  dq[1] = tt_zero();                             // (two workgroups per sequence: slot 1 is not this workgroup's tile -- zero rows in the dq image of P4)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TA(s);
    if (tile < 0) continue;
    const int q = tile * 16 + c;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      f32x4 t[NT];
      const uint32_t bh_rng = (uint32_t)(b * H + h) + a.b_offset * (uint32_t)H;
      sab_pass_a<HD, MODE>(img0, img1, fq[s] + h * KB, fdo[s] + h * KB, sLse[h * R + q], delta[s][h], sM + ((size_t)h * R + q) * 8, tile, h, a.drop,
                           key_rng, (bh_rng * (uint32_t)L + (uint32_t)q) * (uint32_t)L, c, g, t);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dq[s].v[h * NT + nt] = t[nt] * a.scale;
    }
  }
  // pass B's own key / value rows (its tile map differs from pass A's: sb_tile_b), from the K / V images while they are still there
  bf16x8 fk[2][NF], fv[2][NF];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TB(s);
    const int row = (tile >= 0 ? tile : 0) * 16 + c;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        fk[s][h * KB + kb] = sab_rowfrag<HD>(img0, row, h, kb, g);
        fv[s][h * KB + kb] = sab_rowfrag<HD>(img1, row, h, kb, g);
      }
  }
  adt_wait_vm0();            // the transposed weight images (LDS-DMA issued at the start of P1) have landed: published by this barrier
  __syncthreads();
  SB_STAMP(3);
  // ---- P3: Q / dO images, then pass B (dK, dV) with own key and value rows from registers ----------------------------------------------
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TM(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c;
    sb_put_frags<HD>(img0, l, fq[s], l < L, g);
    sb_put_frags<HD>(img1, l, fdo[s], l < L, g);
  }
  __syncthreads();
  TT xk[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s == 1) {
      // x is read once more for P4 and P5 (its registers were needed by the attention passes); requested here, behind slot 0 of pass B
      // (whose operand registers are free by now), the rows arrive while slot 1 is swept instead of stalling P4 (3.4-6.2k cycles)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int tile2 = TM(s2);
        const int l2 = tile2 * 16 + c;
        xk[s2] = tt_load(a.x + (size_t)(b * L + l2) * 64, tile2 >= 0 && l2 < L, g);
      }
    }
    const int tile = TB(s);
    if (tile < 0) continue;
    {
      f32x4 tk[4], tv[4];
      sab_pass_b_heads<HD, MODE>(img0, img1, fk[s], fv[s], sLse, sDelta, sM, R, tile, ntiles, a.drop, key_rng,
                                 (uint32_t)(b * H) + a.b_offset * (uint32_t)H, L, c, g, tk, tv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dk[s].v[i] = tk[i] * 0.6931471805599453f;        // the Q image carries log2(e) / sqrt(hd)
        dv[s].v[i] = tv[i];
      }
    }
  }
  __syncthreads();
  SB_STAMP(4);
  // ---- P4: in-projection weight / bias gradients: three products over all tokens, in two rounds on three token images ------------------
  // round 1: G = dq (img0), dk (img2), X = LN(x) (img1): dWq -- and dWk in the decoder block, whose k reads LN(x) too
  // round 2: G = dv (img0), X = the raw x in the encoder block (img1): dWv -- and the encoder's dWk = dk^T x
  // (three rounds on two images were five barriers; dk waits in the place of the plain weight images, dead since P1)
  {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int tile = TM(s);
      if (tile < 0) continue;
      const int l = tile * 16 + c;
      const bool valid = l < L;
      tt_put_rows(img0, l, dq[s], valid, g);
      tt_put_rows(img1, l, tt_ln_apply(tt_ln_stats(xk[s], a.ln_eps).xhat, vgamma, vbeta, g), valid, g);     // q always reads LN(x)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {                // dk of pass B's tiles
      const int tile = TB(s);
      if (tile < 0) continue;
      const int l = tile * 16 + c;
      tt_put_rows(img2, l, dk[s], l < L, g);
    }
    SB_STAMP(7);
  }
  SB_STAMP(8);
  __syncthreads();
  // P5's inputs are requested here, in front of the products (as loads inside P5 each was a fully exposed round trip: two waves
  // per SIMD do not hide one): the residual-path gradient of both slots now, the accumulated gx one slot ahead of its use
  TT resa[2], gxo[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TA(s);
    const int l = tile * 16 + c;
    resa[s] = tt_load(a.dres + (size_t)(b * L + l) * 64, tile >= 0 && l < L, g);
  }
  sb_dw_product16(img0, img1, npair, a.dWin, part, sRed + 128, w, c, g);
  if (DEC) sb_dw_product16(img2, img1, npair, a.dWin + 4096, part ? part + 4096 : nullptr, sRed + 192, w, c, g);
  SB_STAMP(9);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TM(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c;
    const bool valid = l < L;
    if (!DEC) tt_put_rows(img1, l, xk[s], valid, g);                  // encoder: k, v read the raw x
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {                  // dv of pass B's tiles
    const int tile = TB(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c;
    tt_put_rows(img0, l, dv[s], l < L, g);
  }
  __syncthreads();
  if (!DEC) sb_dw_product16(img2, img1, npair, a.dWin + 4096, part ? part + 4096 : nullptr, sRed + 192, w, c, g);
  sb_dw_product16(img0, img1, npair, a.dWin + 8192, part ? part + 8192 : nullptr, sRed + 256, w, c, g);
  SB_STAMP(5);
  // ---- P5: gradient of the block input ----------------------------------------------------------------------------------------
  TT dgm = tt_zero(), dbt = tt_zero();
  auto gx_request = [&](int s) {
    const int tile = TA(s);
    const int l = tile * 16 + c;
    const float* const src = a.seed_other ? a.seed_other : a.gx;
    gxo[s] = tt_load(src + (size_t)(b * L + l) * 64, (a.acc || a.seed_other) && tile >= 0 && l < L, g);
  };
  const float seed_coef = a.seed_other ? *a.seed_coef : 0.f;
  float seed_sq = 0.f;                             // this lane's share of sum (x - other)^2 over the workgroup's tokens
  gx_request(0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = TA(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    if (s == 0) gx_request(1);
    const TTLn st = tt_ln_stats(xk[s], a.ln_eps);
    TT dn = tt_gemm(tt_bfrags(dq[s]), wtr, c, g);
    // dk / dv of THIS tile were computed by pass B's owner of it: their bf16 rows are in the token images (img2: dk, img0: dv)
    TTB bdk, bdv;
    bdk.kb[0] = sab_rowfrag<32>(img2, l, 0, 0, g); bdk.kb[1] = sab_rowfrag<32>(img2, l, 1, 0, g);
    bdv.kb[0] = sab_rowfrag<32>(img0, l, 0, 0, g); bdv.kb[1] = sab_rowfrag<32>(img0, l, 1, 0, g);
    TT dkv = tt_gemm(bdk, wtr + TT_WIMG, c, g);
    tt_add(dkv, tt_gemm(bdv, wtr + 2 * TT_WIMG, c, g));
    TT res = resa[s];
    if (DEC && (!valid || idv[s] == 0)) res = tt_zero();
    if (DEC && a.dres_scale != 0.f) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) res.v[nt] *= a.dres_scale;
    }
    if (s == 0) SB_STAMP(10);
    tt_add(dn, res);
    TT dx;
    if (!DEC) {
      dx = tt_ln_bwd(dn, st, vgamma, dgm, dbt, g);
      tt_add(dx, dkv);                                  // k, v read the raw x (sasrec/modules.py:647)
    } else {
      tt_add(dn, dkv);
      dx = tt_ln_bwd(dn, st, vgamma, dgm, dbt, g);
    }
    if (valid) {
      float* dst = a.gx + (size_t)row * 64;
      if (a.seed_other) {                               // the reconstruction seed of this input row: coef * (x - the other stack's row)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f32x4 df = xk[s].v[nt] - gxo[s].v[nt];
          dx.v[nt] += seed_coef * df;
          seed_sq += df[0] * df[0] + df[1] * df[1] + df[2] * df[2] + df[3] * df[3];
        }
      } else {
        tt_add(dx, gxo[s]);                             // zeros unless a.acc
      }
      tt_store(dst, dx, true, g);
    }
    if (s == 0) SB_STAMP(15);
  }
  // LayerNorm gamma / beta gradients: per-lane partials over this wave's tokens -> this wave's sums in its own LDS row (the keep-bit area is
  // dead behind the attention passes) -> joined in wave order behind the barrier: no LDS float atomics, whose order is the waves' arrival order
  float* sWave = reinterpret_cast<float*>(sM) + w * 128;
  if (a.seed_loss) {                               // (workgroup-uniform) per-wave sums behind the LayerNorm sums' rows, joined below
    const float ws_ = wave_sum(seed_sq);
    if (lane == 0) reinterpret_cast<float*>(sM)[NW * 128 + w] = ws_;
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = tt_rowsum16(dgm.v[nt][r]), sb = tt_rowsum16(dbt.v[nt][r]);      // over the 16 tokens on this row of lanes
      if (c == 0) { sWave[16 * nt + 4 * g + r] = sg; sWave[64 + 16 * nt + 4 * g + r] = sb; }
    }
  __syncthreads();
  {
    const int t = threadIdx.x;
    if (t < 128) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) sum += reinterpret_cast<const float*>(sM)[k * 128 + t];
      sRed[t] = sum;
    }
    if (a.seed_loss && t == 320) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) sum += reinterpret_cast<const float*>(sM)[NW * 128 + k];
      atomicAdd(a.seed_loss + (blockIdx.x & 63), sum / a.seed_norms[1]);
    }
    if (a.vpart) {
      if (t < 320) a.vpart[(size_t)blockIdx.x * 512 + t] = sRed[t];
    } else if (t < 64) atomicAdd(a.dgamma + t, sRed[t]);
    else if (t < 128) atomicAdd(a.dbeta + t - 64, sRed[t]);
    else if (t < 320) atomicAdd(a.dbin + t - 128, sRed[t]);
  }
  SB_STAMP(6);
}

}  // namespace adt
