// STOSA-ADT Wasserstein attention on the matrix cores (stosa/modules.py:30-43 wasserstein_distance_matmul, :222-275 DistAttention,
// :311-361 DistEDAttention).  The pair score
//   x_ij = -(|mq_i|^2 + sum(Sq_i) + |mk_j|^2 + sum(Sk_j) - 2 (mq_i . mk_j + sqrt(Sq_i) . sqrt(Sk_j))) / sqrt(hd)
// has ONE cross term with inner dimension 2 hd over the concatenation [m | sqrt(S)]: at the Beauty template (hd = 16) exactly one
// v_mfma_f32_16x16x32_bf16 per 16 x 16 score tile (exact mode: eight v_mfma_f32_16x16x4_f32).  The contexts P~ Vm and (P~ * P~) Vc, the
// gradient products dP~ = dOm Vm^T + 2 P~ (dOc Vc^T), dQ = -2 dW [mk | sk], dK, dVm, dVc are MFMA products too; only the softmax
// statistics, the additive mask and the dropout hash stay on the vector ALUs.  k_wattn_fwd / k_wattn_bwd of adt_stosa.cuh (one wave
// per query row, one key per lane, all VALU) remain for the shapes this file does not cover (hd = 64 or L > 128).
//
// One workgroup per (sequence, head), four waves, a wave owns 16-query tiles (forward, backward pass A) or 16-key tiles (pass B).
// LDS holds fp32 row images (stride K + 4: conflict-free 8-float fragments, adt_common.cuh) and, for the products that contract over
// the resident index, transposed images read in MFMA slot order (frag_slotc).  Every score tile of the three passes is produced by
// the same instruction sequence on the same operand values, key on the accumulator row in the forward and in pass A, query on the
// accumulator row in pass B (operands swapped).  The reference's additive -2^32 mask quantises the scores of a fully masked row to
// multiples of 512 (adt_stosa.cuh: w_score): bit-level agreement of the recomputed scores matters only within one ulp of +-256,
// +-768, ..., i.e. for squared distances >= 256 sqrt(hd), and the dot product of an MFMA does not depend on which operand is A.
#pragma once
#include "adt_stosa.cuh"

namespace adt {

constexpr int WM_NW = 4;

template <int HD>
struct WmShape {
  static constexpr int KC = 2 * HD;            // concatenated [mean | sqrt cov] width
  static constexpr int KB = KC / 32;           // 32-slot contraction blocks of the cross term
  static constexpr int NT = HD / 16;           // 16-feature tiles of a context
  static constexpr int RSK = KC + 4, RSV = HD + 4;
};

static inline size_t wattn_mfma_lds_bytes(int L, int hd, bool bwd) {
  const int Lp = (L + 31) / 32 * 32, KC = 2 * hd;
  size_t f = (size_t)Lp * (KC + 4) + (size_t)4 * Lp;                       // concat image, norms, validity, dead, lse / spare
  if (!bwd) f += (size_t)2 * hd * (Lp + 4);                                // Vm^T, Vc^T
  else f += (size_t)KC * (Lp + 4) + (size_t)2 * Lp * (hd + 4) + (size_t)2 * hd * (Lp + 4) + Lp;   // + concat^T, two natural images, two transposed, delta
  return f * sizeof(float);
}

// rows [0, Lp) of the concatenated image [m | sqrt(clamp(S))] (rows >= L zero) and the row norms |m|^2 + sum(S)
template <int HD>
ADT_DEVICE_INLINE void wm_stage_cat(float* sCat, float* sNorm, const float* gm, int ldm, const float* gc, int ldc, int L, int Lp) {
  constexpr int RS = WmShape<HD>::RSK, V = HD / 4;
  for (int i = threadIdx.x; i < Lp * V; i += WM_NW * 64) {
    const int r = i / V, c4 = (i % V) * 4;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), s = m;
    if (r < L) {
      m = *reinterpret_cast<const float4*>(gm + (size_t)r * ldm + c4);
      const float4 cv = *reinterpret_cast<const float4*>(gc + (size_t)r * ldc + c4);
      s = make_float4(w_sqrt_cov(cv.x), w_sqrt_cov(cv.y), w_sqrt_cov(cv.z), w_sqrt_cov(cv.w));
    }
    *reinterpret_cast<float4*>(sCat + r * RS + c4) = m;
    *reinterpret_cast<float4*>(sCat + r * RS + HD + c4) = s;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < Lp; r += WM_NW * 64) {
    float n = 0.f;
    for (int k = 0; k < 2 * HD; ++k) n = fmaf(sCat[r * RS + k], sCat[r * RS + k], n);
    sNorm[r] = n;
  }
}

// transposed image dst[col][row] (row stride Lp + 4) of a (L x W) global slice, optionally through sqrt(clamp(.)); rows >= L zero
template <bool SQRT>
ADT_DEVICE_INLINE void wm_stage_t(float* dst, const float* g, int ld, int L, int Lp, int W) {
  const int RS = Lp + 4, V = W / 4;
  for (int i = threadIdx.x; i < Lp * V; i += WM_NW * 64) {          // 16-byte global reads along the features, four LDS words out
    const int r = i / V, c4 = (i % V) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) {
      v = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c4);
      if (SQRT) v = make_float4(w_sqrt_cov(v.x), w_sqrt_cov(v.y), w_sqrt_cov(v.z), w_sqrt_cov(v.w));
    }
    dst[(c4 + 0) * RS + r] = v.x; dst[(c4 + 1) * RS + r] = v.y; dst[(c4 + 2) * RS + r] = v.z; dst[(c4 + 3) * RS + r] = v.w;
  }
}

template <int HD>
ADT_DEVICE_INLINE void wm_stage_nat(float* dst, const float* g, int ld, int L, int Lp) {
  constexpr int RS = WmShape<HD>::RSV, V = HD / 4;
  for (int i = threadIdx.x; i < Lp * V; i += WM_NW * 64) {
    const int r = i / V, c4 = (i % V) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) v = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c4);
    *reinterpret_cast<float4*>(dst + r * RS + c4) = v;
  }
}

// |m|^2 + sum(S) of one token, by the SAME sequence of fused multiply-adds as wm_stage_cat's LDS loop, so that the norm of a token is the
// same number whether a pass reads it from the resident side's vector or computes it for the tile it owns
template <int HD>
ADT_DEVICE_INLINE float wm_row_norm(const float* m_row, const float* c_row) {
  float n = 0.f;
  for (int k = 0; k < HD; ++k) n = fmaf(m_row[k], m_row[k], n);
  for (int k = 0; k < HD; ++k) { const float sv = w_sqrt_cov(c_row[k]); n = fmaf(sv, sv, n); }
  return n;
}

ADT_DEVICE_INLINE float wm_colsum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
ADT_DEVICE_INLINE float wm_colmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }

// the concatenated operand of one token straight from global memory: slots 8g .. 8g+7 of block kb (mean below HD, sqrt cov above)
template <int HD>
ADT_DEVICE_INLINE Frag8 wm_cat_frag(const float* m_row, const float* c_row, int kb, int g, bool valid) {
  Frag8 f;
  const int k0 = 32 * kb + 8 * g;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
  if (!valid) return f;
  if (k0 < HD) return frag_contig(m_row + k0);
  f = frag_contig(c_row + (k0 - HD));
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = w_sqrt_cov(f.v[j]);
  return f;
}

// B operand of a product that contracts over a 32-row block of an accumulator-resident tile pair (slot (g, j) <-> row 4g + (j & 3) of
// tile j >> 2)
ADT_DEVICE_INLINE Frag8 wm_acc_frag(const f32x4& lo, const f32x4& hi) {
  Frag8 f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { f.v[r] = lo[r]; f.v[4 + r] = hi[r]; }
  return f;
}

ADT_DEVICE_INLINE float wm_keep_scale(const DropCfg& d, uint32_t key, uint32_t idx) {
  if (!d.thr) return 1.0f;
  return adt_keep(key, idx, d.thr) ? d.scale : 0.f;
}
// keep scales of the four consecutive elements idx .. idx + 3 (an accumulator register quad): one hash when idx % 4 == 0
ADT_DEVICE_INLINE f32x4 wm_keep_scale4(const DropCfg& d, uint32_t key, uint32_t idx) {
  f32x4 k = {1.f, 1.f, 1.f, 1.f};
  if (!d.thr) return k;
  if ((idx & 3u) == 0u) {
    const uint32_t bits = adt_keep4(key, idx, d.thr);
#pragma unroll
    for (int r = 0; r < 4; ++r) k[r] = ((bits >> r) & 1u) ? d.scale : 0.f;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) k[r] = adt_keep(key, idx + (uint32_t)r, d.thr) ? d.scale : 0.f;
  }
  return k;
}
// exp(score - reference) of one pair: a masked key of a row that has attendable keys contributes exactly 0 (x - 2^32 underflows),
// so neither the score rounding nor the exponential is evaluated for it
ADT_DEVICE_INLINE float wm_prob(float x, bool masked, bool dead_row, float ref) {
  if (masked && !dead_row) return 0.f;
  return __expf(w_score(x, masked, dead_row) - ref);
}

template <int PREC, int HD, int MAXKT>
__global__ __launch_bounds__(WM_NW * 64) void k_wattn_mfma_fwd(WAttnArgs a) {
  adt_prefetch_kernargs<sizeof(WAttnArgs) <= 512 ? sizeof(WAttnArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef WmShape<HD> S;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, Lp = (L + 31) / 32 * 32, nt16 = (L + 15) / 16, RST = Lp + 4;
  float* sK = smem;                          // [Lp][RSK]  [mk | sqrt Sk]
  float* sNk = sK + Lp * S::RSK;             // [Lp]
  float* sKv = sNk + Lp;                     // key validity
  float* sDead = sKv + Lp;
  float* sSpare = sDead + Lp;
  float* sVmT = sSpare + Lp;                 // [HD][RST]
  float* sVcT = sVmT + HD * RST;
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t row_b = (size_t)b * L;
  for (int i = threadIdx.x; i < Lp; i += WM_NW * 64) sKv[i] = (i < L && a.kid[row_b + i] > 0) ? 1.f : 0.f;
  wm_stage_t<false>(sVmT, a.Vm + row_b * a.ldvm + h * HD, a.ldvm, L, Lp, HD);
  wm_stage_t<false>(sVcT, a.Vc + row_b * a.ldvc + h * HD, a.ldvc, L, Lp, HD);
  wm_stage_cat<HD>(sK, sNk, a.Km + row_b * a.ldkm + h * HD, a.ldkm, a.Kc + row_b * a.ldkc + h * HD, a.ldkc, L, Lp);
  __syncthreads();
  for (int i = threadIdx.x; i < Lp; i += WM_NW * 64) {
    float cnt = 0.f;
    for (int j = 0; j <= i && j < L; ++j) cnt += sKv[j];
    sDead[i] = cnt == 0.f ? 1.f : 0.f;
  }
  __syncthreads();
  const uint32_t key_rng = drop_key(a.drop);
  const float rsq_hd = 1.0f / sqrtf((float)HD);
  for (int qt = w; qt < nt16; qt += WM_NW) {
    const int i = qt * 16 + c;
    const bool vi = i < L;
    const float* qm_row = a.Qm + (row_b + (vi ? i : 0)) * a.ldqm + h * HD;
    const float* qc_row = a.Qc + (row_b + (vi ? i : 0)) * a.ldqc + h * HD;
    Frag8 fq[S::KB];
#pragma unroll
    for (int kb = 0; kb < S::KB; ++kb) fq[kb] = wm_cat_frag<HD>(qm_row, qc_row, kb, g, vi);
    const float nq = vi ? wm_row_norm<HD>(qm_row, qc_row) : 0.f;
    const bool dead_i = vi && sDead[i] != 0.f;
    f32x4 s[MAXKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (kt < nt16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < S::KB; ++kb) acc = mma16<PREC>(acc, frag_contig(sK + (kt * 16 + c) * S::RSK + 32 * kb + 8 * g), fq[kb]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt * 16 + 4 * g + r;
          if (j < L) {
            const float x = -((sNk[j] + nq) - 2.0f * acc[r]) * rsq_hd;
            const bool masked = j > i || sKv[j] == 0.f;
            if (!masked || dead_i) {           // masked keys of a live row: exp(x - 2^32 - m) == 0, left at -inf
              s[kt][r] = w_score(x, masked, dead_i);
              m = fmaxf(m, s[kt][r]);
            }
          }
        }
      }
    }
    m = wm_colmax(m);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (kt < nt16 && s[kt][r] != -INFINITY) ? __expf(s[kt][r] - m) : 0.f;
        s[kt][r] = e;
        sum += e;
      }
    sum = wm_colsum(sum);
    const float inv = 1.0f / sum;
    if (g == 0 && vi) a.LSE[(size_t)bh * L + i] = m + logf(sum);
    const uint32_t idx_q = ((uint32_t)(bh + a.bh_offset) * (uint32_t)L + (uint32_t)i) * (uint32_t)L;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
      const f32x4 ks4 = wm_keep_scale4(a.drop, key_rng, idx_q + (uint32_t)(kt * 16 + 4 * g));
      s[kt] = s[kt] * inv * ks4;
    }
    // contexts: O^T[feature][query] = sum_key V^T[feature][key] P~^T[key][query]
    f32x4 om[S::NT], oc[S::NT];
#pragma unroll
    for (int nt = 0; nt < S::NT; ++nt) { om[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; oc[nt] = om[nt]; }
#pragma unroll
    for (int kb = 0; kb < MAXKT / 2; ++kb) {
      if (2 * kb < nt16) {
        const Frag8 bp = wm_acc_frag(s[2 * kb], s[2 * kb + 1]);
        Frag8 bp2;
#pragma unroll
        for (int j = 0; j < 8; ++j) bp2.v[j] = bp.v[j] * bp.v[j];
#pragma unroll
        for (int nt = 0; nt < S::NT; ++nt) {
          om[nt] = mma16<PREC>(om[nt], frag_slotc(sVmT + (16 * nt + c) * RST + 32 * kb, g), bp);
          oc[nt] = mma16<PREC>(oc[nt], frag_slotc(sVcT + (16 * nt + c) * RST + 32 * kb, g), bp2);
        }
      }
    }
    if (vi) {
#pragma unroll
      for (int nt = 0; nt < S::NT; ++nt) {
        *reinterpret_cast<float4*>(a.Om + (row_b + i) * a.ldom + h * HD + 16 * nt + 4 * g) = make_float4(om[nt][0], om[nt][1], om[nt][2], om[nt][3]);
        *reinterpret_cast<float4*>(a.Oc + (row_b + i) * a.ldoc + h * HD + 16 * nt + 4 * g) = make_float4(oc[nt][0], oc[nt][1], oc[nt][2], oc[nt][3]);
      }
    }
  }
}

// Backward.  Pass A (key side resident, wave owns a query tile) -> dQm, dQc ; pass B (query side resident, wave owns a key tile)
// -> dKm, dKc, dVm, dVc.  Every gradient element is produced by exactly one wave: no atomics, reproducible.
// delta_i = sum_j P_ij dP_ij needs no sweep of its own: with P~ = P * keep, sum_j P_ij dP_ij = dOm_i . Om_i + 2 dOc_i . Oc_i (the saved
// contexts), so each pass makes ONE sweep over the other side in pairs of 16-row tiles, turns each pair of dW / P~ tiles straight into
// MFMA operands and keeps only the output accumulators (the first version held every tile of a sweep: 299 registers, one wave per SIMD).
template <int HD>
ADT_DEVICE_INLINE void wm_pair_bwd(float x, bool masked, bool dead_row, float lse, float ks_, float gm, float gc, float delta, float rsq_hd,
                                   float& dw, float& pd) {
  const float pr = wm_prob(x, masked, dead_row, lse);
  pd = pr * ks_;
  const float dpr = (gm + 2.0f * pd * gc) * ks_;      // d loss / d P_ij
  dw = -(pr * (dpr - delta)) * rsq_hd;                // d loss / d W_ij
}

template <int PREC, int HD>
__global__ __launch_bounds__(WM_NW * 64) void k_wattn_mfma_bwd(WAttnArgs a) {
  adt_prefetch_kernargs<sizeof(WAttnArgs) <= 512 ? sizeof(WAttnArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef WmShape<HD> S;
  constexpr int KBV = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, Lp = (L + 31) / 32 * 32, nt16 = (L + 15) / 16, npair = (nt16 + 1) / 2, RST = Lp + 4;
  float* sC = smem;                          // A: [mk | sqrt Sk] rows   | B: [mq | sqrt Sq] rows
  float* sN = sC + Lp * S::RSK;              // A: key norms            | B: query norms
  float* sKv = sN + Lp;
  float* sDead = sKv + Lp;
  float* sLse = sDead + Lp;
  float* sCT = sLse + Lp;                    // [KC][RST] transposed concat image
  float* sX0 = sCT + S::KC * RST;            // A: Vm rows  | B: dOm rows      [Lp][RSV]
  float* sX1 = sX0 + Lp * S::RSV;            // A: Vc rows  | B: dOc rows
  float* sT0 = sX1 + Lp * S::RSV;            // B only: dOm^T [HD][RST]
  float* sT1 = sT0 + HD * RST;               // B only: dOc^T
  float* sDelta = sT1 + HD * RST;            // [Lp]
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t row_b = (size_t)b * L;
  const float rsq_hd = 1.0f / sqrtf((float)HD);
  const uint32_t key_rng = drop_key(a.drop);
  const uint32_t idx_bh = (uint32_t)(bh + a.bh_offset) * (uint32_t)L;
  for (int i = threadIdx.x; i < Lp; i += WM_NW * 64) {
    sKv[i] = (i < L && a.kid[row_b + i] > 0) ? 1.f : 0.f;
    sLse[i] = i < L ? a.LSE[(size_t)bh * L + i] : INFINITY;
    float dl = 0.f;
    if (i < L) {
      const float* dm = a.dOm + (row_b + i) * a.lddom + h * HD; const float* om = a.Om + (row_b + i) * a.ldom + h * HD;
      const float* dc = a.dOc + (row_b + i) * a.lddoc + h * HD; const float* oc = a.Oc + (row_b + i) * a.ldoc + h * HD;
      for (int k = 0; k < HD; k += 4) {
        const float4 x0 = *reinterpret_cast<const float4*>(dm + k), y0 = *reinterpret_cast<const float4*>(om + k);
        const float4 x1 = *reinterpret_cast<const float4*>(dc + k), y1 = *reinterpret_cast<const float4*>(oc + k);
        dl += (x0.x * y0.x + x0.y * y0.y + x0.z * y0.z + x0.w * y0.w) + 2.0f * (x1.x * y1.x + x1.y * y1.y + x1.z * y1.z + x1.w * y1.w);
      }
    }
    sDelta[i] = dl;
  }
  wm_stage_t<false>(sCT, a.Km + row_b * a.ldkm + h * HD, a.ldkm, L, Lp, HD);
  wm_stage_t<true>(sCT + HD * RST, a.Kc + row_b * a.ldkc + h * HD, a.ldkc, L, Lp, HD);
  wm_stage_nat<HD>(sX0, a.Vm + row_b * a.ldvm + h * HD, a.ldvm, L, Lp);
  wm_stage_nat<HD>(sX1, a.Vc + row_b * a.ldvc + h * HD, a.ldvc, L, Lp);
  wm_stage_cat<HD>(sC, sN, a.Km + row_b * a.ldkm + h * HD, a.ldkm, a.Kc + row_b * a.ldkc + h * HD, a.ldkc, L, Lp);
  __syncthreads();
  for (int i = threadIdx.x; i < Lp; i += WM_NW * 64) {
    float cnt = 0.f;
    for (int j = 0; j <= i && j < L; ++j) cnt += sKv[j];
    sDead[i] = cnt == 0.f ? 1.f : 0.f;
  }
  __syncthreads();

  // ---- pass A: wave owns query tile qt ; tiles T[key 4g+r][query c] ----------------------------------------------------------------
  for (int qt = w; qt < nt16; qt += WM_NW) {
    const int i = qt * 16 + c;
    const bool vi = i < L;
    const size_t qrow = row_b + (vi ? i : 0);
    Frag8 fq[S::KB], fdm[KBV], fdc[KBV];
#pragma unroll
    for (int kb = 0; kb < S::KB; ++kb) fq[kb] = wm_cat_frag<HD>(a.Qm + qrow * a.ldqm + h * HD, a.Qc + qrow * a.ldqc + h * HD, kb, g, vi);
    const float nq = vi ? wm_row_norm<HD>(a.Qm + qrow * a.ldqm + h * HD, a.Qc + qrow * a.ldqc + h * HD) : 0.f;
#pragma unroll
    for (int kb = 0; kb < KBV; ++kb) {
      fdm[kb] = frag_contig_hd<HD>(a.dOm + qrow * a.lddom + h * HD, kb, g);
      fdc[kb] = frag_contig_hd<HD>(a.dOc + qrow * a.lddoc + h * HD, kb, g);
      if (!vi) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { fdm[kb].v[j] = 0.f; fdc[kb].v[j] = 0.f; }
      }
    }
    const float lse = vi ? sLse[i] : INFINITY, delta = vi ? sDelta[i] : 0.f;
    const bool dead_i = vi && sDead[i] != 0.f;
    const uint32_t idx_q = (idx_bh + (uint32_t)i) * (uint32_t)L;
    f32x4 accq[S::KC / 16];
#pragma unroll
    for (int ft = 0; ft < S::KC / 16; ++ft) accq[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dwsum = 0.f;
#pragma unroll 1
    for (int kp = 0; kp < npair; ++kp) {
      f32x4 dwt[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kt = 2 * kp + t;
        dwt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < nt16) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, gm = acc, gc = acc;
#pragma unroll
          for (int kb = 0; kb < S::KB; ++kb) acc = mma16<PREC>(acc, frag_contig(sC + (kt * 16 + c) * S::RSK + 32 * kb + 8 * g), fq[kb]);
#pragma unroll
          for (int kb = 0; kb < KBV; ++kb) {
            gm = mma16<PREC>(gm, frag_contig_hd<HD>(sX0 + (kt * 16 + c) * S::RSV, kb, g), fdm[kb]);
            gc = mma16<PREC>(gc, frag_contig_hd<HD>(sX1 + (kt * 16 + c) * S::RSV, kb, g), fdc[kb]);
          }
          const f32x4 ks4 = wm_keep_scale4(a.drop, key_rng, idx_q + (uint32_t)(kt * 16 + 4 * g));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = kt * 16 + 4 * g + r;
            if (j < L && vi) {
              const float x = -((sN[j] + nq) - 2.0f * acc[r]) * rsq_hd;
              float dw, pd;
              wm_pair_bwd<HD>(x, j > i || sKv[j] == 0.f, dead_i, lse, ks4[r], gm[r], gc[r], delta, rsq_hd, dw, pd);
              dwt[t][r] = dw;
              dwsum += dw;
            }
          }
        }
      }
      const Frag8 bdw = wm_acc_frag(dwt[0], dwt[1]);        // dQcat^T[k][i] += sum_j Kcat^T[k][j] dW^T[j][i] over this pair's 32 keys
#pragma unroll
      for (int ft = 0; ft < S::KC / 16; ++ft) accq[ft] = mma16<PREC>(accq[ft], frag_slotc(sCT + (16 * ft + c) * RST + 32 * kp, g), bdw);
    }
    dwsum = wm_colsum(dwsum);
    if (vi) {
#pragma unroll
      for (int ft = 0; ft < S::KC / 16; ++ft) {
        const int k0 = 16 * ft + 4 * g;          // features k0 .. k0+3 of the concatenation
        const f32x4& acc = accq[ft];
        if (k0 < HD) {
          const float4 q = *reinterpret_cast<const float4*>(a.Qm + qrow * a.ldqm + h * HD + k0);
          *reinterpret_cast<float4*>(a.dQm + qrow * a.ldd + h * HD + k0) =
              make_float4(2.0f * q.x * dwsum - 2.0f * acc[0], 2.0f * q.y * dwsum - 2.0f * acc[1], 2.0f * q.z * dwsum - 2.0f * acc[2], 2.0f * q.w * dwsum - 2.0f * acc[3]);
        } else {
          const float4 qc = *reinterpret_cast<const float4*>(a.Qc + qrow * a.ldqc + h * HD + k0 - HD);
          const float s0 = w_sqrt_cov(qc.x), s1 = w_sqrt_cov(qc.y), s2 = w_sqrt_cov(qc.z), s3 = w_sqrt_cov(qc.w);
          // clamp(min = 1e-24) gates the sqrt path (cov = 0 exactly -> no gradient)
          *reinterpret_cast<float4*>(a.dQc + qrow * a.ldd + h * HD + k0 - HD) =
              make_float4(dwsum - (s0 > 1.00001e-12f ? acc[0] / s0 : 0.f), dwsum - (s1 > 1.00001e-12f ? acc[1] / s1 : 0.f),
                          dwsum - (s2 > 1.00001e-12f ? acc[2] / s2 : 0.f), dwsum - (s3 > 1.00001e-12f ? acc[3] / s3 : 0.f));
        }
      }
    }
  }

  // ---- phase B staging: the query side over the same LDS ------------------------------------------------------------------------------
  __syncthreads();
  wm_stage_t<false>(sCT, a.Qm + row_b * a.ldqm + h * HD, a.ldqm, L, Lp, HD);
  wm_stage_t<true>(sCT + HD * RST, a.Qc + row_b * a.ldqc + h * HD, a.ldqc, L, Lp, HD);
  wm_stage_nat<HD>(sX0, a.dOm + row_b * a.lddom + h * HD, a.lddom, L, Lp);
  wm_stage_nat<HD>(sX1, a.dOc + row_b * a.lddoc + h * HD, a.lddoc, L, Lp);
  wm_stage_t<false>(sT0, a.dOm + row_b * a.lddom + h * HD, a.lddom, L, Lp, HD);
  wm_stage_t<false>(sT1, a.dOc + row_b * a.lddoc + h * HD, a.lddoc, L, Lp, HD);
  wm_stage_cat<HD>(sC, sN, a.Qm + row_b * a.ldqm + h * HD, a.ldqm, a.Qc + row_b * a.ldqc + h * HD, a.ldqc, L, Lp);
  __syncthreads();

  // ---- pass B: wave owns key tile kt ; tiles T[query 4g+r][key c] (operands swapped: the QUERY image rows are the A operand) -----------
  for (int kt = w; kt < nt16; kt += WM_NW) {
    const int j = kt * 16 + c;
    const bool vj = j < L;
    const size_t krow = row_b + (vj ? j : 0);
    Frag8 fk[S::KB], fvm[KBV], fvc[KBV];
#pragma unroll
    for (int kb = 0; kb < S::KB; ++kb) fk[kb] = wm_cat_frag<HD>(a.Km + krow * a.ldkm + h * HD, a.Kc + krow * a.ldkc + h * HD, kb, g, vj);
    const float nk = vj ? wm_row_norm<HD>(a.Km + krow * a.ldkm + h * HD, a.Kc + krow * a.ldkc + h * HD) : 0.f;
#pragma unroll
    for (int kb = 0; kb < KBV; ++kb) {
      fvm[kb] = frag_contig_hd<HD>(a.Vm + krow * a.ldvm + h * HD, kb, g);
      fvc[kb] = frag_contig_hd<HD>(a.Vc + krow * a.ldvc + h * HD, kb, g);
      if (!vj) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) { fvm[kb].v[jj] = 0.f; fvc[kb].v[jj] = 0.f; }
      }
    }
    const bool key_pad = !vj || sKv[j] == 0.f;
    f32x4 acck[S::KC / 16], accvm[S::NT], accvc[S::NT];
#pragma unroll
    for (int ft = 0; ft < S::KC / 16; ++ft) acck[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < S::NT; ++nt) { accvm[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; accvc[nt] = accvm[nt]; }
    float dwsum = 0.f;
#pragma unroll 1
    for (int qp = 0; qp < npair; ++qp) {
      f32x4 dwt[2], pdt[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int it = 2 * qp + t;
        dwt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        pdt[t] = dwt[t];
        if (it < nt16) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, gm = acc, gc = acc;
#pragma unroll
          for (int kb = 0; kb < S::KB; ++kb) acc = mma16<PREC>(acc, frag_contig(sC + (it * 16 + c) * S::RSK + 32 * kb + 8 * g), fk[kb]);
#pragma unroll
          for (int kb = 0; kb < KBV; ++kb) {
            gm = mma16<PREC>(gm, frag_contig_hd<HD>(sX0 + (it * 16 + c) * S::RSV, kb, g), fvm[kb]);
            gc = mma16<PREC>(gc, frag_contig_hd<HD>(sX1 + (it * 16 + c) * S::RSV, kb, g), fvc[kb]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + 4 * g + r;
            if (i < L && vj) {
              const float x = -((nk + sN[i]) - 2.0f * acc[r]) * rsq_hd;      // nk + nq: the same two numbers as in the forward, addition commutes
              const float ks_ = wm_keep_scale(a.drop, key_rng, (idx_bh + (uint32_t)i) * (uint32_t)L + (uint32_t)j);
              float dw, pd;
              wm_pair_bwd<HD>(x, j > i || key_pad, sDead[i] != 0.f, sLse[i], ks_, gm[r], gc[r], sDelta[i], rsq_hd, dw, pd);
              dwt[t][r] = dw;
              pdt[t][r] = pd;
              dwsum += dw;
            }
          }
        }
      }
      // dKcat^T[k][j] += sum_i Qcat^T[k][i] dW[i][j] ; dVm^T[d][j] += sum_i dOm^T[d][i] P~[i][j] ; dVc^T[d][j] += sum_i dOc^T[d][i] P~^2[i][j]
      const Frag8 bdw = wm_acc_frag(dwt[0], dwt[1]), bp = wm_acc_frag(pdt[0], pdt[1]);
      Frag8 bp2;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) bp2.v[jj] = bp.v[jj] * bp.v[jj];
#pragma unroll
      for (int ft = 0; ft < S::KC / 16; ++ft) acck[ft] = mma16<PREC>(acck[ft], frag_slotc(sCT + (16 * ft + c) * RST + 32 * qp, g), bdw);
#pragma unroll
      for (int nt = 0; nt < S::NT; ++nt) {
        accvm[nt] = mma16<PREC>(accvm[nt], frag_slotc(sT0 + (16 * nt + c) * RST + 32 * qp, g), bp);
        accvc[nt] = mma16<PREC>(accvc[nt], frag_slotc(sT1 + (16 * nt + c) * RST + 32 * qp, g), bp2);
      }
    }
    dwsum = wm_colsum(dwsum);
    if (vj) {
#pragma unroll
      for (int ft = 0; ft < S::KC / 16; ++ft) {
        const int k0 = 16 * ft + 4 * g;
        const f32x4& acc = acck[ft];
        if (k0 < HD) {
          const float4 k = *reinterpret_cast<const float4*>(a.Km + krow * a.ldkm + h * HD + k0);
          *reinterpret_cast<float4*>(a.dKm + krow * a.ldd + h * HD + k0) =
              make_float4(2.0f * k.x * dwsum - 2.0f * acc[0], 2.0f * k.y * dwsum - 2.0f * acc[1], 2.0f * k.z * dwsum - 2.0f * acc[2], 2.0f * k.w * dwsum - 2.0f * acc[3]);
        } else {
          const float4 kc = *reinterpret_cast<const float4*>(a.Kc + krow * a.ldkc + h * HD + k0 - HD);
          const float s0 = w_sqrt_cov(kc.x), s1 = w_sqrt_cov(kc.y), s2 = w_sqrt_cov(kc.z), s3 = w_sqrt_cov(kc.w);
          *reinterpret_cast<float4*>(a.dKc + krow * a.ldd + h * HD + k0 - HD) =
              make_float4(dwsum - (s0 > 1.00001e-12f ? acc[0] / s0 : 0.f), dwsum - (s1 > 1.00001e-12f ? acc[1] / s1 : 0.f),
                          dwsum - (s2 > 1.00001e-12f ? acc[2] / s2 : 0.f), dwsum - (s3 > 1.00001e-12f ? acc[3] / s3 : 0.f));
        }
      }
#pragma unroll
      for (int nt = 0; nt < S::NT; ++nt) {
        *reinterpret_cast<float4*>(a.dVm + krow * a.ldd + h * HD + 16 * nt + 4 * g) = make_float4(accvm[nt][0], accvm[nt][1], accvm[nt][2], accvm[nt][3]);
        *reinterpret_cast<float4*>(a.dVc + krow * a.ldd + h * HD + 16 * nt + 4 * g) = make_float4(accvc[nt][0], accvc[nt][1], accvc[nt][2], accvc[nt][3]);
      }
    }
  }
}

}  // namespace adt
