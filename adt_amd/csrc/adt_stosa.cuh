// STOSA-ADT kernels: Wasserstein-distance attention (stosa/modules.py:30-43 wasserstein_distance_matmul, :222-275
// DistAttention.forward, :311-361 DistEDAttention), the BPR / positive-vs-negative loss on elementwise Wasserstein
// distances (stosa/trainer.py:358-391) and the full-sort distance of the last state to every item (:464-479).
//
// All of it is exact fp32 on the vector ALUs: the attention is 0.65 GFLOP per launch at the Beauty template
// (B 256, H 4, L 100, hd 16), so the work is organised for wavefront-level reductions instead of MFMA tiles --
// one workgroup per (sequence, head) with the key side (or, backward pass B, the query side) of the head resident
// in LDS, one wave per query row, one key per lane (up to 4 for L <= 256); softmax statistics are wave reductions,
// and the probability row goes through LDS once so the lanes can re-map to (feature, partition) for P V.
//
//   s_ij = -(|mq_i|^2 + sum(Sq_i) + |mk_j|^2 + sum(Sk_j) - 2 (mq_i . mk_j + sqrt(Sq_i) . sqrt(Sk_j))) / sqrt(hd) + A_ij
//   A_ij = -2^32 where key j is padding or j > i (additive: a fully masked row is uniform over all keys AND keeps
//          its gradient, exactly as in the reference), P = softmax(s), Pd = dropout(P),
//   mean context = Pd Vm, covariance context = (Pd * Pd) Vc                      (modules.py:253-256)
#pragma once
#include "adt_common.cuh"

namespace adt {

struct WAttnArgs {
  const float* Qm; int ldqm; const float* Qc; int ldqc;   // query mean / covariance (T x >= H*hd), cov = ELU(.)+1 already
  const float* Km; int ldkm; const float* Kc; int ldkc;
  const float* Vm; int ldvm; const float* Vc; int ldvc;
  const int* kid;            // (B*L) ids of the KEY sequence: key j is padding when kid <= 0
  int B, H, L, hd;
  float scale;               // 1/sqrt(hd) (unused: the kernels divide, like the reference)
  DropCfg drop; uint32_t bh_offset;     // idx = ((bh + bh_offset) * L + i) * L + j
  float* Om; int ldom; float* Oc; int ldoc;   // contexts (T x H*hd)
  float* LSE;                // (B*H*L)
  const float* dOm; int lddom; const float* dOc; int lddoc;
  float *dQm, *dQc, *dKm, *dKc, *dVm, *dVc; int ldd;   // gradients (T x ldd), overwritten
};

constexpr float W_MASK = -4294967296.0f;   // float32(-2**32 + 1), stosa/models.py:226,230
constexpr int W_KPL = 4;                    // keys per lane: L <= 256

static inline size_t wattn_lds_bytes(int L, int hd, bool bwd) {
  // 4 resident tensors [L][hd+1], 2 row vectors [L] (+ 2 more in the backward), per-wave scratch 4 x (3*L' + 4*hd)
  const int Lp = (L + 63) / 64 * 64;
  size_t f = (size_t)4 * L * (hd + 1) + (size_t)(bwd ? 4 : 2) * Lp + (size_t)4 * (3 * Lp + 4 * hd);
  return f * sizeof(float);
}

ADT_DEVICE_INLINE float w_sqrt_cov(float c) { return sqrtf(fmaxf(c, 1e-24f)); }

// stage a head slice (L x hd) into LDS rows of stride hd+1, optionally through sqrt(clamp(.)); also return via `norm`
// (if non-null) the per-row sums |m|^2 (SQUARE) or sum(c) (plain) ACCUMULATED into norm[r]
template <bool SQRT, bool SQUARE_NORM>
ADT_DEVICE_INLINE void w_stage(float* dst, const float* g, int ld, int L, int hd, float* norm) {
  const int RS = hd + 1;
  // 16 lanes per row (hd / 16 columns each): the row's partial sums meet with 4 shuffles
  const int sub = threadIdx.x & 15;
  for (int r = threadIdx.x >> 4; r < L; r += 16) {
    float part = 0.f;
    for (int cidx = sub; cidx < hd; cidx += 16) {
      const float v = g[(size_t)r * ld + cidx];
      dst[r * RS + cidx] = SQRT ? w_sqrt_cov(v) : v;
      part += SQUARE_NORM ? v * v : v;
    }
    if (norm) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if (sub == 0) norm[r] += part;
    }
  }
}

// A query with no attendable key (the left-padded prefix) has EVERY score shifted by the mask: in fp32 x - 2^32 rounds to a
// multiple of 512, so all of them collapse to -2^32 and the row is uniform over all L keys.  -2^32 + log(L) is not
// representable, so such rows keep the reference's rounding but drop the common shift (softmax is shift-invariant and
// (x + MASK) - MASK is exact): probabilities and gradients are unchanged, the saved LSE stays accurate.
ADT_DEVICE_INLINE float w_score(float x, bool masked, bool dead_row) {
  if (dead_row) return (x + W_MASK) - W_MASK;
  return masked ? x + W_MASK : x;
}

// The scaled negative Wasserstein distance of one (query, key) pair, computed by ONE sequence of fused multiply-adds so that
// the forward pass and both backward passes reproduce it bit for bit: on fully masked rows x - 2^32 is quantised to multiples
// of 512 (see w_score), and a last-bit difference between the forward's and the backward's x would flip that rounding and
// turn exp(s - LSE) into exp(+-512).  (sum(cov) is taken as sum(sqrt(cov)^2): identical for cov >= 1e-24.)
ADT_DEVICE_INLINE float w_pair_x(const float* qm, const float* qs, const float* km, const float* ks, int hd, float sq_hd) {
  float dm = 0.f, dc = 0.f, nqm = 0.f, nqc = 0.f, nkm = 0.f, nkc = 0.f;
  for (int d0 = 0; d0 < hd; ++d0) {
    const float a = qm[d0], b = qs[d0], c = km[d0], e = ks[d0];
    dm = fmaf(a, c, dm);
    dc = fmaf(b, e, dc);
    nqm = fmaf(a, a, nqm);
    nqc = fmaf(b, b, nqc);
    nkm = fmaf(c, c, nkm);
    nkc = fmaf(e, e, nkc);
  }
  const float wd = (fmaf(-2.0f, dm, nqm) + nkm) + (fmaf(-2.0f, dc, nqc) + nkc);
  return (-wd) / sq_hd;
}

ADT_DEVICE_INLINE void w_mark_dead(float* sDead, const float* sKv, int L, int Lp) {
  for (int i = threadIdx.x; i < Lp; i += 256) {
    float cnt = 0.f;
    for (int j = 0; j <= i && j < L; ++j) cnt += sKv[j];
    sDead[i] = cnt == 0.f ? 1.f : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_wattn_fwd(WAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, hd = a.hd, RS = hd + 1, Lp = (L + 63) / 64 * 64;
  float* sKm = smem;                 // [L][RS]
  float* sKs = sKm + L * RS;         // sqrt(cov)
  float* sVm = sKs + L * RS;
  float* sVc = sVm + L * RS;
  float* sDead = sVc + L * RS;       // 1 where query i has no attendable key (j <= i, real item)   [Lp]
  float* sKv = sDead + Lp;           // key validity (1/0) [Lp]
  float* sWave = sKv + Lp;           // per wave: P row [Lp], P^2 row [Lp], (unused) [Lp], q mean [hd], q sqrt cov [hd], 2*hd spare
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t row_b = (size_t)b * L;
  for (int i = threadIdx.x; i < Lp; i += 256) sKv[i] = (i < L && a.kid[row_b + i] > 0) ? 1.f : 0.f;
  __syncthreads();
  w_mark_dead(sDead, sKv, L, Lp);
  // reference order (modules.py:31-41): mean part and covariance part are summed separately, then added
  w_stage<false, true>(sKm, a.Km + row_b * a.ldkm + h * hd, a.ldkm, L, hd, nullptr);
  w_stage<true, false>(sKs, a.Kc + row_b * a.ldkc + h * hd, a.ldkc, L, hd, nullptr);
  w_stage<false, false>(sVm, a.Vm + row_b * a.ldvm + h * hd, a.ldvm, L, hd, nullptr);
  w_stage<false, false>(sVc, a.Vc + row_b * a.ldvc + h * hd, a.ldvc, L, hd, nullptr);
  __syncthreads();
  float* sP = sWave + w * (3 * Lp + 4 * hd);
  float* sP2 = sP + Lp;
  float* sQm = sP + 3 * Lp;
  float* sQs = sQm + hd;
  const uint32_t key_rng = drop_key(a.drop);
  const float sq_hd = sqrtf((float)hd);
  const int nd = 64 / hd >= 1 ? hd : 64;      // lanes along the feature index in the P V step
  const int nparts = 64 / nd;                  // partitions of the key range (hd = 16: 4, 32: 2, 64: 1)
  for (int i = w; i < L; i += 4) {
    // query row -> per-wave LDS (broadcast reads below); norms of the query
    for (int d0 = lane; d0 < hd; d0 += 64) {
      sQm[d0] = a.Qm[(row_b + i) * a.ldqm + h * hd + d0];
      sQs[d0] = w_sqrt_cov(a.Qc[(row_b + i) * a.ldqc + h * hd + d0]);
    }
    const bool dead_i = sDead[i] != 0.f;
    float s[W_KPL];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < W_KPL; ++t) {
      const int j = lane + 64 * t;
      s[t] = -INFINITY;
      if (j < L) {
        s[t] = w_score(w_pair_x(sQm, sQs, sKm + j * RS, sKs + j * RS, hd, sq_hd), j > i || sKv[j] == 0.f, dead_i);
        m = fmaxf(m, s[t]);
      }
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < W_KPL; ++t) {
      const int j = lane + 64 * t;
      if (j < L) { s[t] = expf(s[t] - m); sum += s[t]; }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    if (lane == 0) a.LSE[(size_t)bh * L + i] = m + logf(sum);
    const uint32_t idx_q = ((uint32_t)(bh + a.bh_offset) * (uint32_t)L + (uint32_t)i) * (uint32_t)L;
#pragma unroll
    for (int t = 0; t < W_KPL; ++t) {
      const int j = lane + 64 * t;
      if (j < Lp) {
        float p = 0.f;
        if (j < L) {
          p = s[t] * inv;
          if (a.drop.thr) p = adt_keep(key_rng, idx_q + (uint32_t)j, a.drop.thr) ? p * a.drop.scale : 0.f;
        }
        sP[j] = p;
        sP2[j] = p * p;
      }
    }
    // contexts: lane -> (feature d0 = lane % nd [+ nd per step], partition of the keys)
    const int part = lane / nd;
    for (int d0 = lane % nd; d0 < hd; d0 += nd) {
      float cm = 0.f, cc = 0.f;
      for (int j = part; j < L; j += nparts) {
        cm += sP[j] * sVm[j * RS + d0];
        cc += sP2[j] * sVc[j * RS + d0];
      }
      for (int o = nd; o < 64; o <<= 1) { cm += __shfl_xor(cm, o, 64); cc += __shfl_xor(cc, o, 64); }
      if (part == 0) {
        a.Om[(row_b + i) * a.ldom + h * hd + d0] = cm;
        a.Oc[(row_b + i) * a.ldoc + h * hd + d0] = cc;
      }
    }
  }
}

// Backward.  Pass A (key side resident; wave owns a query row) -> dQm, dQc and delta_i; pass B (query side resident;
// wave owns a key row, lanes over the queries) -> dKm, dKc, dVm, dVc.  Every gradient element is produced by exactly
// one wave: no atomics, bitwise reproducible.
__global__ __launch_bounds__(256) void k_wattn_bwd(WAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, hd = a.hd, RS = hd + 1, Lp = (L + 63) / 64 * 64;
  float* sA0 = smem;                 // phase A: Km        | phase B: Qm
  float* sA1 = sA0 + L * RS;         // phase A: sqrt(Kc)  | phase B: sqrt(Qc)
  float* sA2 = sA1 + L * RS;         // phase A: Vm        | phase B: dOm
  float* sA3 = sA2 + L * RS;         // phase A: Vc        | phase B: dOc
  float* sKv = sA3 + L * RS;         // key validity [Lp]
  float* sDelta = sKv + Lp;          // [Lp] sum_j P_ij dP_ij
  float* sLse = sDelta + Lp;         // [Lp]
  float* sDead = sLse + Lp;          // [Lp] 1 where query i has no attendable key
  float* sWave = sDead + Lp;           // per wave: 3 rows [Lp] + 4 vectors [hd]
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t row_b = (size_t)b * L;
  const float sq_hd = sqrtf((float)hd);
  const uint32_t key_rng = drop_key(a.drop);
  const uint32_t idx_bh = (uint32_t)(bh + a.bh_offset) * (uint32_t)L;
  const int nd = 64 / hd >= 1 ? hd : 64;
  const int nparts = 64 / nd;
  float* sR0 = sWave + w * (3 * Lp + 4 * hd);   // rows over the other index
  float* sR1 = sR0 + Lp;
  float* sR2 = sR1 + Lp;
  float* sV0 = sR0 + 3 * Lp;                    // this wave's own row: mean, sqrt cov, dOm, dOc (or km, ks, vm, vc)
  float* sV1 = sV0 + hd;
  float* sV2 = sV1 + hd;
  float* sV3 = sV2 + hd;

  for (int i = threadIdx.x; i < Lp; i += 256) {
    sKv[i] = (i < L && a.kid[row_b + i] > 0) ? 1.f : 0.f;
    sLse[i] = i < L ? a.LSE[(size_t)bh * L + i] : INFINITY;
    sDelta[i] = 0.f;
  }
  w_stage<false, false>(sA0, a.Km + row_b * a.ldkm + h * hd, a.ldkm, L, hd, nullptr);
  w_stage<true, false>(sA1, a.Kc + row_b * a.ldkc + h * hd, a.ldkc, L, hd, nullptr);
  w_stage<false, false>(sA2, a.Vm + row_b * a.ldvm + h * hd, a.ldvm, L, hd, nullptr);
  w_stage<false, false>(sA3, a.Vc + row_b * a.ldvc + h * hd, a.ldvc, L, hd, nullptr);
  __syncthreads();
  w_mark_dead(sDead, sKv, L, Lp);
  __syncthreads();

  // ---- pass A ---------------------------------------------------------------------------------------------------
  for (int i = w; i < L; i += 4) {
    for (int d0 = lane; d0 < hd; d0 += 64) {
      sV0[d0] = a.Qm[(row_b + i) * a.ldqm + h * hd + d0];
      sV1[d0] = w_sqrt_cov(a.Qc[(row_b + i) * a.ldqc + h * hd + d0]);
      sV2[d0] = a.dOm[(row_b + i) * a.lddom + h * hd + d0];
      sV3[d0] = a.dOc[(row_b + i) * a.lddoc + h * hd + d0];
    }
    const float lse = sLse[i];
    const bool dead_i = sDead[i] != 0.f;
    const uint32_t idx_q = (idx_bh + (uint32_t)i) * (uint32_t)L;
    float p[W_KPL], dp[W_KPL];
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < W_KPL; ++t) {
      const int j = lane + 64 * t;
      p[t] = 0.f; dp[t] = 0.f;
      if (j < L) {
        float gm = 0.f, gc = 0.f;
        for (int d0 = 0; d0 < hd; ++d0) {
          gm += sV2[d0] * sA2[j * RS + d0];
          gc += sV3[d0] * sA3[j * RS + d0];
        }
        const float sv = w_score(w_pair_x(sV0, sV1, sA0 + j * RS, sA1 + j * RS, hd, sq_hd), j > i || sKv[j] == 0.f, dead_i);
        const float pr = expf(sv - lse);
        float ks_ = 1.0f;
        if (a.drop.thr) ks_ = adt_keep(key_rng, idx_q + (uint32_t)j, a.drop.thr) ? a.drop.scale : 0.f;
        const float pd = pr * ks_;
        const float dpd = gm + 2.0f * pd * gc;      // d loss / d Pd_ij
        p[t] = pr;
        dp[t] = dpd * ks_;                           // d loss / d P_ij
        delta += pr * dp[t];
      }
    }
    delta = wave_sum(delta);
    if (lane == 0) sDelta[i] = delta;
    // dW_ij = -dS_ij / sqrt(hd), dS_ij = P_ij (dP_ij - delta_i)
    float dwsum = 0.f;
#pragma unroll
    for (int t = 0; t < W_KPL; ++t) {
      const int j = lane + 64 * t;
      if (j < Lp) {
        const float dwv = j < L ? -(p[t] * (dp[t] - delta)) / sq_hd : 0.f;
        sR0[j] = dwv;
        dwsum += dwv;
      }
    }
    dwsum = wave_sum(dwsum);
    const int part = lane / nd;
    for (int d0 = lane % nd; d0 < hd; d0 += nd) {
      float am = 0.f, as = 0.f;
      for (int j = part; j < L; j += nparts) {
        am += sR0[j] * sA0[j * RS + d0];
        as += sR0[j] * sA1[j * RS + d0];
      }
      for (int o = nd; o < 64; o <<= 1) { am += __shfl_xor(am, o, 64); as += __shfl_xor(as, o, 64); }
      if (part == 0) {
        // W = |mq|^2 + sum(Sq) + ... - 2 (mq . mk + sqrt(Sq) . sqrt(Sk))
        a.dQm[(row_b + i) * a.ldd + h * hd + d0] = 2.0f * sV0[d0] * dwsum - 2.0f * am;
        a.dQc[(row_b + i) * a.ldd + h * hd + d0] = dwsum - (sV1[d0] > 1.00001e-12f ? as / sV1[d0] : 0.f);   // clamp(min=1e-24) gates the sqrt path (cov = 0 exactly -> no gradient)
      }
    }
  }

  // ---- phase B staging: the query side over the same LDS -----------------------------------------------------------
  __syncthreads();
  w_stage<false, true>(sA0, a.Qm + row_b * a.ldqm + h * hd, a.ldqm, L, hd, nullptr);
  w_stage<true, false>(sA1, a.Qc + row_b * a.ldqc + h * hd, a.ldqc, L, hd, nullptr);
  w_stage<false, false>(sA2, a.dOm + row_b * a.lddom + h * hd, a.lddom, L, hd, nullptr);
  w_stage<false, false>(sA3, a.dOc + row_b * a.lddoc + h * hd, a.lddoc, L, hd, nullptr);
  __syncthreads();

  // ---- pass B: wave owns key j, lanes over queries i ------------------------------------------------------------------
  for (int j = w; j < L; j += 4) {
    for (int d0 = lane; d0 < hd; d0 += 64) {
      sV0[d0] = a.Km[(row_b + j) * a.ldkm + h * hd + d0];
      sV1[d0] = w_sqrt_cov(a.Kc[(row_b + j) * a.ldkc + h * hd + d0]);
      sV2[d0] = a.Vm[(row_b + j) * a.ldvm + h * hd + d0];
      sV3[d0] = a.Vc[(row_b + j) * a.ldvc + h * hd + d0];
    }
    const bool key_pad = sKv[j] == 0.f;
    float dwsum = 0.f;
#pragma unroll
    for (int t = 0; t < W_KPL; ++t) {
      const int i = lane + 64 * t;
      if (i < Lp) {
        float dwv = 0.f, pd = 0.f;
        if (i < L) {
          float gm = 0.f, gc = 0.f;
          for (int d0 = 0; d0 < hd; ++d0) {
            gm += sA2[i * RS + d0] * sV2[d0];
            gc += sA3[i * RS + d0] * sV3[d0];
          }
          const float sv = w_score(w_pair_x(sA0 + i * RS, sA1 + i * RS, sV0, sV1, hd, sq_hd), j > i || key_pad, sDead[i] != 0.f);
          const float pr = expf(sv - sLse[i]);
          float ks_ = 1.0f;
          if (a.drop.thr) ks_ = adt_keep(key_rng, (idx_bh + (uint32_t)i) * (uint32_t)L + (uint32_t)j, a.drop.thr) ? a.drop.scale : 0.f;
          pd = pr * ks_;
          const float dpr = (gm + 2.0f * pd * gc) * ks_;
          dwv = -(pr * (dpr - sDelta[i])) / sq_hd;
        }
        sR0[i] = dwv;
        sR1[i] = pd;
        sR2[i] = pd * pd;
        dwsum += dwv;
      }
    }
    dwsum = wave_sum(dwsum);
    const int part = lane / nd;
    for (int d0 = lane % nd; d0 < hd; d0 += nd) {
      float am = 0.f, as = 0.f, vm = 0.f, vc = 0.f;
      for (int i = part; i < L; i += nparts) {
        am += sR0[i] * sA0[i * RS + d0];
        as += sR0[i] * sA1[i * RS + d0];
        vm += sR1[i] * sA2[i * RS + d0];
        vc += sR2[i] * sA3[i * RS + d0];
      }
      for (int o = nd; o < 64; o <<= 1) {
        am += __shfl_xor(am, o, 64); as += __shfl_xor(as, o, 64);
        vm += __shfl_xor(vm, o, 64); vc += __shfl_xor(vc, o, 64);
      }
      if (part == 0) {
        a.dKm[(row_b + j) * a.ldd + h * hd + d0] = 2.0f * sV0[d0] * dwsum - 2.0f * am;
        a.dKc[(row_b + j) * a.ldd + h * hd + d0] = dwsum - (sV1[d0] > 1.00001e-12f ? as / sV1[d0] : 0.f);
        a.dVm[(row_b + j) * a.ldd + h * hd + d0] = vm;
        a.dVc[(row_b + j) * a.ldd + h * hd + d0] = vc;
      }
    }
  }
}

// ---- BPR + positive-vs-negative loss on elementwise Wasserstein distances, forward and backward in one pass ------
// stosa/trainer.py:358-391.  One wave per token; the item covariance rows go through ELU(.)+1 (:361-364).
struct WBprArgs {
  const float* Sm; const float* Sc; int lds;     // sequence mean / covariance outputs (T x d)
  const float* Em; const float* Ec;              // item tables ((V+..) x d)
  const int* pos; const int* neg; int T, d;
  float pvn_weight; const float* inv_count;      // 1 / sum(istarget) of the GLOBAL batch
  float* dSm; float* dSc; int ldds;              // overwritten
  float* dEm; float* dEc;                        // accumulated with atomics (rows of id 0 skipped: padding_idx)
  float* loss3;                                  // 3 x 64 slots: bpr, pvn (weighted), auc
};

ADT_DEVICE_INLINE float w_elu1(float x) { return (x > 0.f ? x : expf(x) - 1.0f) + 1.0f; }
ADT_DEVICE_INLINE float w_elu_grad(float x) { return x > 0.f ? 1.f : expf(x); }

__global__ __launch_bounds__(256) void k_wdist_bpr(WBprArgs a) {
  const int lane = threadIdx.x & 63;
  const float wgt = *a.inv_count;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float l_bpr = 0.f, l_pvn = 0.f, l_auc = 0.f;
  for (int t = wave; t < a.T; t += nwaves) {
    const int ip = a.pos[t], in = a.neg[t];
    const bool tgt = ip > 0;
    float dp = 0.f, dn = 0.f, dpn = 0.f;
    if (tgt) {
      for (int c = lane; c < a.d; c += 64) {
        const float sm = a.Sm[(size_t)t * a.lds + c], ss = w_sqrt_cov(a.Sc[(size_t)t * a.lds + c]);
        const float pm = a.Em[(size_t)ip * a.d + c], ps = w_sqrt_cov(w_elu1(a.Ec[(size_t)ip * a.d + c]));
        const float nm = a.Em[(size_t)in * a.d + c], ns = w_sqrt_cov(w_elu1(a.Ec[(size_t)in * a.d + c]));
        dp += (sm - pm) * (sm - pm) + (ss - ps) * (ss - ps);
        dn += (sm - nm) * (sm - nm) + (ss - ns) * (ss - ns);
        dpn += (pm - nm) * (pm - nm) + (ps - ns) * (ps - ns);
      }
    }
    // reference sums the mean part over d, then adds the covariance part's sum; a fused per-lane sum differs only in
    // rounding order (<= 1e-6 relative)
    dp = wave_sum(dp); dn = wave_sum(dn); dpn = wave_sum(dpn);
    float g_pos = 0.f, g_neg = 0.f, g_pvn = 0.f;
    if (tgt) {
      const float x = dn - dp + 1e-24f;
      const float sg = 1.0f / (1.0f + expf(-x));
      l_bpr += -logf(sg) * wgt;
      const float hinge = dp - dpn;
      l_pvn += a.pvn_weight * fmaxf(hinge, 0.f) * wgt;
      l_auc += ((dn - dp) > 0.f ? 1.f : ((dn - dp) == 0.f ? 0.5f : 0.f)) * wgt;   // (sign(neg - pos) + 1) / 2
      const float gx = -(1.0f - sg) * wgt;                 // d(-log sigmoid(x))/dx
      const float gh = hinge >= 0.f ? a.pvn_weight * wgt : 0.f;   // torch.clamp(., 0) passes the gradient at 0
      g_neg = gx;
      g_pos = -gx + gh;
      g_pvn = -gh;
    }
    for (int c = lane; c < a.d; c += 64) {
      float dsm = 0.f, dsc = 0.f;
      if (tgt) {
        const float sm = a.Sm[(size_t)t * a.lds + c], ss = w_sqrt_cov(a.Sc[(size_t)t * a.lds + c]);
        const float pe = a.Ec[(size_t)ip * a.d + c], ne = a.Ec[(size_t)in * a.d + c];
        const float pm = a.Em[(size_t)ip * a.d + c], ps = w_sqrt_cov(w_elu1(pe));
        const float nm = a.Em[(size_t)in * a.d + c], ns = w_sqrt_cov(w_elu1(ne));
        dsm = g_pos * 2.0f * (sm - pm) + g_neg * 2.0f * (sm - nm);
        // torch.clamp(cov, min=1e-24) passes no gradient below the clamp: the 1/sqrt factors are gated
        const float iss = ss > 1.00001e-12f ? 1.0f / ss : 0.f, ips = ps > 1.00001e-12f ? 1.0f / ps : 0.f, ins = ns > 1.00001e-12f ? 1.0f / ns : 0.f;
        dsc = g_pos * (ss - ps) * iss + g_neg * (ss - ns) * iss;
        const float dpm = -g_pos * 2.0f * (sm - pm) + g_pvn * 2.0f * (pm - nm);
        const float dpc = -g_pos * (ss - ps) * ips + g_pvn * (ps - ns) * ips;
        const float dnm = -g_neg * 2.0f * (sm - nm) - g_pvn * 2.0f * (pm - nm);
        const float dnc = -g_neg * (ss - ns) * ins - g_pvn * (ps - ns) * ins;
        atomicAdd(a.dEm + (size_t)ip * a.d + c, dpm);
        atomicAdd(a.dEc + (size_t)ip * a.d + c, dpc * w_elu_grad(pe));
        if (in > 0) {
          atomicAdd(a.dEm + (size_t)in * a.d + c, dnm);
          atomicAdd(a.dEc + (size_t)in * a.d + c, dnc * w_elu_grad(ne));
        }
      }
      a.dSm[(size_t)t * a.ldds + c] = dsm;
      a.dSc[(size_t)t * a.ldds + c] = dsc;
    }
  }
  if (lane == 0) {
    const int slot = wave & 63;
    if (l_bpr != 0.f) atomicAdd(a.loss3 + slot, l_bpr);
    if (l_pvn != 0.f) atomicAdd(a.loss3 + 64 + slot, l_pvn);
    if (l_auc != 0.f) atomicAdd(a.loss3 + 128 + slot, l_auc);
  }
}

// ---- full-sort scores: dist[b][v] = wasserstein_distance_matmul(last state b, item v) (stosa/trainer.py:464-479) ----
struct WFullArgs {
  const float* Sm; const float* Sc; int lds;   // B rows
  const float* Em; const float* Ec;            // V rows; covariance through ELU(.)+1
  int B, V, d;
  float* dist; int ldo;
};

__global__ __launch_bounds__(256) void k_wdist_full(WFullArgs a) {
  const int sub = threadIdx.x & 15;
  const size_t n = (size_t)a.B * a.V;
  for (size_t i = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); i < n; i += (size_t)gridDim.x * 16) {
    const int b = (int)(i / a.V), v = (int)(i % a.V);
    float dm = 0.f, dc = 0.f, n1m = 0.f, n2m = 0.f, n1c = 0.f, n2c = 0.f;
    for (int c4 = 4 * sub; c4 < a.d; c4 += 64) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sm = a.Sm[(size_t)b * a.lds + c4 + e], sc = a.Sc[(size_t)b * a.lds + c4 + e];
        const float em = a.Em[(size_t)v * a.d + c4 + e], ec = w_elu1(a.Ec[(size_t)v * a.d + c4 + e]);
        dm += sm * em;
        dc += w_sqrt_cov(sc) * w_sqrt_cov(ec);
        n1m += sm * sm; n2m += em * em; n1c += sc; n2c += ec;
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      dm += __shfl_xor(dm, o, 64); dc += __shfl_xor(dc, o, 64);
      n1m += __shfl_xor(n1m, o, 64); n2m += __shfl_xor(n2m, o, 64);
      n1c += __shfl_xor(n1c, o, 64); n2c += __shfl_xor(n2c, o, 64);
    }
    if (sub == 0) a.dist[(size_t)b * a.ldo + v] = ((-2.0f * dm + n1m) + n2m) + ((-2.0f * dc + n1c) + n2c);
  }
}

}  // namespace adt
