// bf16-operand attention (PREC_BF16 only): same algorithm, tiling and C-layout tricks as adt_attn.cuh, but the
// per-(b,h) LDS images are stored ONCE as bf16 -- row-major [L][hd] for operands contracted over hd and
// transposed [hd][L] for operands contracted over keys/queries -- so every MFMA operand that comes from LDS is
// a single ds_read_b128 / two ds_read_b64 with no conversion, instead of 2 ds_read_b128 / 8 ds_read_b32 of fp32
// plus 4 v_cvt_pk per fragment.  Scores, softmax statistics, P, dP, dS and all accumulators stay fp32.
#pragma once
#include "adt_attn.cuh"

namespace adt {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int HD> struct BImg {
  static constexpr int RS = HD + 8;             // row image stride (bf16 elements): conflict-free b128 reads
};

// fp32 global (L x HD slice, leading dim ld) -> bf16 row image [LP][RS] (+ optional transposed image [HD][LPT]);
// rows >= L are zero.  mul scales (1/sqrt(hd) for Q in the backward).
template <int HD, int NTH>
ADT_DEVICE_INLINE void stage_bf16(__bf16* rowimg, __bf16* timg, int LPT, const float* g, int ld, int L, int LP, float mul) {
  constexpr int RS = BImg<HD>::RS;
  constexpr int V8 = HD / 8;
  for (int i = threadIdx.x; i < LP * V8; i += NTH) {
    const int r = i / V8, c8 = (i % V8) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < L) {
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c8);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c8 + 4);
    }
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)(v[j] * mul);
    if (rowimg) *reinterpret_cast<bf16x8*>(rowimg + r * RS + c8) = b;
    if (timg) {
#pragma unroll
      for (int j = 0; j < 8; ++j) timg[(c8 + j) * LPT + r] = b[j];
    }
  }
}

// One 8-element chunk (row r, columns c8..c8+7) of a staged tensor, as loaded from global memory.
struct Chunk8 { float4 lo, hi; };

template <int HD>
ADT_DEVICE_INLINE Chunk8 chunk_load(const float* g, int ld, int i, int L, int LP) {
  constexpr int V8 = HD / 8;
  const int r = i / V8, c8 = (i % V8) * 8;
  Chunk8 x;
  x.lo = make_float4(0.f, 0.f, 0.f, 0.f);
  x.hi = x.lo;
  if (i < LP * V8 && r < L) {
    x.lo = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c8);
    x.hi = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c8 + 4);
  }
  return x;
}

template <int HD>
ADT_DEVICE_INLINE void chunk_store(__bf16* rowimg, __bf16* timg, int LPT, const Chunk8& x, int i, int LP, float mul) {
  constexpr int RS = BImg<HD>::RS;
  constexpr int V8 = HD / 8;
  if (i >= LP * V8) return;
  const int r = i / V8, c8 = (i % V8) * 8;
  const float v[8] = {x.lo.x, x.lo.y, x.lo.z, x.lo.w, x.hi.x, x.hi.y, x.hi.z, x.hi.w};
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)(v[j] * mul);
  if (rowimg) *reinterpret_cast<bf16x8*>(rowimg + r * RS + c8) = b;
  if (timg) {
#pragma unroll
    for (int j = 0; j < 8; ++j) timg[(c8 + j) * LPT + r] = b[j];
  }
}

// 8 contiguous hd-elements of one row (operand with the lane on the row, contraction over hd)
template <int HD>
ADT_DEVICE_INLINE bf16x8 rfrag(const __bf16* rowimg, int row, int kb, int g) {
  constexpr int RS = BImg<HD>::RS;
  if (kb * 32 + 8 * g < HD) return *reinterpret_cast<const bf16x8*>(rowimg + row * RS + kb * 32 + 8 * g);
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
  return z;
}

// operand with the lane on an hd column, contraction over 32 rows in slot order (4g+j | 16+4g+j)
ADT_DEVICE_INLINE bf16x8 tfrag(const __bf16* timg, int LPT, int col, int rowbase, int g) {
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(timg + col * LPT + rowbase + 4 * g);
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(timg + col * LPT + rowbase + 16 + 4 * g);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
  return o;
}

ADT_DEVICE_INLINE bf16x8 pack8(const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
  return o;
}

ADT_DEVICE_INLINE f32x4 mfma_bf16(f32x4 acc, bf16x8 a, bf16x8 b) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); }

template <int HD, int MAXKT>
struct AttnBf16Lds {
  static constexpr int LP = MAXKT * 16, LPT = LP + 8, RS = BImg<HD>::RS;
  static constexpr size_t fwd_bytes = (size_t)(LP * RS + HD * LPT) * 2;
  static constexpr size_t bwd_bytes = (size_t)(4 * LP * RS + 3 * HD * LPT) * 2 + 2 * LP * sizeof(float) + (size_t)LP * 8 * sizeof(uint32_t);
};

template <int HD, int MAXKT, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_fwd_bf16(AttnArgs a) {
  constexpr int LP = MAXKT * 16, LPT = LP + 8, RS = BImg<HD>::RS, NT = HD / 16, KB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);       // [LP][RS]
  __bf16* sVT = sK + LP * RS;                              // [HD][LPT]
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L;
  const size_t row_b = (size_t)b * L;
  {
    // all global loads of the staging are issued before the first conversion: one memory latency, not one per image
    constexpr int NIT = (LP * (HD / 8) + NW * 64 - 1) / (NW * 64);
    Chunk8 ck[NIT], cv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      ck[it] = chunk_load<HD>(a.K + row_b * a.ldk + h * HD, a.ldk, threadIdx.x + it * NW * 64, L, LP);
      cv[it] = chunk_load<HD>(a.V + row_b * a.ldv + h * HD, a.ldv, threadIdx.x + it * NW * 64, L, LP);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      chunk_store<HD>(sK, nullptr, LPT, ck[it], threadIdx.x + it * NW * 64, LP, 1.0f);
      chunk_store<HD>(nullptr, sVT, LPT, cv[it], threadIdx.x + it * NW * 64, LP, 1.0f);
    }
  }
  __syncthreads();
  const uint32_t key_rng = drop_key(a.drop);
  const int nqt = (L + 15) / 16;
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    const int tix = rnd * NW + ((rnd & 1) ? NW - 1 - w : w);   // snake order, heaviest causal tile first
    if (tix >= nqt) continue;
    const int qt = nqt - 1 - tix;
    const int q = qt * 16 + c;
    bf16x8 fq[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (q < L && kb * 32 + 8 * g < HD) {
        const float* p = a.Q + (row_b + q) * a.ldq + h * HD + kb * 32 + 8 * g;
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(p);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= a.scale;
      }
      fq[kb] = pack8(v);
    }
    const int nkt = a.causal ? qt + 1 : nqt;
    // Two sweeps over the key tiles, the scores recomputed in the second (one MFMA per tile at hd = 32) instead of kept: the
    // 56 registers of a 14-tile score array were what held this kernel at 168 VGPRs = one workgroup per CU; without them two
    // or three workgroups share a CU and one stages K/V while another multiplies.
    auto score = [&](int kt, f32x4& sc) {
      sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) sc = mfma_bf16(sc, rfrag<HD>(sK, kt * 16 + c, kb, g), fq[kb]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const bool valid = key < L && (!a.causal || key <= q);
        sc[r] = valid ? sc[r] : -INFINITY;
      }
    };
    float m = -INFINITY;
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
      f32x4 sc;
      score(kt, sc);
      m = fmaxf(fmaxf(m, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const uint32_t idx_q = ((uint32_t)(bh + a.bh_offset) * (uint32_t)L + (uint32_t)q) * (uint32_t)L;
    float sum = 0.f;
    uint32_t mw[MAXKT / 2];      // keep bits of this lane's keys: word kp, bit 16*(kt&1) + 4g + r
#pragma unroll
    for (int i = 0; i < MAXKT / 2; ++i) mw[i] = 0u;
    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < MAXKT / 2; ++kp) {
      if (2 * kp < nkt) {
        float pv[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * kp + t;
          f32x4 sc = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          if (kt < nkt) score(kt, sc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __expf(sc[r] - m);        // exp(-inf) = 0 for masked / absent keys
            sum += e;
            float p = e;
            if (a.drop.thr) {
              const uint32_t key = kt * 16 + 4 * g + r;
              const bool keep = kt < nkt && adt_keep(key_rng, idx_q + key, a.drop.thr);
              p = keep ? e * a.drop.scale : 0.f;
              mw[kp] |= (keep ? 1u : 0u) << (16 * t + 4 * g + r);
            }
            pv[4 * t + r] = p;
          }
        }
        const bf16x8 fp = pack8(pv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = mfma_bf16(o[nt], fp, tfrag(sVT, LPT, nt * 16 + c, kp * 32, g));
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (g == 0 && q < L) a.LSE[(size_t)bh * L + q] = m + __logf(sum);
    // the output rows of this lane are queries 4g + r (accumulator layout), the softmax sums live on lanes c = query: fetch them
    float inv_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) inv_r[r] = 1.0f / __shfl(sum, (lane & 48) | (4 * g + r), 64);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[nt][r] *= inv_r[r];
    if (a.mask && a.drop.thr) {
      // the four g-lanes of a query hold disjoint nibbles: OR them together, lane g == 0 stores the 8 words
      uint32_t ow[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t v = i < MAXKT / 2 ? mw[i] : 0u;
        v |= (uint32_t)__shfl_xor((int)v, 16, 64);
        v |= (uint32_t)__shfl_xor((int)v, 32, 64);
        ow[i] = v;
      }
      if (g == 0 && q < L) {
        uint4* dst = reinterpret_cast<uint4*>(a.mask + ((size_t)bh * L + q) * 8);
        dst[0] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        dst[1] = make_uint4(ow[4], ow[5], ow[6], ow[7]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qt * 16 + 4 * g + r;
        if (qq < L) a.O[(row_b + qq) * a.ldo + h * HD + nt * 16 + c] = o[nt][r];
      }
  }
}

// Backward, two passes over bf16 images (see adt_attn.cuh for the algorithm): pass A (wave owns a q tile) -> dQ,
// pass B (wave owns a key tile) -> dK, dV.  Images: Qs (scaled), K, V, dO row-major; QsT, KT, dOT transposed.
template <int HD, int MAXKT, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_bwd_bf16(AttnArgs a) {
  constexpr int LP = MAXKT * 16, LPT = LP + 8, RS = BImg<HD>::RS, NT = HD / 16, KB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sQ = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sK = sQ + LP * RS;
  __bf16* sV = sK + LP * RS;
  __bf16* sdO = sV + LP * RS;
  __bf16* sQT = sdO + LP * RS;
  __bf16* sKT = sQT + HD * LPT;
  __bf16* sdOT = sKT + HD * LPT;
  float* sLse = reinterpret_cast<float*>(sdOT + HD * LPT);
  float* sDelta = sLse + LP;
  uint32_t* sM = reinterpret_cast<uint32_t*>(sDelta + LP);   // [LP][8] dropout keep bits (forward's), 0 beyond L
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L;
  const size_t row_b = (size_t)b * L;
  const bool use_bits = a.mask != nullptr && a.drop.thr != 0;
  if (use_bits) {
    for (int i = threadIdx.x; i < LP * 2; i += NW * 64) {
      const int r = i >> 1;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (r < L) v = reinterpret_cast<const uint4*>(a.mask + ((size_t)bh * L + r) * 8)[i & 1];
      reinterpret_cast<uint4*>(sM + r * 8)[i & 1] = v;
    }
  }
  {
    // all global loads of the staging (Q, K, V, dO, O) are issued before the first conversion; delta = rowsum(dO*O)
    // comes from the same dO / O chunks (the HD/8 lanes of a row are adjacent)
    constexpr int V8 = HD / 8;
    constexpr int NIT = (LP * V8 + NW * 64 - 1) / (NW * 64);
    Chunk8 cq[NIT], ck[NIT], cv[NIT], cd[NIT], co[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + it * NW * 64;
      cq[it] = chunk_load<HD>(a.Q + row_b * a.ldq + h * HD, a.ldq, i, L, LP);
      ck[it] = chunk_load<HD>(a.K + row_b * a.ldk + h * HD, a.ldk, i, L, LP);
      cv[it] = chunk_load<HD>(a.V + row_b * a.ldv + h * HD, a.ldv, i, L, LP);
      cd[it] = chunk_load<HD>(a.dO + row_b * a.lddo + h * HD, a.lddo, i, L, LP);
      co[it] = chunk_load<HD>(a.O + row_b * a.ldo + h * HD, a.ldo, i, L, LP);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + it * NW * 64;
      chunk_store<HD>(sQ, sQT, LPT, cq[it], i, LP, a.scale);
      chunk_store<HD>(sK, sKT, LPT, ck[it], i, LP, 1.0f);
      chunk_store<HD>(sV, nullptr, LPT, cv[it], i, LP, 1.0f);
      chunk_store<HD>(sdO, sdOT, LPT, cd[it], i, LP, 1.0f);
      float part = cd[it].lo.x * co[it].lo.x + cd[it].lo.y * co[it].lo.y + cd[it].lo.z * co[it].lo.z + cd[it].lo.w * co[it].lo.w +
                   cd[it].hi.x * co[it].hi.x + cd[it].hi.y * co[it].hi.y + cd[it].hi.z * co[it].hi.z + cd[it].hi.w * co[it].hi.w;
#pragma unroll
      for (int off = V8 / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      const int r = i / V8;
      if (i < LP * V8 && (i % V8) == 0) {
        sDelta[r] = part;
        sLse[r] = (r < L) ? a.LSE[(size_t)bh * L + r] : INFINITY;
      }
    }
  }
  __syncthreads();
  const uint32_t key_rng = drop_key(a.drop);
  const uint32_t idx_bh = (uint32_t)(bh + a.bh_offset) * (uint32_t)L;
  const int nqt = (L + 15) / 16;

  // ---- pass A: dQ ---------------------------------------------------------------------------------
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    const int tix = rnd * NW + ((rnd & 1) ? NW - 1 - w : w);
    if (tix >= nqt) continue;
    const int qt = nqt - 1 - tix;
    const int q = qt * 16 + c;
    const float lse_q = sLse[q], delta_q = sDelta[q];
    bf16x8 fq[KB], fdo[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      fq[kb] = rfrag<HD>(sQ, q, kb, g);
      fdo[kb] = rfrag<HD>(sdO, q, kb, g);
    }
    const int nkt = a.causal ? qt + 1 : nqt;
    f32x4 dq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dq[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t idx_q = (idx_bh + (uint32_t)q) * (uint32_t)L;
#pragma unroll 1
    for (int kp = 0; kp < MAXKT / 2; ++kp) {
      if (2 * kp < nkt) {
        float dsv[8];
        const uint32_t mword = use_bits ? (sM[q * 8 + kp] >> (4 * g)) : 0u;   // bits 16t + r of this lane's keys
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * kp + t;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          if (kt < nkt) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
              s = mfma_bf16(s, rfrag<HD>(sK, kt * 16 + c, kb, g), fq[kb]);
              dp = mfma_bf16(dp, rfrag<HD>(sV, kt * 16 + c, kb, g), fdo[kb]);
            }
          }
          // interior tiles (every key <= every query of the tile, all keys real) need no per-element mask test
          const bool edge = kt >= nkt || (a.causal && kt == qt) || kt == nqt - 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            float p = __expf(s[r] - lse_q);
            if (edge) {
              const bool valid = kt < nkt && key < L && (!a.causal || key <= q);
              p = valid ? p : 0.f;
            }
            float d = dp[r];
            if (use_bits) d = ((mword >> (16 * t + r)) & 1u) ? d * a.drop.scale : 0.f;
            else if (a.drop.thr) d = adt_keep(key_rng, idx_q + (uint32_t)key, a.drop.thr) ? d * a.drop.scale : 0.f;
            dsv[4 * t + r] = p * (d - delta_q);
          }
        }
        const bf16x8 fds = pack8(dsv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dq[nt] = mfma_bf16(dq[nt], fds, tfrag(sKT, LPT, nt * 16 + c, kp * 32, g));
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qt * 16 + 4 * g + r;
        if (qq < L) a.dQ[(row_b + qq) * a.lddq + h * HD + nt * 16 + c] = dq[nt][r] * a.scale;
      }
  }

  // ---- pass B: dK, dV ---------------------------------------------------------------------------
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    const int kt = rnd * NW + ((rnd & 1) ? NW - 1 - w : w);   // key tile 0 is the heaviest under the causal mask
    if (kt >= nqt) continue;
    const int key = kt * 16 + c;
    bf16x8 fk[KB], fv[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      fk[kb] = rfrag<HD>(sK, key, kb, g);
      fv[kb] = rfrag<HD>(sV, key, kb, g);
    }
    f32x4 dk[NT], dv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      dk[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int qp0 = a.causal ? kt / 2 : 0;
#pragma unroll 1
    for (int qp = 0; qp < MAXKT / 2; ++qp) {
      if (qp >= qp0 && 2 * qp < nqt) {
        float pv[8], dsv[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int qt = 2 * qp + t;
          const bool live = qt < nqt && (!a.causal || qt >= kt);
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          if (live) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
              s = mfma_bf16(s, rfrag<HD>(sQ, qt * 16 + c, kb, g), fk[kb]);
              dp = mfma_bf16(dp, rfrag<HD>(sdO, qt * 16 + c, kb, g), fv[kb]);
            }
          }
          const bool edge = !live || (a.causal && qt == kt) || kt == nqt - 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qq = qt * 16 + 4 * g + r;
            float p = __expf(s[r] - sLse[qq]);      // sLse = +inf for padded queries -> 0
            if (edge) {
              const bool valid = live && key < L && (!a.causal || key <= qq);
              p = valid ? p : 0.f;
            }
            float ks = 1.0f;
            if (use_bits) ks = ((sM[qq * 8 + (kt >> 1)] >> (16 * (kt & 1) + c)) & 1u) ? a.drop.scale : 0.f;
            else if (a.drop.thr)
              ks = adt_keep(key_rng, (idx_bh + (uint32_t)qq) * (uint32_t)L + (uint32_t)key, a.drop.thr) ? a.drop.scale : 0.f;
            pv[4 * t + r] = p * ks;
            dsv[4 * t + r] = p * (dp[r] * ks - sDelta[qq]);
          }
        }
        const bf16x8 fp = pack8(pv), fds = pack8(dsv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          dv[nt] = mfma_bf16(dv[nt], fp, tfrag(sdOT, LPT, nt * 16 + c, qp * 32, g));
          dk[nt] = mfma_bf16(dk[nt], fds, tfrag(sQT, LPT, nt * 16 + c, qp * 32, g));
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kt * 16 + 4 * g + r;
        if (kk < L) {
          a.dK[(row_b + kk) * a.lddk + h * HD + nt * 16 + c] = dk[nt][r];
          a.dV[(row_b + kk) * a.lddv + h * HD + nt * 16 + c] = dv[nt][r];
        }
      }
  }
}

}  // namespace adt
