// General masked attention (either precision, any mask): the bidirectional key-padding attention of BERT4Rec-ADT
// (bert4rec/model/modules.py:76-101: softmax(masked_fill(q k^T / sqrt(hd), mask == 0, -1e9)) -> dropout -> @ v),
// also causal / causal x key-padding, for head sizes 16..64 and L <= 224.  Same residency and C-layout scheme as
// adt_attn_bf16.cuh (whole sequence of one (b, h) in LDS, one 16-query tile per wave, probabilities never leave
// registers), with three differences: (1) the images are typed by the precision -- bf16 for the bf16-operand
// mode, fp32 for the exact mode -- behind one fragment interface; (2) masked scores are REPLACED by `fill`
// (finite for BERT: a fully masked row becomes uniform over all L keys, as in the reference) and get no gradient;
// (3) the backward runs in two phases that re-use one LDS region (phase A: K, V, K^T -> dQ; phase B: Q, dO, Q^T,
// dO^T -> dK, dV), so hd = 64 at L = 200 fits in 160 KB.
#pragma once
#include "adt_attn_bf16.cuh"

namespace adt {

template <int PREC> struct Img;
template <> struct Img<PREC_BF16> {
  typedef __bf16 E;
  typedef bf16x8 F;
  static ADT_DEVICE_INLINE F pack(const float (&v)[8]) { return pack8(v); }
  static ADT_DEVICE_INLINE f32x4 mma(f32x4 acc, const F& a, const F& b) { return mfma_bf16(acc, a, b); }
  static ADT_DEVICE_INLINE F row8(const E* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static ADT_DEVICE_INLINE F slot8(const E* p, int g) {   // p[4g..4g+3], p[16+4g..16+4g+3]
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p + 4 * g);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(p + 16 + 4 * g);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
    return o;
  }
  static ADT_DEVICE_INLINE void put8(E* p, const float (&v)[8]) { *reinterpret_cast<bf16x8*>(p) = pack8(v); }
};
template <> struct Img<PREC_F32> {
  typedef float E;
  typedef Frag8 F;
  static ADT_DEVICE_INLINE F pack(const float (&v)[8]) {
    Frag8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = v[j];
    return f;
  }
  static ADT_DEVICE_INLINE f32x4 mma(f32x4 acc, const F& a, const F& b) { return mma16<PREC_F32>(acc, a, b); }
  static ADT_DEVICE_INLINE F row8(const E* p) { return frag_contig(p); }
  static ADT_DEVICE_INLINE F slot8(const E* p, int g) { return frag_slotc(p, g); }
  static ADT_DEVICE_INLINE void put8(E* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};

struct AttnGenArgs {
  AttnArgs a;            // tensors, B/H/L, causal, scale, dropout (adt_attn.cuh)
  const int* kid;        // optional (B*L) ids: key j of sequence b is masked when kid[b*L + j] <= 0
  float fill;            // value masked scores are replaced with (-1e9 for BERT, -inf for a hard mask)
};

template <int PREC, int HD, int MAXKT>
struct AttnGenLds {
  typedef typename Img<PREC>::E E;
  static constexpr int LP = MAXKT * 16, LPT = LP + 8, RS = HD + 8;
  static constexpr size_t fwd_bytes = (size_t)(LP * RS + HD * LPT) * sizeof(E) + 2 * LP * sizeof(int);
  static constexpr size_t bwd_bytes = (size_t)(2 * LP * RS + 2 * HD * LPT) * sizeof(E) + 2 * LP * sizeof(float) + 2 * LP * sizeof(int);
};

// global fp32 (L x HD slice) -> row image [LP][RS] and/or transposed image [HD][LPT]; rows >= L zero
template <int PREC, int HD, int NTH>
ADT_DEVICE_INLINE void stage_img(typename Img<PREC>::E* rowimg, typename Img<PREC>::E* timg, int LPT, const float* g, int ld, int L, int LP,
                                 float mul) {
  typedef typename Img<PREC>::E E;
  constexpr int RS = HD + 8, V8 = HD / 8;
  for (int i = threadIdx.x; i < LP * V8; i += NTH) {
    const int r = i / V8, c8 = (i % V8) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < L) {
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c8);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(g + (size_t)r * ld + c8 + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= mul;
    }
    if (rowimg) Img<PREC>::put8(rowimg + r * RS + c8, v);
    if (timg) {
#pragma unroll
      for (int j = 0; j < 8; ++j) timg[(c8 + j) * LPT + r] = (E)v[j];
    }
  }
}

// 8 contiguous hd-values (columns kb*32 + 8g ..) of global row `row`, scaled; zero beyond HD or when !ok
template <int PREC, int HD>
ADT_DEVICE_INLINE typename Img<PREC>::F gfrag(const float* base, int ld, int row, bool ok, int kb, int g, float mul) {
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ok && kb * 32 + 8 * g < HD) {
    const float* p = base + (size_t)row * ld + kb * 32 + 8 * g;
    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= mul;
  }
  return Img<PREC>::pack(v);
}

template <int PREC, int HD>
ADT_DEVICE_INLINE typename Img<PREC>::F rfrag_g(const typename Img<PREC>::E* rowimg, int row, int kb, int g) {
  constexpr int RS = HD + 8;
  if (kb * 32 + 8 * g < HD) return Img<PREC>::row8(rowimg + row * RS + kb * 32 + 8 * g);
  const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return Img<PREC>::pack(z);
}

// A query none of whose keys is attendable (a fully padded sequence; under the causal mask also the padded prefix) has
// every score replaced by `fill`: the softmax is uniform over all L keys.  With fill = -1e9, fill + log(L) is not
// representable in fp32, so such rows use 0 instead of `fill` (same uniform probabilities, LSE = log L exactly).
template <int NTH>
ADT_DEVICE_INLINE void mark_dead_rows(int* sDead, const int* sKv, int L, int LP, int causal) {
  for (int q = threadIdx.x; q < LP; q += NTH) {
    int cnt = 0;
    const int lim = causal ? (q < L ? q + 1 : L) : L;
    for (int j = 0; j < lim; ++j) cnt += sKv[j];
    sDead[q] = cnt == 0 ? 1 : 0;
  }
}

template <int PREC, int HD, int MAXKT, int NW, bool CSK = false>
__global__ __launch_bounds__(NW * 64) void k_attn_gen_fwd(AttnGenArgs ga) {
  adt_prefetch_kernargs<sizeof(AttnGenArgs) <= 512 ? sizeof(AttnGenArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef Img<PREC> I;
  typedef typename I::E E;
  typedef typename I::F F;
  const AttnArgs& a = ga.a;
  constexpr int LP = MAXKT * 16, LPT = LP + 8, RS = HD + 8, NT = HD / 16, KB = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  E* sK = reinterpret_cast<E*>(smem_raw);    // [LP][RS]
  E* sVT = sK + LP * RS;                      // [HD][LPT]
  int* sKv = reinterpret_cast<int*>(sVT + HD * LPT);   // key validity (1 = attendable)
  int* sDead = sKv + LP;                                // query has no attendable key
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L;
  const size_t row_b = (size_t)b * L;
  stage_img<PREC, HD, NW * 64>(sK, nullptr, LPT, a.K + row_b * a.ldk + h * HD, a.ldk, L, LP, 1.0f);
  stage_img<PREC, HD, NW * 64>(nullptr, sVT, LPT, a.V + row_b * a.ldv + h * HD, a.ldv, L, LP, 1.0f);
  for (int i = threadIdx.x; i < LP; i += NW * 64) sKv[i] = (i < L && (!ga.kid || ga.kid[row_b + i] > 0)) ? 1 : 0;
  __syncthreads();
  mark_dead_rows<NW * 64>(sDead, sKv, L, LP, a.causal);
  __syncthreads();
  const uint32_t key_rng = drop_key(a.drop);
  const int nqt = (L + 15) / 16;
  // CSK (chosen by the host for a causal mask without key padding and fill <= -1e9): key tiles above the diagonal are fully
  // masked with probability exactly 0 and the diagonal is always attendable, so they are skipped instead of masked.  A template
  // flag: as a runtime test it cost the bidirectional (BERT) instantiations 6 %.
  for (int qt = w; qt < nqt; qt += NW) {
    const int nkt = CSK ? qt + 1 : nqt;
    const int q = qt * 16 + c;
    const float fill_q = sDead[q] ? 0.f : ga.fill;
    F fq[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) fq[kb] = gfrag<PREC, HD>(a.Q + row_b * a.ldq + h * HD, a.ldq, q, q < L, kb, g, a.scale);
    // Two sweeps over the key tiles with the scores recomputed in the second, not kept (see k_attn_fwd_bf16): the score array
    // (56-64 registers) was what limited this kernel to one workgroup per CU at hd = 64 and made it spill at hd = 128.
    auto score = [&](int kt, f32x4& sc) {
      sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) sc = I::mma(sc, rfrag_g<PREC, HD>(sK, kt * 16 + c, kb, g), fq[kb]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const bool masked = (a.causal && key > q) || !sKv[key];
        const float v = masked ? fill_q : sc[r];
        sc[r] = key < L ? v : -INFINITY;
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
    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kp = 0; 2 * kp < nkt; ++kp) {
      float pv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kt = 2 * kp + t;
        f32x4 sc = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt < nkt) score(kt, sc);
        // the quad's four keys are consecutive elements of probability row q: one hash for the four when L % 4 == 0
        const uint32_t kbits = (a.drop.thr && kt < nkt) ? adt_keep4_any(key_rng, idx_q + (uint32_t)(kt * 16 + 4 * g), a.drop.thr, (L & 3) == 0) : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(sc[r] - m);        // exp(-inf) = 0 for absent keys
          sum += e;
          float p = e;
          if (a.drop.thr) p = ((kbits >> r) & 1u) ? e * a.drop.scale : 0.f;
          pv[4 * t + r] = p;
        }
      }
      const F fp = I::pack(pv);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] = I::mma(o[nt], fp, I::slot8(sVT + (nt * 16 + c) * LPT + kp * 32, g));
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
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qt * 16 + 4 * g + r;
        if (qq < L) a.O[(row_b + qq) * a.ldo + h * HD + nt * 16 + c] = o[nt][r];
      }
  }
}

template <int PREC, int HD, int MAXKT, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_gen_bwd(AttnGenArgs ga) {
  adt_prefetch_kernargs<sizeof(AttnGenArgs) <= 512 ? sizeof(AttnGenArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef Img<PREC> I;
  typedef typename I::E E;
  typedef typename I::F F;
  const AttnArgs& a = ga.a;
  constexpr int LP = MAXKT * 16, LPT = LP + 8, RS = HD + 8, NT = HD / 16, KB = (HD + 31) / 32, V8 = HD / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  E* sR0 = reinterpret_cast<E*>(smem_raw);   // phase A: K    | phase B: Q (scaled)
  E* sR1 = sR0 + LP * RS;                     // phase A: V    | phase B: dO
  E* sT0 = sR1 + LP * RS;                     // phase A: K^T  | phase B: Q^T (scaled)
  E* sT1 = sT0 + HD * LPT;                    //               | phase B: dO^T
  float* sLse = reinterpret_cast<float*>(sT1 + HD * LPT);   // +inf for padded queries -> P = 0
  float* sDelta = sLse + LP;                                 // rowsum(dO * O)
  int* sKv = reinterpret_cast<int*>(sDelta + LP);
  int* sDead = sKv + LP;
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L;
  const size_t row_b = (size_t)b * L;
  const float* gQ = a.Q + row_b * a.ldq + h * HD;
  const float* gK = a.K + row_b * a.ldk + h * HD;
  const float* gV = a.V + row_b * a.ldv + h * HD;
  const float* gdO = a.dO + row_b * a.lddo + h * HD;
  const float* gO = a.O + row_b * a.ldo + h * HD;
  // ---- phase A staging: K, V, K^T; delta and LSE; key validity
  stage_img<PREC, HD, NW * 64>(sR0, sT0, LPT, gK, a.ldk, L, LP, 1.0f);
  stage_img<PREC, HD, NW * 64>(sR1, nullptr, LPT, gV, a.ldv, L, LP, 1.0f);
  for (int i = threadIdx.x; i < LP * V8; i += NW * 64) {
    const int r = i / V8, c8 = (i % V8) * 8;
    float part = 0.f;
    if (r < L) {
      const float4 d0 = *reinterpret_cast<const float4*>(gdO + (size_t)r * a.lddo + c8), d1 = *reinterpret_cast<const float4*>(gdO + (size_t)r * a.lddo + c8 + 4);
      const float4 o0 = *reinterpret_cast<const float4*>(gO + (size_t)r * a.ldo + c8), o1 = *reinterpret_cast<const float4*>(gO + (size_t)r * a.ldo + c8 + 4);
      part = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w;
    }
#pragma unroll
    for (int off = V8 / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((i % V8) == 0) {
      sDelta[r] = part;
      sLse[r] = (r < L) ? a.LSE[(size_t)bh * L + r] : INFINITY;
    }
  }
  for (int i = threadIdx.x; i < LP; i += NW * 64) sKv[i] = (i < L && (!ga.kid || ga.kid[row_b + i] > 0)) ? 1 : 0;
  __syncthreads();
  mark_dead_rows<NW * 64>(sDead, sKv, L, LP, a.causal);
  __syncthreads();
  const uint32_t key_rng = drop_key(a.drop);
  const uint32_t idx_bh = (uint32_t)(bh + a.bh_offset) * (uint32_t)L;
  const int nqt = (L + 15) / 16;

  // ---- pass A: dQ (wave owns a query tile; its Q / dO fragments come straight from global memory)
  for (int qt = w; qt < nqt; qt += NW) {
    const int nkt = nqt;
    const int q = qt * 16 + c;
    const float lse_q = sLse[q], delta_q = sDelta[q];
    const float fill_q = sDead[q] ? 0.f : ga.fill;
    F fq[KB], fdo[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      fq[kb] = gfrag<PREC, HD>(gQ, a.ldq, q, q < L, kb, g, a.scale);
      fdo[kb] = gfrag<PREC, HD>(gdO, a.lddo, q, q < L, kb, g, 1.0f);
    }
    f32x4 dq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dq[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t idx_q = (idx_bh + (uint32_t)q) * (uint32_t)L;
#pragma unroll 1
    for (int kp = 0; 2 * kp < nkt; ++kp) {
      float dsv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kt = 2 * kp + t;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        if (kt < nkt) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            s = I::mma(s, rfrag_g<PREC, HD>(sR0, kt * 16 + c, kb, g), fq[kb]);
            dp = I::mma(dp, rfrag_g<PREC, HD>(sR1, kt * 16 + c, kb, g), fdo[kb]);
          }
        }
        const uint32_t kbits = a.drop.thr ? adt_keep4_any(key_rng, idx_q + (uint32_t)(kt * 16 + 4 * g), a.drop.thr, (L & 3) == 0) : 0u;      // see k_attn_gen_fwd
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;   // < LP
          const bool masked = (a.causal && key > q) || !sKv[key];
          const float sv = masked ? fill_q : s[r];
          const float p = (kt < nkt && key < L) ? __expf(sv - lse_q) : 0.f;
          float d = dp[r];
          if (a.drop.thr) d = ((kbits >> r) & 1u) ? d * a.drop.scale : 0.f;
          dsv[4 * t + r] = masked ? 0.f : p * (d - delta_q);
        }
      }
      const F fds = I::pack(dsv);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dq[nt] = I::mma(dq[nt], fds, I::slot8(sT0 + (nt * 16 + c) * LPT + kp * 32, g));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qt * 16 + 4 * g + r;
        if (qq < L) a.dQ[(row_b + qq) * a.lddq + h * HD + nt * 16 + c] = dq[nt][r] * a.scale;
      }
  }

  // ---- phase B staging: Q (scaled), dO and their transposes over the same LDS
  __syncthreads();
  stage_img<PREC, HD, NW * 64>(sR0, sT0, LPT, gQ, a.ldq, L, LP, a.scale);
  stage_img<PREC, HD, NW * 64>(sR1, sT1, LPT, gdO, a.lddo, L, LP, 1.0f);
  __syncthreads();

  // ---- pass B: dK, dV (wave owns a key tile; its K / V fragments come straight from global memory)
  for (int kt = w; kt < nqt; kt += NW) {
    const int key = kt * 16 + c;
    const bool key_ok = key < L;
    const bool key_attend = sKv[key] != 0;
    F fk[KB], fv[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      fk[kb] = gfrag<PREC, HD>(gK, a.ldk, key, key_ok, kb, g, 1.0f);
      fv[kb] = gfrag<PREC, HD>(gV, a.ldv, key, key_ok, kb, g, 1.0f);
    }
    f32x4 dk[NT], dv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      dk[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int qp = 0; 2 * qp < nqt; ++qp) {
      float pv[8], dsv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qt = 2 * qp + t;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        if (qt < nqt) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            s = I::mma(s, rfrag_g<PREC, HD>(sR0, qt * 16 + c, kb, g), fk[kb]);
            dp = I::mma(dp, rfrag_g<PREC, HD>(sR1, qt * 16 + c, kb, g), fv[kb]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = qt * 16 + 4 * g + r;     // < LP
          const bool masked = (a.causal && key > qq) || !key_attend;
          const float sv = masked ? (sDead[qq] ? 0.f : ga.fill) : s[r];
          const float p = (qt < nqt && key_ok) ? __expf(sv - sLse[qq]) : 0.f;   // sLse = +inf for padded queries
          float ks = 1.0f;
          if (a.drop.thr) ks = adt_keep(key_rng, (idx_bh + (uint32_t)qq) * (uint32_t)L + (uint32_t)key, a.drop.thr) ? a.drop.scale : 0.f;
          pv[4 * t + r] = p * ks;
          dsv[4 * t + r] = masked ? 0.f : p * (dp[r] * ks - sDelta[qq]);
        }
      }
      const F fp = I::pack(pv), fds = I::pack(dsv);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        dv[nt] = I::mma(dv[nt], fp, I::slot8(sT1 + (nt * 16 + c) * LPT + qp * 32, g));
        dk[nt] = I::mma(dk[nt], fds, I::slot8(sT0 + (nt * 16 + c) * LPT + qp * 32, g));
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

// Chunked backward for head sizes whose (b, h) images do not fit in LDS at once (hd = 128, the d = 256 / 2-head template of
// sasrec/templates/ml-1m.json): the resident side is staged in NCH chunks of LP / NCH rows -- keys (K, V, K^T) in pass A,
// queries (Q, dO, Q^T, dO^T) in pass B -- while every wave keeps the dQ (pass A) or dK, dV (pass B) accumulators of its (up
// to two) tiles in registers across the chunks.  Same arithmetic as k_attn_gen_bwd (P from the saved LSE, no cross-wave sums).
template <int PREC, int HD, int MAXKT, int NCH>
struct AttnChunkLds {
  typedef typename Img<PREC>::E E;
  static constexpr int LP = MAXKT * 16, LPC = LP / NCH, LPTC = LPC + 8, RS = HD + 8;
  static constexpr size_t bwd_bytes = (size_t)(2 * LPC * RS + 2 * HD * LPTC) * sizeof(E) + 2 * LP * sizeof(float) + 2 * LP * sizeof(int);
};

template <int PREC, int HD, int MAXKT, int NCH, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_gen_bwd_chunked(AttnGenArgs ga) {
  adt_prefetch_kernargs<sizeof(AttnGenArgs) <= 512 ? sizeof(AttnGenArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  typedef Img<PREC> I;
  typedef typename I::E E;
  typedef typename I::F F;
  const AttnArgs& a = ga.a;
  constexpr int LP = MAXKT * 16, LPC = LP / NCH, LPTC = LPC + 8, RS = HD + 8, NT = HD / 16, KB = (HD + 31) / 32, V8 = HD / 8;
  constexpr int QPW = (MAXKT + NW - 1) / NW;       // tiles a wave owns
  constexpr int TPC = LPC / 16;                    // 16-row tiles per chunk (even)
  static_assert(LPC % 32 == 0, "chunks hold whole tile pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  E* sR0 = reinterpret_cast<E*>(smem_raw);   // pass A: K chunk   | pass B: Q chunk (scaled)
  E* sR1 = sR0 + LPC * RS;                    // pass A: V chunk   | pass B: dO chunk
  E* sT0 = sR1 + LPC * RS;                    // pass A: K^T chunk | pass B: Q^T chunk (scaled)
  E* sT1 = sT0 + HD * LPTC;                   //                   | pass B: dO^T chunk
  float* sLse = reinterpret_cast<float*>(sT1 + HD * LPTC);
  float* sDelta = sLse + LP;
  int* sKv = reinterpret_cast<int*>(sDelta + LP);
  int* sDead = sKv + LP;
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int L = a.L;
  const size_t row_b = (size_t)b * L;
  const float* gQ = a.Q + row_b * a.ldq + h * HD;
  const float* gK = a.K + row_b * a.ldk + h * HD;
  const float* gV = a.V + row_b * a.ldv + h * HD;
  const float* gdO = a.dO + row_b * a.lddo + h * HD;
  const float* gO = a.O + row_b * a.ldo + h * HD;
  for (int i = threadIdx.x; i < LP * V8; i += NW * 64) {
    const int r = i / V8, c8 = (i % V8) * 8;
    float part = 0.f;
    if (r < L) {
      const float4 d0 = *reinterpret_cast<const float4*>(gdO + (size_t)r * a.lddo + c8), d1 = *reinterpret_cast<const float4*>(gdO + (size_t)r * a.lddo + c8 + 4);
      const float4 o0 = *reinterpret_cast<const float4*>(gO + (size_t)r * a.ldo + c8), o1 = *reinterpret_cast<const float4*>(gO + (size_t)r * a.ldo + c8 + 4);
      part = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w;
    }
#pragma unroll
    for (int off = V8 / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((i % V8) == 0) {
      sDelta[r] = part;
      sLse[r] = (r < L) ? a.LSE[(size_t)bh * L + r] : INFINITY;
    }
  }
  for (int i = threadIdx.x; i < LP; i += NW * 64) sKv[i] = (i < L && (!ga.kid || ga.kid[row_b + i] > 0)) ? 1 : 0;
  __syncthreads();
  mark_dead_rows<NW * 64>(sDead, sKv, L, LP, a.causal);
  const uint32_t key_rng = drop_key(a.drop);
  const uint32_t idx_bh = (uint32_t)(bh + a.bh_offset) * (uint32_t)L;
  const int nqt = (L + 15) / 16;
  const bool csk = a.causal && ga.kid == nullptr && ga.fill <= -1e9f;     // see k_attn_gen_fwd

  // ---- pass A: dQ, key chunks resident --------------------------------------------------------------------------------
  {
    f32x4 dq[QPW][NT];
#pragma unroll
    for (int t = 0; t < QPW; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dq[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const int r0 = ch * LPC;
      if (r0 >= L) break;
      __syncthreads();
      const int Lc = L - r0 < LPC ? L - r0 : LPC;
      stage_img<PREC, HD, NW * 64>(sR0, sT0, LPTC, gK + (size_t)r0 * a.ldk, a.ldk, Lc, LPC, 1.0f);
      stage_img<PREC, HD, NW * 64>(sR1, nullptr, LPTC, gV + (size_t)r0 * a.ldv, a.ldv, Lc, LPC, 1.0f);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < QPW; ++t) {
        const int qt = w + t * NW;
        if (qt >= nqt) continue;
        const int q = qt * 16 + c;
        const float lse_q = sLse[q], delta_q = sDelta[q];
        const float fill_q = sDead[q] ? 0.f : ga.fill;
        F fq[KB], fdo[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          fq[kb] = gfrag<PREC, HD>(gQ, a.ldq, q, q < L, kb, g, a.scale);
          fdo[kb] = gfrag<PREC, HD>(gdO, a.lddo, q, q < L, kb, g, 1.0f);
        }
        const uint32_t idx_q = (idx_bh + (uint32_t)q) * (uint32_t)L;
#pragma unroll 1
        for (int kp = 0; kp < TPC / 2; ++kp) {
          if (r0 + kp * 32 >= L) break;
          if (csk && r0 + kp * 32 > qt * 16 + 15) break;      // every key of the pair lies above the diagonal of this query tile
          float dsv[8];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int ktl = 2 * kp + tt;                  // tile inside the chunk
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
              sacc = I::mma(sacc, rfrag_g<PREC, HD>(sR0, ktl * 16 + c, kb, g), fq[kb]);
              dp = I::mma(dp, rfrag_g<PREC, HD>(sR1, ktl * 16 + c, kb, g), fdo[kb]);
            }
            const uint32_t kbits = a.drop.thr ? adt_keep4_any(key_rng, idx_q + (uint32_t)(r0 + ktl * 16 + 4 * g), a.drop.thr, (L & 3) == 0 && (r0 & 3) == 0) : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = r0 + ktl * 16 + 4 * g + r;  // < LP
              const bool masked = (a.causal && key > q) || !sKv[key];
              const float sv = masked ? fill_q : sacc[r];
              const float p = key < L ? __expf(sv - lse_q) : 0.f;
              float d = dp[r];
              if (a.drop.thr) d = ((kbits >> r) & 1u) ? d * a.drop.scale : 0.f;
              dsv[4 * tt + r] = masked ? 0.f : p * (d - delta_q);
            }
          }
          const F fds = I::pack(dsv);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) dq[t][nt] = I::mma(dq[t][nt], fds, I::slot8(sT0 + (nt * 16 + c) * LPTC + kp * 32, g));
        }
      }
    }
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
      const int qt = w + t * NW;
      if (qt >= nqt) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = qt * 16 + 4 * g + r;
          if (qq < L) a.dQ[(row_b + qq) * a.lddq + h * HD + nt * 16 + c] = dq[t][nt][r] * a.scale;
        }
    }
  }

  // ---- pass B: dK, dV, query chunks resident ---------------------------------------------------------------------------
  {
    f32x4 dk[QPW][NT], dv[QPW][NT];
#pragma unroll
    for (int t = 0; t < QPW; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { dk[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const int r0 = ch * LPC;
      if (r0 >= L) break;
      __syncthreads();
      const int Lc = L - r0 < LPC ? L - r0 : LPC;
      stage_img<PREC, HD, NW * 64>(sR0, sT0, LPTC, gQ + (size_t)r0 * a.ldq, a.ldq, Lc, LPC, a.scale);
      stage_img<PREC, HD, NW * 64>(sR1, sT1, LPTC, gdO + (size_t)r0 * a.lddo, a.lddo, Lc, LPC, 1.0f);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < QPW; ++t) {
        const int kt = w + t * NW;
        if (kt >= nqt) continue;
        const int key = kt * 16 + c;
        const bool key_ok = key < L;
        const bool key_attend = sKv[key] != 0;
        F fk[KB], fv[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          fk[kb] = gfrag<PREC, HD>(gK, a.ldk, key, key_ok, kb, g, 1.0f);
          fv[kb] = gfrag<PREC, HD>(gV, a.ldv, key, key_ok, kb, g, 1.0f);
        }
#pragma unroll 1
        for (int qp = 0; qp < TPC / 2; ++qp) {
          if (r0 + qp * 32 >= L) break;
          if (csk && r0 + qp * 32 + 31 < kt * 16) continue;   // every query of the pair lies before this key tile
          float pv[8], dsv[8];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int qtl = 2 * qp + tt;
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
              sacc = I::mma(sacc, rfrag_g<PREC, HD>(sR0, qtl * 16 + c, kb, g), fk[kb]);
              dp = I::mma(dp, rfrag_g<PREC, HD>(sR1, qtl * 16 + c, kb, g), fv[kb]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qq = r0 + qtl * 16 + 4 * g + r;     // < LP
              const bool masked = (a.causal && key > qq) || !key_attend;
              const float sv = masked ? (sDead[qq] ? 0.f : ga.fill) : sacc[r];
              const float p = key_ok ? __expf(sv - sLse[qq]) : 0.f;   // sLse = +inf for padded queries
              float ks = 1.0f;
              if (a.drop.thr) ks = adt_keep(key_rng, (idx_bh + (uint32_t)qq) * (uint32_t)L + (uint32_t)key, a.drop.thr) ? a.drop.scale : 0.f;
              pv[4 * tt + r] = p * ks;
              dsv[4 * tt + r] = masked ? 0.f : p * (dp[r] * ks - sDelta[qq]);
            }
          }
          const F fp = I::pack(pv), fds = I::pack(dsv);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            dv[t][nt] = I::mma(dv[t][nt], fp, I::slot8(sT1 + (nt * 16 + c) * LPTC + qp * 32, g));
            dk[t][nt] = I::mma(dk[t][nt], fds, I::slot8(sT0 + (nt * 16 + c) * LPTC + qp * 32, g));
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
      const int kt = w + t * NW;
      if (kt >= nqt) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = kt * 16 + 4 * g + r;
          if (kk < L) {
            a.dK[(row_b + kk) * a.lddk + h * HD + nt * 16 + c] = dk[t][nt][r];
            a.dV[(row_b + kk) * a.lddv + h * HD + nt * 16 + c] = dv[t][nt][r];
          }
        }
    }
  }
}

}  // namespace adt
