// Causal attention BACKWARD, one workgroup per sequence (all heads), bf16 MFMA operands -- replaces k_attn_bwd_bf16 (one workgroup
// per (b, h), four row images + three TRANSPOSED images per head, written with two-byte LDS stores) for d = H * hd = 64.
//
// LDS holds ONE natural-order row image [token][64] (bf16, stride 72) each of Q (pre-scaled by log2(e)/sqrt(hd)), K, V and dO for
// the whole sequence and all heads.  Operands that contract over FEATURES read a row of an image in slot order (two 8-byte reads:
// columns 4g .. 4g+3 and 16 + 4g .. of the head's 32-feature block); operands that contract over TOKENS (K^T for dQ, dO^T for dV,
// Q^T for dK) come from the SAME images through ds_read_b64_tr_b16 (adt_tt.cuh: tt_trfrag) -- no transposed copies, staging is
// plain 16-byte stores.  Algorithm as before (adt_attn.cuh): P is recomputed from the saved log-sum-exp, pass A (a wave owns a
// query tile) accumulates dQ, pass B (a wave owns a key tile) accumulates dK and dV, no cross-wave sums => bitwise reproducible.
// All gradients come out as TRANSPOSED tiles (feature on the accumulator row, token on the lane): 16-byte global stores.
#pragma once
#include "adt_attn.cuh"
#include "adt_tt.cuh"

namespace adt {

constexpr int SAB_NW = 16;      // 16 waves = 4 per SIMD (the kernel needs <= 128 VGPRs): with 13 tiles at L = 200 every wave owns at most one tile per pass
// Tile of wave w in a pass whose tiles are sorted heaviest-first (index j = 0 the heaviest): waves w, w + 4, w + 8, w + 12 share a SIMD and
// the passes are bound by each SIMD's instruction issue, so the tiles are dealt to the four SIMD classes in snake order (see tq_tile12)
ADT_DEVICE_INLINE int sab_rank16(int w) {
  const int cl = w & 3, k = w >> 2;
  return 4 * k + ((k & 1) ? 3 - cl : cl);
}

__host__ __device__ inline int sab_rows(int L) { return (L + 31) / 32 * 32; }       // image rows: whole tile pairs
__host__ __device__ inline size_t sab_lds_bytes(int L, int H) {
  const size_t R = (size_t)((L + 31) / 32 * 32);
  return 4 * R * TT_RS * 2 + (size_t)H * R * 8 * 4 + 2 * (size_t)H * R * 4;
}

// 8 features of image row `row` in slot order for head h (block kb of the head): {4g .. 4g+3} and {16 + 4g ..} of the 32-feature block;
// a 16-wide head fills the low four slots only (the high four are zero in BOTH operands of its products)
template <int HD>
ADT_DEVICE_INLINE bf16x8 sab_rowfrag(const __bf16* img, int row, int h, int kb, int g) {
  const __bf16* p = img + row * TT_RS + h * HD + 32 * kb + 4 * g;
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
  bf16x4 hi;
  if constexpr (HD == 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) hi[j] = (__bf16)0.f;
  } else {
    hi = *reinterpret_cast<const bf16x4*>(p + 16);
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
  return o;
}

// ---- the two passes over one (tile, head), shared by the stand-alone kernel below and the fused layer backward (adt_seqbwd_tt.cuh) ----
// MODE: 0 no dropout, 1 keep bits saved by the forward, 2 keep decisions recomputed from the hash RNG -- a compile-time choice: as
// run-time branches inside the element loops the three variants cost ~100 branch instructions per key-tile pair.
//
// Dropout scale: the dO operands / image CARRY the factor 1 / (1 - p) (the callers scale them once when they build them), so
// dp = dO V^T arrives scaled, a dropped element is an AND with its keep bit (v_bfe_i32 + v_and_b32, no compare / select / multiply) and
// the probabilities that multiply dO for dV stay unscaled.  delta = rowsum(dO * O) is taken from the UNSCALED dO by the callers.
//
// Only the pair of key (query) tiles that holds the diagonal tile needs the per-element causal mask: it is peeled out of the loops
// (as `if (edge)` inside them the compiler if-converted it into two selects per element of EVERY pair).

// p where bit `pos` (a lane variable) of `word` is set, else +0.0 (one statement: see tt_keep_if_bit; p comes from v_exp_f32, never from an MFMA)
ADT_DEVICE_INLINE float sab_keep1(float p, uint32_t word, uint32_t pos) {
  float r;
  asm("v_bfe_i32 %0, %1, %2, 1\n\tv_and_b32 %0, %0, %3" : "=&v"(r) : "v"(word), "v"(pos), "v"(p));
  return r;
}

// pass A: query tile qt on the lane (q = 16 qt + c), keys on the accumulator rows.  fq / fdo: this lane's query and (scaled) dO rows of head h
// in slot order (fq pre-scaled by log2(e)/sqrt(hd)); lse_q in the log2 domain; mrow: the 8 keep-bit words of query q (MODE 1).
// Result: dq[nt] = rows (features 16 nt + 4g + r of the head) x column (query c) of dS K, NOT yet multiplied by 1/sqrt(hd).
template <int HD, int MODE, bool EDGE>
ADT_DEVICE_INLINE void sab_pair_a(const __bf16* sK, const __bf16* sV, const bf16x8* fq, const bf16x8* fdo, float lse_q, float delta_q, const uint32_t* mrow,
                                  int kp, int qt, int h, const DropCfg& drop, uint32_t key_rng, uint32_t idx_q, int c, int g, f32x4 (&dq)[HD / 16]) {
  constexpr int NT = HD / 16, KB = (HD + 31) / 32;
  f32x4 s[2], dp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int kt = 2 * kp + t;
    s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    dp[t] = s[t];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {          // rows beyond the causal range are zero-filled or masked below
      s[t] = mfma_bf16(s[t], sab_rowfrag<HD>(sK, kt * 16 + c, h, kb, g), fq[kb]);
      dp[t] = mfma_bf16(dp[t], sab_rowfrag<HD>(sV, kt * 16 + c, h, kb, g), fdo[kb]);
    }
  }
  uint32_t mword = 0u;
  if constexpr (MODE == 1) mword = mrow[kp] >> (4 * g);
  // dS = P o (keep o dP - delta) = (keep o P) o dP - P delta: the keep bit is applied to P (a v_exp result), dP stays an ordinary operand
  f32x4 ds[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int kt = 2 * kp + t;
    float p[4], pk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = __builtin_amdgcn_exp2f(s[t][r] - lse_q);
      if constexpr (EDGE) p[r] = (kt < qt || (kt == qt && 4 * g + r <= c)) ? p[r] : 0.f;
      pk[r] = p[r];
      if constexpr (MODE == 2) pk[r] = adt_keep(key_rng, idx_q + (uint32_t)(kt * 16 + 4 * g + r), drop.thr) ? p[r] : 0.f;
    }
    if constexpr (MODE == 1) {
      if (t == 0) {
        pk[0] = tt_keep_if_bit<0>(p[0], mword); pk[1] = tt_keep_if_bit<1>(p[1], mword); pk[2] = tt_keep_if_bit<2>(p[2], mword); pk[3] = tt_keep_if_bit<3>(p[3], mword);
      } else {
        pk[0] = tt_keep_if_bit<16>(p[0], mword); pk[1] = tt_keep_if_bit<17>(p[1], mword); pk[2] = tt_keep_if_bit<18>(p[2], mword); pk[3] = tt_keep_if_bit<19>(p[3], mword);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[t][r] = pk[r] * dp[t][r] - p[r] * delta_q;
  }
  const bf16x8 fds = tt_pack(ds[0], ds[1]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dq[nt] = mfma_bf16(dq[nt], tt_trfrag(sK, kp * 32, h * HD + nt * 16, c, g), fds);
}

template <int HD, int MODE>
ADT_DEVICE_INLINE void sab_pass_a(const __bf16* sK, const __bf16* sV, const bf16x8* fq, const bf16x8* fdo, float lse_q, float delta_q,
                                  const uint32_t* mrow, int qt, int h, const DropCfg& drop, uint32_t key_rng, uint32_t idx_q, int c, int g,
                                  f32x4 (&dq)[HD / 16]) {
  constexpr int NT = HD / 16;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dq[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nfull = qt >> 1;                    // pairs strictly below the diagonal tile; pair nfull holds it (and, for even qt, a tile beyond it)
#pragma unroll 2
  for (int kp = 0; kp < nfull; ++kp) sab_pair_a<HD, MODE, false>(sK, sV, fq, fdo, lse_q, delta_q, mrow, kp, qt, h, drop, key_rng, idx_q, c, g, dq);
  sab_pair_a<HD, MODE, true>(sK, sV, fq, fdo, lse_q, delta_q, mrow, nfull, qt, h, drop, key_rng, idx_q, c, g, dq);
}

// pass B: key tile kt on the lane (key = 16 kt + c), queries on the accumulator rows.  fk / fv: this lane's key and value rows of head h;
// lse_h / del_h: the head's per-query log2-domain log-sum-exp (+inf for padded queries) and delta; sM_h: [rows][8] keep bits (MODE 1).
// Results: dk = dS^T Q (carries the Q image's factor log2(e)/sqrt(hd)), dv = P'^T dO (the dO image carries 1 / (1 - p)); rows = features, column = key c.
template <int HD, int MODE, bool EDGE>
ADT_DEVICE_INLINE void sab_pair_b(const __bf16* sQ, const __bf16* sdO, const bf16x8* fk, const bf16x8* fv, const float* lse_h, const float* del_h,
                                  const uint32_t* sM_h, int qp, int kt, int h, const DropCfg& drop, uint32_t key_rng, uint32_t idx_bh, int L, int c, int g,
                                  f32x4 (&dk)[HD / 16], f32x4 (&dv)[HD / 16]) {
  constexpr int NT = HD / 16, KB = (HD + 31) / 32;
  const int key = kt * 16 + c;
  f32x4 s[2], dp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qt = 2 * qp + t;
    s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    dp[t] = s[t];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      s[t] = mfma_bf16(s[t], sab_rowfrag<HD>(sQ, qt * 16 + c, h, kb, g), fk[kb]);
      dp[t] = mfma_bf16(dp[t], sab_rowfrag<HD>(sdO, qt * 16 + c, h, kb, g), fv[kb]);
    }
  }
  f32x4 pv[2], ds[2];
  const uint32_t bitpos = (uint32_t)(16 * (kt & 1) + c);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qt = 2 * qp + t;
    const float4 lse4 = *reinterpret_cast<const float4*>(lse_h + qt * 16 + 4 * g);      // +inf for padded queries -> p = 0
    const float4 del4 = *reinterpret_cast<const float4*>(del_h + qt * 16 + 4 * g);
    const float lq[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dq4[4] = {del4.x, del4.y, del4.z, del4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qt * 16 + 4 * g + r;
      float p = __builtin_amdgcn_exp2f(s[t][r] - lq[r]);
      if constexpr (EDGE) p = (qt > kt || (qt == kt && c <= 4 * g + r)) ? p : 0.f;
      float pk = p;
      if constexpr (MODE == 1) pk = sab_keep1(p, sM_h[(size_t)qq * 8 + (kt >> 1)], bitpos);
      if constexpr (MODE == 2) pk = adt_keep(key_rng, (idx_bh + (uint32_t)qq) * (uint32_t)L + (uint32_t)key, drop.thr) ? p : 0.f;
      pv[t][r] = pk;
      ds[t][r] = pk * dp[t][r] - p * dq4[r];          // dS = (keep o P) o dP - P delta
    }
  }
  const bf16x8 fp = tt_pack(pv[0], pv[1]), fds = tt_pack(ds[0], ds[1]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    dv[nt] = mfma_bf16(dv[nt], tt_trfrag(sdO, qp * 32, h * HD + nt * 16, c, g), fp);
    dk[nt] = mfma_bf16(dk[nt], tt_trfrag(sQ, qp * 32, h * HD + nt * 16, c, g), fds);
  }
}

template <int HD, int MODE>
ADT_DEVICE_INLINE void sab_pass_b(const __bf16* sQ, const __bf16* sdO, const bf16x8* fk, const bf16x8* fv, const float* lse_h, const float* del_h,
                                  const uint32_t* sM_h, int kt, int nqt, int h, const DropCfg& drop, uint32_t key_rng, uint32_t idx_bh, int L,
                                  int c, int g, f32x4 (&dk)[HD / 16], f32x4 (&dv)[HD / 16]) {
  constexpr int NT = HD / 16;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    dk[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int q0 = kt >> 1;                       // the pair that holds the diagonal tile (and, for odd kt, the tile above it, fully masked)
  sab_pair_b<HD, MODE, true>(sQ, sdO, fk, fv, lse_h, del_h, sM_h, q0, kt, h, drop, key_rng, idx_bh, L, c, g, dk, dv);
#pragma unroll 2
  for (int qp = q0 + 1; 2 * qp < nqt; ++qp) sab_pair_b<HD, MODE, false>(sQ, sdO, fk, fv, lse_h, del_h, sM_h, qp, kt, h, drop, key_rng, idx_bh, L, c, g, dk, dv);
}

// pass B for ALL heads of a key tile in one sweep over the query pairs: the heads are independent chains (LDS reads -> MFMA -> exp -> MFMA),
// and at two waves per SIMD one chain at a time leaves the wave waiting on its own latencies
template <int HD, int MODE>
ADT_DEVICE_INLINE void sab_pass_b_heads(const __bf16* sQ, const __bf16* sdO, const bf16x8* fk, const bf16x8* fv, const float* sLse, const float* sDelta,
                                        const uint32_t* sM, int R, int kt, int nqt, const DropCfg& drop, uint32_t key_rng, uint32_t bh0, int L,
                                        int c, int g, f32x4 (&dk)[64 / 16], f32x4 (&dv)[64 / 16]) {
  constexpr int H = 64 / HD, NT = HD / 16, KB = (HD + 31) / 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int q0 = kt >> 1;
#pragma unroll
  for (int h = 0; h < H; ++h)
    sab_pair_b<HD, MODE, true>(sQ, sdO, fk + h * KB, fv + h * KB, sLse + h * R, sDelta + h * R, sM + (size_t)h * R * 8, q0, kt, h, drop, key_rng,
                               (bh0 + (uint32_t)h) * (uint32_t)L, L, c, g, *reinterpret_cast<f32x4 (*)[NT]>(&dk[h * NT]), *reinterpret_cast<f32x4 (*)[NT]>(&dv[h * NT]));
#pragma unroll 1
  for (int qp = q0 + 1; 2 * qp < nqt; ++qp) {
#pragma unroll
    for (int h = 0; h < H; ++h)
      sab_pair_b<HD, MODE, false>(sQ, sdO, fk + h * KB, fv + h * KB, sLse + h * R, sDelta + h * R, sM + (size_t)h * R * 8, qp, kt, h, drop, key_rng,
                                  (bh0 + (uint32_t)h) * (uint32_t)L, L, c, g, *reinterpret_cast<f32x4 (*)[NT]>(&dk[h * NT]), *reinterpret_cast<f32x4 (*)[NT]>(&dv[h * NT]));
  }
}

template <int HD, int MODE>
__global__ __launch_bounds__(SAB_NW * 64) void k_seq_attn_bwd(AttnArgs a) {
  adt_prefetch_kernargs<(sizeof(AttnArgs) + 63) / 64 * 64 <= 512 ? sizeof(AttnArgs) : 512>();      // adt_common.cuh
  constexpr int H = 64 / HD, NT = HD / 16, KB = (HD + 31) / 32, MAXKT = 14, NW = SAB_NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, R = sab_rows(L);
  __bf16* sQ = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sK = sQ + R * TT_RS;
  __bf16* sV = sK + R * TT_RS;
  __bf16* sdO = sV + R * TT_RS;
  uint32_t* sM = reinterpret_cast<uint32_t*>(sdO + R * TT_RS);     // [H][R][8] dropout keep bits of the forward, 0 beyond L
  float* sLse = reinterpret_cast<float*>(sM + (size_t)H * R * 8);   // [H][R] log2-domain log-sum-exp (+inf beyond L)
  float* sDelta = sLse + H * R;                                     // [H][R] rowsum(dO * O) per head
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t row_b = (size_t)b * L;
  constexpr bool use_bits = MODE == 1;
  const float qmul = a.scale * 1.4426950408889634f;
#define SAB_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[w * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  SAB_STAMP(0);
  if constexpr (MODE == 1) {
    for (int i = threadIdx.x; i < H * R * 2; i += NW * 64) {
      const int hr = i >> 1, h = hr / R, r = hr - h * R;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (r < L) v = reinterpret_cast<const uint4*>(a.mask + ((size_t)(b * H + h) * L + r) * 8)[i & 1];
      reinterpret_cast<uint4*>(sM + (size_t)hr * 8)[i & 1] = v;
    }
  }
  for (int i = threadIdx.x; i < H * R; i += NW * 64) {
    const int h = i / R, r = i - h * R;
    sLse[i] = r < L ? a.LSE[(size_t)(b * H + h) * L + r] * 1.4426950408889634f : INFINITY;
  }
  // images: 8 features per thread and step; delta from the same dO / O chunks (the chunks of a head are adjacent lanes)
  for (int i = threadIdx.x; i < R * 8; i += NW * 64) {
    const int r = i >> 3, c8 = (i & 7) * 8;
    float q[8], k[8], v[8], d[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = k[j] = v[j] = d[j] = o[j] = 0.f;
    if (r < L && a.in_bf16) {          // operands saved as bf16 rows by the lean forward; the gradient dO is always fp32
      const __bf16* Qb = reinterpret_cast<const __bf16*>(a.Q); const __bf16* Kb = reinterpret_cast<const __bf16*>(a.K);
      const __bf16* Vb = reinterpret_cast<const __bf16*>(a.V); const __bf16* Ob = reinterpret_cast<const __bf16*>(a.O);
      // saved rows are in the register order of a transposed tile (adt_tt.cuh: tt_store_bf16): features c8 .. c8+3 and c8+4 .. c8+7 of a
      // 64-feature row are two 8-byte pieces at 16 g + 4 nt and 16 (g + 1) + 4 nt, nt = c8 / 16, g = (c8 / 4) % 4
      const int pa = 16 * ((c8 >> 2) & 3) + 4 * (c8 >> 4), pb = pa + 16;
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      const bf16x4_t qa = *reinterpret_cast<const bf16x4_t*>(Qb + (row_b + r) * a.ldq + pa), qc = *reinterpret_cast<const bf16x4_t*>(Qb + (row_b + r) * a.ldq + pb);
      const bf16x4_t ka = *reinterpret_cast<const bf16x4_t*>(Kb + (row_b + r) * a.ldk + pa), kc = *reinterpret_cast<const bf16x4_t*>(Kb + (row_b + r) * a.ldk + pb);
      const bf16x4_t va = *reinterpret_cast<const bf16x4_t*>(Vb + (row_b + r) * a.ldv + pa), vc = *reinterpret_cast<const bf16x4_t*>(Vb + (row_b + r) * a.ldv + pb);
      const bf16x4_t oa = *reinterpret_cast<const bf16x4_t*>(Ob + (row_b + r) * a.ldo + pa), oc = *reinterpret_cast<const bf16x4_t*>(Ob + (row_b + r) * a.ldo + pb);
      bf16x8 qb, kb, vb, ob;
#pragma unroll
      for (int j = 0; j < 4; ++j) { qb[j] = qa[j]; qb[4 + j] = qc[j]; kb[j] = ka[j]; kb[4 + j] = kc[j]; vb[j] = va[j]; vb[4 + j] = vc[j]; ob[j] = oa[j]; ob[4 + j] = oc[j]; }
      *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(a.dO + (row_b + r) * a.lddo + c8);
      *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(a.dO + (row_b + r) * a.lddo + c8 + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) { q[j] = (float)qb[j]; k[j] = (float)kb[j]; v[j] = (float)vb[j]; o[j] = (float)ob[j]; }
    } else if (r < L) {
      *reinterpret_cast<float4*>(q) = *reinterpret_cast<const float4*>(a.Q + (row_b + r) * a.ldq + c8);
      *reinterpret_cast<float4*>(q + 4) = *reinterpret_cast<const float4*>(a.Q + (row_b + r) * a.ldq + c8 + 4);
      *reinterpret_cast<float4*>(k) = *reinterpret_cast<const float4*>(a.K + (row_b + r) * a.ldk + c8);
      *reinterpret_cast<float4*>(k + 4) = *reinterpret_cast<const float4*>(a.K + (row_b + r) * a.ldk + c8 + 4);
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(a.V + (row_b + r) * a.ldv + c8);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(a.V + (row_b + r) * a.ldv + c8 + 4);
      *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(a.dO + (row_b + r) * a.lddo + c8);
      *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(a.dO + (row_b + r) * a.lddo + c8 + 4);
      *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(a.O + (row_b + r) * a.ldo + c8);
      *reinterpret_cast<float4*>(o + 4) = *reinterpret_cast<const float4*>(a.O + (row_b + r) * a.ldo + c8 + 4);
    }
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { part += d[j] * o[j]; q[j] *= qmul; d[j] *= a.drop.scale; }      // the dO image carries 1 / (1 - p): see sab_pass_a
    *reinterpret_cast<bf16x8*>(sQ + r * TT_RS + c8) = pack8(q);
    *reinterpret_cast<bf16x8*>(sK + r * TT_RS + c8) = pack8(k);
    *reinterpret_cast<bf16x8*>(sV + r * TT_RS + c8) = pack8(v);
    *reinterpret_cast<bf16x8*>(sdO + r * TT_RS + c8) = pack8(d);
#pragma unroll
    for (int off = HD / 16; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);      // HD/8 adjacent lanes hold one head's chunks
    if (((i & 7) % (HD / 8)) == 0) sDelta[((i & 7) / (HD / 8)) * R + r] = part;
  }
  SAB_STAMP(1);
  __syncthreads();
  SAB_STAMP(2);
  const uint32_t key_rng = drop_key(a.drop);
  const int nqt = (L + 15) / 16;
  const float ln2 = 0.6931471805599453f;

  // ---- pass A: dQ (a wave owns query tile qt, keys on the accumulator rows) -----------------------------------------------------
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    const int tix = rnd * NW + sab_rank16(w);                      // heaviest causal tile first, balanced over the SIMDs
    if (tix >= nqt) continue;
    const int qt = nqt - 1 - tix;
    const int q = qt * 16 + c;
    const int nkt = qt + 1;
#pragma unroll 1
    for (int h = 0; h < H; ++h) {
      const float lse_q = sLse[h * R + q], delta_q = sDelta[h * R + q];
      bf16x8 fq[KB], fdo[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        fq[kb] = sab_rowfrag<HD>(sQ, q, h, kb, g);
        fdo[kb] = sab_rowfrag<HD>(sdO, q, h, kb, g);
      }
      f32x4 dq[NT];
      const uint32_t bh_rng = (uint32_t)(b * H + h) + a.bh_offset;
      const uint32_t idx_q = (bh_rng * (uint32_t)L + (uint32_t)q) * (uint32_t)L;
      sab_pass_a<HD, MODE>(sK, sV, fq, fdo, lse_q, delta_q, sM + ((size_t)h * R + q) * 8, qt, h, a.drop, key_rng, idx_q, c, g, dq);
      if (q < L && a.out_bf16) {          // saved-row order: feature 16 nt + 4 g + r of a 64-feature row at element 16 g + 4 nt + r
        __bf16* dst = reinterpret_cast<__bf16*>(a.dQ) + (row_b + q) * a.lddq + 16 * g + 4 * h * NT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf16x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (__bf16)(dq[nt][r] * a.scale);
          *reinterpret_cast<bf16x4*>(dst + 4 * nt) = v;
        }
      } else if (q < L) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          *reinterpret_cast<float4*>(a.dQ + (row_b + q) * a.lddq + h * HD + nt * 16 + 4 * g) =
              make_float4(dq[nt][0] * a.scale, dq[nt][1] * a.scale, dq[nt][2] * a.scale, dq[nt][3] * a.scale);
      }
    }
  }

  SAB_STAMP(3);
  // ---- pass B: dK, dV (a wave owns key tile kt, queries on the accumulator rows) ---------------------------------------------
  for (int rnd = 0; rnd * NW < nqt; ++rnd) {
    const int kt = rnd * NW + sab_rank16(w);                       // key tile 0 is the heaviest under the causal mask
    if (kt >= nqt) continue;
    const int key = kt * 16 + c;
#pragma unroll 1
    for (int h = 0; h < H; ++h) {
      bf16x8 fk[KB], fv[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        fk[kb] = sab_rowfrag<HD>(sK, key, h, kb, g);
        fv[kb] = sab_rowfrag<HD>(sV, key, h, kb, g);
      }
      f32x4 dk[NT], dv[NT];
      const uint32_t bh_rng = (uint32_t)(b * H + h) + a.bh_offset;
      sab_pass_b<HD, MODE>(sQ, sdO, fk, fv, sLse + h * R, sDelta + h * R, sM + (size_t)h * R * 8, kt, nqt, h, a.drop, key_rng, bh_rng * (uint32_t)L, L, c,
                           g, dk, dv);
      if (key < L && a.out_bf16) {
        __bf16* dstk = reinterpret_cast<__bf16*>(a.dK) + (row_b + key) * a.lddk + 16 * g + 4 * h * NT;
        __bf16* dstv = reinterpret_cast<__bf16*>(a.dV) + (row_b + key) * a.lddv + 16 * g + 4 * h * NT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf16x4 vk, vv;
#pragma unroll
          for (int r = 0; r < 4; ++r) { vk[r] = (__bf16)(dk[nt][r] * ln2); vv[r] = (__bf16)dv[nt][r]; }
          *reinterpret_cast<bf16x4*>(dstk + 4 * nt) = vk;
          *reinterpret_cast<bf16x4*>(dstv + 4 * nt) = vv;
        }
      } else if (key < L) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          // the Q image carries the factor log2(e) / sqrt(hd): dK = dS^T Q / sqrt(hd)
          *reinterpret_cast<float4*>(a.dK + (row_b + key) * a.lddk + h * HD + nt * 16 + 4 * g) =
              make_float4(dk[nt][0] * ln2, dk[nt][1] * ln2, dk[nt][2] * ln2, dk[nt][3] * ln2);
          *reinterpret_cast<float4*>(a.dV + (row_b + key) * a.lddv + h * HD + nt * 16 + 4 * g) =
              make_float4(dv[nt][0], dv[nt][1], dv[nt][2], dv[nt][3]);
        }
      }
    }
  }
  SAB_STAMP(4);
}

}  // namespace adt
