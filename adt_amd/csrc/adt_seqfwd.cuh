// Per-sequence fused layer forward (bf16 MFMA operands): ONE workgroup runs a whole EncoderLayer or DecoderLayer
// (sasrec/modules.py:644-655, :666-677) for ONE user sequence, 256 sequences = 256 CUs.
//
// The token-parallel parts are the wave-local row chains of adt_fwdchain.cuh (a wave owns a 16-token tile in registers through
// LayerNorm, the 64x64 products against LDS weight images, bias / dropout / ReLU / residual / mask and the head classifier); what
// the fusion removes is the trip through HBM around the attention core: a wave's q tile stays in its registers as MFMA operand
// fragments, k and v of the whole sequence go straight into the LDS images the attention sweeps (K row-major [L][64+8], V
// transposed [64][L+8], bf16), and the attention output of both heads -- an accumulator tile in exactly the row chains' C layout
// -- continues into out_proj without leaving the registers.  One workgroup barrier per attention (its keys and values come
// from every wave).  Arithmetic, operand rounding and dropout indices are those of the staged kernels (k_pre_fwd, k_attn_fwd_bf16,
// k_enc_post_fwd, k_dec_mid_fwd, k_dec_post_fwd), so both paths produce the same numbers.
//
// Causal work grows with the tile index: wave w owns tiles (n-1-w) and, for the first n-8 waves, w -- heaviest paired with lightest.
#pragma once
#include "adt_attn_bf16.cuh"
#include "adt_seq_args.h"
#include "adt_wave.cuh"

namespace adt {

constexpr int SQ_NW = 8;                       // waves per workgroup
constexpr int SQ_MAXKT = 14;                   // 16-row tiles per sequence: L <= 224
constexpr int SQ_LP = SQ_MAXKT * 16, SQ_LPT = SQ_LP + 8, SQ_RSK = 72;
constexpr int SQ_WIMG = 64 * WImg<PREC_BF16>::RS;

template <int NWT>
struct SeqFwdLds {
  static constexpr size_t wbytes = (size_t)NWT * SQ_WIMG * 2;
  static constexpr size_t kbytes = (size_t)SQ_LP * SQ_RSK * 2, vbytes = (size_t)64 * SQ_LPT * 2;
  static constexpr size_t bytes = wbytes + kbytes + vbytes + (size_t)SQ_NW * WV_SCR * sizeof(float);
  __bf16* w[NWT]; __bf16* sK; __bf16* sVT; float* scr;
  __device__ SeqFwdLds(unsigned char* base, int wave) {
    __bf16* pw = reinterpret_cast<__bf16*>(base);
    for (int i = 0; i < NWT; ++i) w[i] = pw + i * SQ_WIMG;
    sK = reinterpret_cast<__bf16*>(base + wbytes);
    sVT = reinterpret_cast<__bf16*>(base + wbytes + kbytes);
    scr = reinterpret_cast<float*>(base + wbytes + kbytes + vbytes) + wave * WV_SCR;
  }
};

// tile of slot s (0, 1) owned by wave w; -1 = none
ADT_DEVICE_INLINE int seq_tile(int s, int w, int ntiles) {
  if (s == 0) return ntiles - 1 - w;                 // may be negative for very short sequences
  return w < ntiles - SQ_NW ? w : -1;
}

// q fragments of one tile (scaled by 1/sqrt(hd)) from the wave scratch holding q (16 x 64 fp32 rows): fq[h*KB + kb]
template <int HD>
ADT_DEVICE_INLINE void seq_qfrags(const float* scr, float scale, int c, int g, bf16x8 (&fq)[(64 / HD) * ((HD + 31) / 32)]) {
  constexpr int H = 64 / HD, KB = (HD + 31) / 32;
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (kb * 32 + 8 * g < HD) {
        const float* p = scr + c * WV_RS + h * HD + kb * 32 + 8 * g;
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(p);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= scale;
      }
      fq[h * KB + kb] = pack8(v);
    }
}

// rows of the wave scratch (a 16 x 64 fp32 tile) -> rows tile*16.. of the K image; rows >= L become zero
ADT_DEVICE_INLINE void seq_put_k(__bf16* sK, const float* scr, int tile, int L, int lane) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = q * 64 + lane, r = i >> 3, c8 = (i & 7) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tile * 16 + r < L) {
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(scr + r * WV_RS + c8);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(scr + r * WV_RS + c8 + 4);
    }
    *reinterpret_cast<bf16x8*>(sK + (tile * 16 + r) * SQ_RSK + c8) = pack8(v);
  }
}

// a C-layout tile -> columns tile*16.. of the transposed V image (4 consecutive tokens of one feature per store)
ADT_DEVICE_INLINE void seq_put_vt(__bf16* sVT, const CT& v, int tile, int L, int c, int g) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    bf16x4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = (__bf16)((tile * 16 + 4 * g + r < L) ? v.v[nt][r] : 0.f);
    *reinterpret_cast<bf16x4*>(sVT + (16 * nt + c) * SQ_LPT + tile * 16 + 4 * g) = b;
  }
}

template <int HD>
ADT_DEVICE_INLINE bf16x8 seq_kfrag(const __bf16* sK, int row, int h, int kb, int g) {
  if (kb * 32 + 8 * g < HD) return *reinterpret_cast<const bf16x8*>(sK + row * SQ_RSK + h * HD + kb * 32 + 8 * g);
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
  return z;
}

// Causal softmax(q k^T) v for query tile qt of head h: the loop of k_attn_fwd_bf16 on this sequence's LDS images.  Writes the
// log-sum-exp and the dropout keep bits of the tile's queries; returns the normalised output tile (rows 4g+r, columns 16nt+c).
template <int HD>
ADT_DEVICE_INLINE void seq_attn_tile(const __bf16* sK, const __bf16* sVT, const bf16x8* fq, int qt, int L, int h, int bh,
                                     uint32_t bh_rng, const DropCfg& drop, uint32_t key_rng, float* lse, uint32_t* mask, int lane,
                                     int c, int g, f32x4 (&o)[HD / 16]) {
  constexpr int NT = HD / 16, KB = (HD + 31) / 32, MAXKT = SQ_MAXKT;
  const int q = qt * 16 + c;
  const int nkt = qt + 1;
  auto score = [&](int kt, f32x4& sc) {
    sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) sc = mfma_bf16(sc, seq_kfrag<HD>(sK, kt * 16 + c, h, kb, g), fq[kb]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      sc[r] = (key < L && key <= q) ? sc[r] : -INFINITY;
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
  const uint32_t idx_q = (bh_rng * (uint32_t)L + (uint32_t)q) * (uint32_t)L;
  float sum = 0.f;
  uint32_t mw[MAXKT / 2];
#pragma unroll
  for (int i = 0; i < MAXKT / 2; ++i) mw[i] = 0u;
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
          const float e = __expf(sc[r] - m);
          sum += e;
          float p = e;
          if (drop.thr) {
            const uint32_t key = kt * 16 + 4 * g + r;
            const bool keep = kt < nkt && adt_keep(key_rng, idx_q + key, drop.thr);
            p = keep ? e * drop.scale : 0.f;
            mw[kp] |= (keep ? 1u : 0u) << (16 * t + 4 * g + r);
          }
          pv[4 * t + r] = p;
        }
      }
      const bf16x8 fp = pack8(pv);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] = mfma_bf16(o[nt], fp, tfrag(sVT, SQ_LPT, h * HD + nt * 16 + c, kp * 32, g));
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  if (g == 0 && q < L) lse[(size_t)bh * L + q] = m + __logf(sum);
  float inv_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) inv_r[r] = 1.0f / __shfl(sum, (lane & 48) | (4 * g + r), 64);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[nt][r] *= inv_r[r];
  if (mask && drop.thr) {
    uint32_t ow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t v = i < MAXKT / 2 ? mw[i] : 0u;
      v |= (uint32_t)__shfl_xor((int)v, 16, 64);
      v |= (uint32_t)__shfl_xor((int)v, 32, 64);
      ow[i] = v;
    }
    if (g == 0 && q < L) {
      uint4* dst = reinterpret_cast<uint4*>(mask + ((size_t)bh * L + q) * 8);
      dst[0] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      dst[1] = make_uint4(ow[4], ow[5], ow[6], ow[7]);
    }
  }
}

// attention of every head for one query tile -> a 16 x 64 tile in the row chains' C layout
template <int HD>
ADT_DEVICE_INLINE CT seq_attn_heads(const __bf16* sK, const __bf16* sVT, const bf16x8* fq, int tile, int L, int b, uint32_t b_offset,
                                    DropCfg drop, uint32_t site, uint32_t seedv, float* lse, uint32_t* mask, int lane, int c, int g, int ablate = 0) {
  constexpr int H = 64 / HD, NT = HD / 16, KB = (HD + 31) / 32;
  drop.site = site;
  const uint32_t key_rng = drop.thr ? adt_site_key(seedv, site) : 0u;
  CT o;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    f32x4 oh[NT];
    const int bh = b * H + h;
    if (ablate & 2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o.v[h * NT + nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    seq_attn_tile<HD>(sK, sVT, fq + h * KB, tile, L, h, bh, (uint32_t)bh + b_offset * (uint32_t)H, drop, key_rng, lse, mask, lane, c, g, oh);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o.v[h * NT + nt] = oh[nt];
  }
  return o;
}

ADT_DEVICE_INLINE void seq_zero_images(__bf16* sK, int nbytes) {
  // keys / values beyond the sequence must read as zeros (an odd tile count leaves a half-used pair in the P.V sweep)
  uint4* p = reinterpret_cast<uint4*>(sK);
  for (int i = threadIdx.x; i < nbytes / 16; i += SQ_NW * 64) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// x tile -> wave scratch: a load, or the embedding gather x = dropout(E[id] * sqrt(d) + P[l]) * (id != 0)   (sasrec/model.py:34-41)
ADT_DEVICE_INLINE void seq_load_x(const SeqFwdArgs& a, float* scr, int row0, int Tend, uint32_t key0, int lane) {
  wave_fence();
  if (a.x) {
    rows_to_scr(scr, a.x, 64, row0, Tend, lane);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = q * 64 + lane, r = i >> 4, c4 = (i & 15) * 4, row = row0 + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int id = row < Tend ? a.ids[row] : 0;
      if (id != 0) {
        const float4 e = *reinterpret_cast<const float4*>(a.E + (size_t)id * 64 + c4);
        const float4 p = *reinterpret_cast<const float4*>(a.P + (size_t)(row % a.L) * 64 + c4);
        v[0] = e.x * a.emb_scale + p.x; v[1] = e.y * a.emb_scale + p.y; v[2] = e.z * a.emb_scale + p.z; v[3] = e.w * a.emb_scale + p.w;
        if (a.drop.thr) {
          const uint32_t base = (uint32_t)(row + a.b_offset * (uint32_t)a.L) * 64u + (uint32_t)c4;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = adt_keep(key0, base + j, a.drop.thr) ? v[j] * a.drop.scale : 0.f;
        }
      }
      *reinterpret_cast<float4*>(scr + r * WV_RS + c4) = *reinterpret_cast<float4*>(v);
      if (row < Tend && a.x_out) *reinterpret_cast<float4*>(a.x_out + (size_t)row * 64 + c4) = *reinterpret_cast<float4*>(v);
    }
  }
  wave_fence();
}

// LayerNorm + packed in-projection of one tile: q fragments to registers, k / v into the LDS images (and everything the
// backward reads to HBM).  ENC: q from LN(x), k / v from the raw x (sasrec/modules.py:646-647); decoder: all from LN(x).
template <int HD, bool ENC>
ADT_DEVICE_INLINE void seq_pre_tile(const SeqFwdArgs& a, const __bf16* wq, const __bf16* wk, const __bf16* wv, __bf16* sK, __bf16* sVT,
                                    float* scr, int tile, int row0, int Tend, uint32_t key0, int lane, int c, int g,
                                    bf16x8 (&fq)[(64 / HD) * ((HD + 31) / 32)]) {
  seq_load_x(a, scr, row0, Tend, key0, lane);
  const CT x = scr_to_ct(scr, c, g);
  AFrags<PREC_BF16> ax;
  if (ENC) ax = scr_to_a<PREC_BF16>(scr, c, g);
  LnStat st;
  const CT xn = ln_apply(ln_xhat(x, a.ln_eps, st), a.gamma, a.beta, c);
  wave_fence();
  ct_to_scr(scr, xn, c, g);
  wave_fence();
  scr_to_rows(a.xn, 64, scr, row0, Tend, lane);
  const AFrags<PREC_BF16> an = scr_to_a<PREC_BF16>(scr, c, g);
  {
    CT q = gemm_w<PREC_BF16>(an, wq, c, g);
    ct_add_bias(q, a.bin, c);
    wave_fence();
    ct_to_scr(scr, q, c, g);
    wave_fence();
    if (a.qkv) scr_to_rows(a.qkv, 192, scr, row0, Tend, lane);
    seq_qfrags<HD>(scr, a.scale, c, g, fq);
  }
  {
    CT k = gemm_w<PREC_BF16>(ENC ? ax : an, wk, c, g);
    ct_add_bias(k, a.bin + 64, c);
    wave_fence();
    ct_to_scr(scr, k, c, g);
    wave_fence();
    if (a.qkv) scr_to_rows(a.qkv + 64, 192, scr, row0, Tend, lane);
    seq_put_k(sK, scr, tile, a.L, lane);
  }
  {
    CT v = gemm_w<PREC_BF16>(ENC ? ax : an, wv, c, g);
    ct_add_bias(v, a.bin + 128, c);
    seq_put_vt(sVT, v, tile, a.L, c, g);
    if (a.qkv) store_ct(scr, a.qkv + 128, 192, v, row0, Tend, lane, c, g);
  }
}

// u = relu(dropout1(xin W1^T + b1)) -> stored ; returns dropout2(u W2^T + b2)      (PointWiseFeedForward, sasrec/modules.py:629-633)
ADT_DEVICE_INLINE CT seq_ffn_tile(const SeqFwdArgs& a, const __bf16* w1, const __bf16* w2, float* scr, const AFrags<PREC_BF16>& ain,
                                  uint32_t key1, uint32_t key2, int row0, int Tend, int lane, int c, int g) {
  const uint32_t rbase = (uint32_t)row0 + a.b_offset * (uint32_t)a.L;
  if (a.ablate & 4) { CT z; for (int nt = 0; nt < 4; ++nt) z.v[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; return z; }
  CT u = gemm_w<PREC_BF16>(ain, w1, c, g);
  ct_add_bias(u, a.b1, c);
  ct_dropmask(u, key1, a.drop, rbase, c, g);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) u.v[nt][r] = fmaxf(u.v[nt][r], 0.f);
  wave_fence();
  ct_to_scr(scr, u, c, g);
  wave_fence();
  if (a.u) scr_to_rows(a.u, 64, scr, row0, Tend, lane);
  CT y = gemm_w<PREC_BF16>(scr_to_a<PREC_BF16>(scr, c, g), w2, c, g);
  ct_add_bias(y, a.b2, c);
  ct_dropmask(y, key2, a.drop, rbase, c, g);
  return y;
}

#define SQ_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[w * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

// ---- encoder layer -----------------------------------------------------------------------------------------------------------
// weight images: 0 Wq, 1 Wk, 2 Wv, 3 out_proj, 4 conv1, 5 conv2
template <int HD, int HC>     // HC: compile-time cap on the classifier width (number of heads)
__global__ __launch_bounds__(SQ_NW * 64) void k_seq_enc_fwd(SeqFwdArgs a) {
  constexpr int H = 64 / HD, KB = (HD + 31) / 32, NF = H * KB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  SeqFwdLds<6> lds(smem_raw, w);
  const int b = blockIdx.x, L = a.L, ntiles = (L + 15) / 16;
  const int Tend = b * L + L;
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const WPack wpk{a.wp_base, reinterpret_cast<const __bf16*>(a.wp_img)};
  SQ_STAMP(0);
  {
    __bf16* const im[6] = {lds.w[0], lds.w[1], lds.w[2], lds.w[3], lds.w[4], lds.w[5]};
    const float* const wsrc[6] = {a.Win, a.Win + 4096, a.Win + 8192, a.Wo, a.W1, a.W2};
    stage_w_set<PREC_BF16, SQ_NW * 64, 6>(im, wsrc, false, wpk);
  }
  seq_zero_images(lds.sK, (int)(SeqFwdLds<6>::kbytes + SeqFwdLds<6>::vbytes));
  __syncthreads();
  SQ_STAMP(1);
  const uint32_t key0 = adt_site_key(seedv, a.site_emb), key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  bf16x8 fq[2][NF];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = seq_tile(s, w, ntiles);
    if (tile >= 0) seq_pre_tile<HD, true>(a, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sVT, lds.scr, tile, b * L + tile * 16, Tend, key0, lane, c, g, fq[s]);
    SQ_STAMP(2 + s);
  }
  __syncthreads();
  SQ_STAMP(4);
  // head classifier constants (k_enc_post_fwd): column 16nt+c of this lane is element j of head hcol[nt]
  constexpr int HCM = HC > 0 ? HC : 1;
  int hcol[4];
  float wcls[4][HCM];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    hcol[nt] = (16 * nt + c) / HD;
    const int j = 16 * nt + c - hcol[nt] * HD;
#pragma unroll
    for (int cc = 0; cc < HCM; ++cc) wcls[nt][cc] = (a.rec && cc < H) ? a.Ws[cc * HD + j] : 0.f;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = seq_tile(s, w, ntiles);
    if (tile < 0) continue;
    const int row0 = b * L + tile * 16;
    const RowRegs qn_rows = rows_load(a.xn, 64, row0, Tend, lane);      // the residual adds LN1(x): requested now, used after the attention
    const CT o = seq_attn_heads<HD>(lds.sK, lds.sVT, fq[s], tile, L, b, a.b_offset, a.drop, a.site_attn, seedv, a.lse, a.mask, lane, c, g, a.ablate);
    SQ_STAMP(5 + 3 * s);
    wave_fence();
    ct_to_scr(lds.scr, o, c, g);
    wave_fence();
    if (a.o) scr_to_rows(a.o, 64, lds.scr, row0, Tend, lane);
    const AFrags<PREC_BF16> ao = scr_to_a<PREC_BF16>(lds.scr, c, g);
    if (HC > 0 && a.rec) {
      // z[h][cc] = sum_j o[h*hd + j] Ws[cc][j] + bs[cc]; log-softmax over cc        (sasrec/modules.py:648-649)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
#pragma unroll 1
        for (int h = 0; h < H; ++h) {
          float z[HCM];
#pragma unroll
          for (int cc = 0; cc < HCM; ++cc) {
            z[cc] = -INFINITY;
            if (cc < H) {
              float sacc = 0.f;
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) sacc += (hcol[nt] == h) ? o.v[nt][r] * wcls[nt][cc] : 0.f;
              z[cc] = row_sum16(sacc) + a.bs[cc];
            }
          }
          float m = z[0];
#pragma unroll
          for (int cc = 1; cc < HCM; ++cc) m = fmaxf(m, z[cc]);
          float se = 0.f;
#pragma unroll
          for (int cc = 0; cc < HCM; ++cc) se += (cc < H) ? __expf(z[cc] - m) : 0.f;
          const float lz = m + __logf(se);
          if (c == 0 && row < Tend) {
            const int l = row - b * L;
            float* dst = a.rec + ((size_t)(l * a.B + b) * H + h) * H;
#pragma unroll
            for (int cc = 0; cc < HCM; ++cc)
              if (cc < H) dst[cc] = z[cc] - lz;
          }
        }
      }
    }
    CT hh = gemm_w<PREC_BF16>(ao, lds.w[3], c, g);
    ct_add_bias(hh, a.bo, c);
    ct_add(hh, rows_to_ct(lds.scr, qn_rows, lane, c, g));
    SQ_STAMP(6 + 3 * s);
    if (a.h) store_ct(lds.scr, a.h, 64, hh, row0, Tend, lane, c, g);
    LnStat st;
    const CT h2 = ln_apply(ln_xhat(hh, a.ln_eps, st), a.gamma2, a.beta2, c);
    CT y = seq_ffn_tile(a, lds.w[4], lds.w[5], lds.scr, ct_to_a<PREC_BF16>(lds.scr, h2, c, g), key1, key2, row0, Tend, lane, c, g);
    ct_add(y, h2);
    ct_mask_rows(y, a.ids, row0, Tend, g);
    store_ct(lds.scr, a.y, 64, y, row0, Tend, lane, c, g);
    SQ_STAMP(7 + 3 * s);
  }
}

// ---- decoder layer -----------------------------------------------------------------------------------------------------------
// weight set A: 0 Wq, 1 Wk, 2 Wv (slf_attn), 3 slf out_proj, 4 enc_attn Wq ; set B: 0 enc_attn Wk, 1 Wv, 2 enc_attn out_proj, 3 conv1, 4 conv2
template <int HD>
__global__ __launch_bounds__(SQ_NW * 64) void k_seq_dec_fwd(SeqFwdArgs a) {
  constexpr int H = 64 / HD, KB = (HD + 31) / 32, NF = H * KB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  SeqFwdLds<5> lds(smem_raw, w);
  const int b = blockIdx.x, L = a.L, ntiles = (L + 15) / 16;
  const int Tend = b * L + L;
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const WPack wpk{a.wp_base, reinterpret_cast<const __bf16*>(a.wp_img)};
  {
    __bf16* const im[5] = {lds.w[0], lds.w[1], lds.w[2], lds.w[3], lds.w[4]};
    const float* const wsrc[5] = {a.Win, a.Win + 4096, a.Win + 8192, a.Wo, a.Win2};
    stage_w_set<PREC_BF16, SQ_NW * 64, 5>(im, wsrc, false, wpk);
  }
  seq_zero_images(lds.sK, (int)(SeqFwdLds<5>::kbytes + SeqFwdLds<5>::vbytes));
  __syncthreads();
  const uint32_t key0 = adt_site_key(seedv, a.site_emb), key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  bf16x8 fq[2][NF];
  // self attention: D = LN(x); q, k, v = D Win^T + b                                                 (sasrec/modules.py:668-670)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = seq_tile(s, w, ntiles);
    if (tile >= 0) seq_pre_tile<HD, false>(a, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sVT, lds.scr, tile, b * L + tile * 16, Tend, key0, lane, c, g, fq[s]);
  }
  __syncthreads();
  // a1 = out_proj(o1) ; q2 = a1 Wq2^T + b : the cross attention's queries replace the self attention's in the registers
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = seq_tile(s, w, ntiles);
    if (tile < 0) continue;
    const int row0 = b * L + tile * 16;
    const CT o1 = seq_attn_heads<HD>(lds.sK, lds.sVT, fq[s], tile, L, b, a.b_offset, a.drop, a.site_attn, seedv, a.lse, a.mask, lane, c, g, a.ablate);
    wave_fence();
    ct_to_scr(lds.scr, o1, c, g);
    wave_fence();
    if (a.o) scr_to_rows(a.o, 64, lds.scr, row0, Tend, lane);
    CT a1 = gemm_w<PREC_BF16>(scr_to_a<PREC_BF16>(lds.scr, c, g), lds.w[3], c, g);
    ct_add_bias(a1, a.bo, c);
    wave_fence();
    ct_to_scr(lds.scr, a1, c, g);
    wave_fence();
    if (a.a1) scr_to_rows(a.a1, 64, lds.scr, row0, Tend, lane);
    CT q2 = gemm_w<PREC_BF16>(scr_to_a<PREC_BF16>(lds.scr, c, g), lds.w[4], c, g);
    ct_add_bias(q2, a.bin2, c);
    wave_fence();
    ct_to_scr(lds.scr, q2, c, g);
    wave_fence();
    if (a.q2) scr_to_rows(a.q2, 64, lds.scr, row0, Tend, lane);
    seq_qfrags<HD>(lds.scr, a.scale, c, g, fq[s]);
  }
  __syncthreads();                      // every wave is done with the self-attention images and with weight set A
  {
    __bf16* const im[5] = {lds.w[0], lds.w[1], lds.w[2], lds.w[3], lds.w[4]};
    const float* const wsrc[5] = {a.Win2 + 4096, a.Win2 + 8192, a.Wo2, a.W1, a.W2};
    stage_w_set<PREC_BF16, SQ_NW * 64, 5>(im, wsrc, false, wpk);
  }
  __syncthreads();
  // cross attention keys / values from the encoder's log_feats: [k2, v2] = f Wkv^T + b            (memory = log_feats, model.py:69-70)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = seq_tile(s, w, ntiles);
    if (tile < 0) continue;
    const int row0 = b * L + tile * 16;
    wave_fence();
    rows_to_scr(lds.scr, a.f, 64, row0, Tend, lane);
    wave_fence();
    const AFrags<PREC_BF16> af = scr_to_a<PREC_BF16>(lds.scr, c, g);
    CT k2 = gemm_w<PREC_BF16>(af, lds.w[0], c, g);
    ct_add_bias(k2, a.bin2 + 64, c);
    wave_fence();
    ct_to_scr(lds.scr, k2, c, g);
    wave_fence();
    if (a.kv2) scr_to_rows(a.kv2, 128, lds.scr, row0, Tend, lane);
    seq_put_k(lds.sK, lds.scr, tile, L, lane);
    CT v2 = gemm_w<PREC_BF16>(af, lds.w[1], c, g);
    ct_add_bias(v2, a.bin2 + 128, c);
    seq_put_vt(lds.sVT, v2, tile, L, c, g);
    if (a.kv2) store_ct(lds.scr, a.kv2 + 64, 128, v2, row0, Tend, lane, c, g);
  }
  __syncthreads();
  // a2 = out_proj(o2) ; y = (D + a2 + FFN(a2)) * mask                                               (sasrec/modules.py:673-676)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = seq_tile(s, w, ntiles);
    if (tile < 0) continue;
    const int row0 = b * L + tile * 16;
    const RowRegs dn_rows = rows_load(a.xn, 64, row0, Tend, lane);
    const CT o2 = seq_attn_heads<HD>(lds.sK, lds.sVT, fq[s], tile, L, b, a.b_offset, a.drop, a.site_attn2, seedv, a.lse2, a.mask2, lane, c, g, a.ablate);
    wave_fence();
    ct_to_scr(lds.scr, o2, c, g);
    wave_fence();
    if (a.o2) scr_to_rows(a.o2, 64, lds.scr, row0, Tend, lane);
    CT a2 = gemm_w<PREC_BF16>(scr_to_a<PREC_BF16>(lds.scr, c, g), lds.w[2], c, g);
    ct_add_bias(a2, a.bo2, c);
    wave_fence();
    ct_to_scr(lds.scr, a2, c, g);
    wave_fence();
    if (a.h) scr_to_rows(a.h, 64, lds.scr, row0, Tend, lane);
    CT y = seq_ffn_tile(a, lds.w[3], lds.w[4], lds.scr, scr_to_a<PREC_BF16>(lds.scr, c, g), key1, key2, row0, Tend, lane, c, g);
    ct_add(y, a2);
    ct_add(y, rows_to_ct(lds.scr, dn_rows, lane, c, g));
    ct_mask_rows(y, a.ids, row0, Tend, g);
    store_ct(lds.scr, a.y, 64, y, row0, Tend, lane, c, g);
  }
}

}  // namespace adt
