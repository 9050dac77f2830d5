// Wave-local forward row chains (see adt_wave.cuh): everything between two attention cores of EncoderLayer /
// DecoderLayer (sasrec/modules.py:644-655, :666-677) plus the embedding gather and the last LayerNorm + logits
// (sasrec/model.py:34-41,48,72-76).  A wave owns 16 tokens; weights are staged once per workgroup as (bf16) LDS
// images; there is no workgroup barrier after the staging.
#pragma once
#include "adt_fwdchain_args.h"
#include "adt_misc.cuh"
#include "adt_wave.cuh"

namespace adt {

template <int PREC, int NW, int NWT>
struct FwdLds {
  typedef typename WImg<PREC>::T WT;
  static constexpr int WIMG = 64 * WImg<PREC>::RS;
  static constexpr size_t bytes = NWT * WIMG * sizeof(WT) + (size_t)NW * WV_SCR * sizeof(float);
  WT* w[4]; float* scr;
  __device__ FwdLds(unsigned char* base, int wave) {
    WT* pw = reinterpret_cast<WT*>(base);
    for (int i = 0; i < NWT; ++i) w[i] = pw + i * WIMG;
    scr = reinterpret_cast<float*>(base + NWT * WIMG * sizeof(WT)) + wave * WV_SCR;
  }
};

#define FWD_PROLOGUE(NWT_)                                                                          \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];                          \
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;            \
  FwdLds<PREC, NW, NWT_> lds(smem_raw, w);                                                          \
  const int ntiles = (a.T + 15) / 16;                                                               \
  const int tstride = gridDim.x * NW;                                                               \
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;

// ---- pre: [gather] -> LN -> q, k, v ---------------------------------------------------------------------
// ENC: q from LN(x), k/v from raw x (sasrec/modules.py:646-647); DEC: all three from LN(x) (:668-670)
template <int PREC, int NW, bool ENC>
__global__ __launch_bounds__(NW * 64) void k_pre_fwd(FwdChainArgs a) {
  FWD_PROLOGUE(3)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W[0], false);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W[1], false);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W[2], false);
  __syncthreads();
  const uint32_t key0 = adt_site_key(seedv, a.site0);
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    wave_fence();
    if (a.x) {
      rows_to_scr(lds.scr, a.x, 64, row0, a.T, lane);
    } else {
      // x = dropout(E[id] * sqrt(d) + P[l]) * (id != 0)            (sasrec/model.py:34-41)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = q * 64 + lane, r = i >> 4, c4 = (i & 15) * 4, row = row0 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int id = row < a.T ? a.ids[row] : 0;
        if (id != 0) {
          const float4 e = *reinterpret_cast<const float4*>(a.E + (size_t)id * 64 + c4);
          const float4 p = *reinterpret_cast<const float4*>(a.P + (size_t)(row % a.L) * 64 + c4);
          v[0] = e.x * a.emb_scale + p.x; v[1] = e.y * a.emb_scale + p.y; v[2] = e.z * a.emb_scale + p.z; v[3] = e.w * a.emb_scale + p.w;
          if (a.drop.thr) {
            const uint32_t base = (uint32_t)(row + a.row_offset) * 64u + (uint32_t)c4;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = adt_keep(key0, base + j, a.drop.thr) ? v[j] * a.drop.scale : 0.f;
          }
        }
        *reinterpret_cast<float4*>(lds.scr + r * WV_RS + c4) = *reinterpret_cast<float4*>(v);
        if (row < a.T) *reinterpret_cast<float4*>(a.o0 + (size_t)row * a.ld0 + c4) = *reinterpret_cast<float4*>(v);
      }
    }
    wave_fence();
    const CT x = scr_to_ct(lds.scr, c, g);
    AFrags<PREC> ax;
    if (ENC) ax = scr_to_a<PREC>(lds.scr, c, g);
    LnStat st;
    const CT xn = ln_apply(ln_xhat(x, a.ln_eps, st), a.gamma, a.beta, c);
    wave_fence();
    ct_to_scr(lds.scr, xn, c, g);
    wave_fence();
    scr_to_rows(a.o1, a.ld1, lds.scr, row0, a.T, lane);
    const AFrags<PREC> an = scr_to_a<PREC>(lds.scr, c, g);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      CT y = gemm_w<PREC>((ENC && j > 0) ? ax : an, lds.w[j], c, g);
      ct_add_bias(y, a.b[j], c);
      store_ct(lds.scr, a.o2 + 64 * j, a.ld2, y, row0, a.T, lane, c, g);
    }
  }
}

// ---- encoder post: h = qn + o Wo^T + bo ; h2 = LN2(h) ; u = relu(drop1(h2 W1^T + b1)) ; y = (h2 + drop2(u W2^T + b2)) * mask
// W0 = out_proj, W1 = conv1, W2 = conv2 ; o0 = h, o1 = u, o2 = y ; optional head classifier on o -> rec
template <int PREC, int NW, int HC>   // HC: compile-time cap on the number of heads of the fused classifier (0 = none)
// Up to two classifier heads the kernel fits 128 registers (4 waves per SIMD = two workgroups per CU: 32.6 -> 25.7 us); it needed 134
// without the hint.  Wider classifiers keep the two-waves budget (at 128 they would spill).
__global__ __launch_bounds__(NW * 64, (HC <= 2 ? 4 : 2)) void k_enc_post_fwd(FwdChainArgs a) {
  constexpr int HCM = HC > 0 ? HC : 1;
  FWD_PROLOGUE(3)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W[0], false);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W[1], false);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W[2], false);
  __syncthreads();
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  const int hd = 64 / a.H;
  int hcol[4];
  float wcls[4][HCM];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    hcol[nt] = (16 * nt + c) / hd;
    const int j = 16 * nt + c - hcol[nt] * hd;
#pragma unroll
    for (int cc = 0; cc < HCM; ++cc) wcls[nt][cc] = (a.rec && cc < a.H) ? a.Ws[cc * hd + j] : 0.f;
  }
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    const RowRegs qn_rows = rows_load(a.r0, 64, row0, a.T, lane);
    wave_fence();
    rows_to_scr(lds.scr, a.x, 64, row0, a.T, lane);
    wave_fence();
    const AFrags<PREC> ao = scr_to_a<PREC>(lds.scr, c, g);
    if (HC > 0 && a.rec) {
      // head classifier (sasrec/modules.py:648-649): z[h][cc] = sum_j o[h*hd + j] Ws[cc][j] + bs[cc]; log-softmax over cc.
      // Column 16nt+c of this lane belongs to head hcol[nt]; one 16-lane row reduction per (row, head, class).
      const CT o = scr_to_ct(lds.scr, c, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
#pragma unroll 1
        for (int h = 0; h < a.H; ++h) {
          float z[HCM];
#pragma unroll
          for (int cc = 0; cc < HCM; ++cc) {
            z[cc] = -INFINITY;
            if (cc < a.H) {
              float s = 0.f;
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) s += (hcol[nt] == h) ? o.v[nt][r] * wcls[nt][cc] : 0.f;
              z[cc] = row_sum16(s) + a.bs[cc];
            }
          }
          float m = z[0];
#pragma unroll
          for (int cc = 1; cc < HCM; ++cc) m = fmaxf(m, z[cc]);
          float se = 0.f;
#pragma unroll
          for (int cc = 0; cc < HCM; ++cc) se += (cc < a.H) ? __expf(z[cc] - m) : 0.f;
          const float lz = m + __logf(se);
          if (c == 0 && row < a.T) {
            const int b = row / a.L, l = row - b * a.L;
            float* dst = a.rec + ((size_t)(l * a.B + b) * a.H + h) * a.H;
#pragma unroll
            for (int cc = 0; cc < HCM; ++cc)
              if (cc < a.H) dst[cc] = z[cc] - lz;
          }
        }
      }
    }
    CT h = gemm_w<PREC>(ao, lds.w[0], c, g);
    ct_add_bias(h, a.b[0], c);
    ct_add(h, rows_to_ct(lds.scr, qn_rows, lane, c, g));
    store_ct(lds.scr, a.o0, a.ld0, h, row0, a.T, lane, c, g);
    LnStat st;
    const CT h2 = ln_apply(ln_xhat(h, a.ln_eps, st), a.gamma, a.beta, c);
    CT u = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, h2, c, g), lds.w[1], c, g);
    ct_add_bias(u, a.b[1], c);
    ct_dropmask(u, key1, a.drop, (uint32_t)row0 + a.row_offset, c, g);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) u.v[nt][r] = fmaxf(u.v[nt][r], 0.f);
    wave_fence();
    ct_to_scr(lds.scr, u, c, g);
    wave_fence();
    scr_to_rows(a.o1, a.ld1, lds.scr, row0, a.T, lane);
    CT y = gemm_w<PREC>(scr_to_a<PREC>(lds.scr, c, g), lds.w[2], c, g);
    ct_add_bias(y, a.b[2], c);
    ct_dropmask(y, key2, a.drop, (uint32_t)row0 + a.row_offset, c, g);
    ct_add(y, h2);
    ct_mask_rows(y, a.ids, row0, a.T, g);
    store_ct(lds.scr, a.o2, a.ld2, y, row0, a.T, lane, c, g);
  }
}

// ---- decoder mid: a1 = o1 Wo1^T + b ; q2 = a1 Wq2^T + b -----------------------------------------------------
// W0 = slf out_proj, W1 = enc_attn Wq ; o0 = a1, o2 = q2
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_mid_fwd(FwdChainArgs a) {
  FWD_PROLOGUE(2)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W[0], false);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W[1], false);
  __syncthreads();
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    wave_fence();
    rows_to_scr(lds.scr, a.x, 64, row0, a.T, lane);
    wave_fence();
    CT a1 = gemm_w<PREC>(scr_to_a<PREC>(lds.scr, c, g), lds.w[0], c, g);
    ct_add_bias(a1, a.b[0], c);
    wave_fence();
    ct_to_scr(lds.scr, a1, c, g);
    wave_fence();
    scr_to_rows(a.o0, a.ld0, lds.scr, row0, a.T, lane);
    CT q2 = gemm_w<PREC>(scr_to_a<PREC>(lds.scr, c, g), lds.w[1], c, g);
    ct_add_bias(q2, a.b[1], c);
    store_ct(lds.scr, a.o2, a.ld2, q2, row0, a.T, lane, c, g);
  }
}

// ---- decoder post: a2 = o2 Wo2^T + b ; u = relu(drop1(a2 W1^T + b1)) ; y = (dn + a2 + drop2(u W2^T + b2)) * mask
// W0 = enc_attn out_proj, W1 = conv1, W2 = conv2 ; r0 = dn ; o0 = a2, o1 = u, o2 = y
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_post_fwd(FwdChainArgs a) {
  FWD_PROLOGUE(3)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W[0], false);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W[1], false);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W[2], false);
  __syncthreads();
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    const RowRegs dn_rows = rows_load(a.r0, 64, row0, a.T, lane);
    wave_fence();
    rows_to_scr(lds.scr, a.x, 64, row0, a.T, lane);
    wave_fence();
    CT a2 = gemm_w<PREC>(scr_to_a<PREC>(lds.scr, c, g), lds.w[0], c, g);
    ct_add_bias(a2, a.b[0], c);
    wave_fence();
    ct_to_scr(lds.scr, a2, c, g);
    wave_fence();
    scr_to_rows(a.o0, a.ld0, lds.scr, row0, a.T, lane);
    CT u = gemm_w<PREC>(scr_to_a<PREC>(lds.scr, c, g), lds.w[1], c, g);
    ct_add_bias(u, a.b[1], c);
    ct_dropmask(u, key1, a.drop, (uint32_t)row0 + a.row_offset, c, g);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) u.v[nt][r] = fmaxf(u.v[nt][r], 0.f);
    wave_fence();
    ct_to_scr(lds.scr, u, c, g);
    wave_fence();
    scr_to_rows(a.o1, a.ld1, lds.scr, row0, a.T, lane);
    CT y = gemm_w<PREC>(scr_to_a<PREC>(lds.scr, c, g), lds.w[2], c, g);
    ct_add_bias(y, a.b[2], c);
    ct_dropmask(y, key2, a.drop, (uint32_t)row0 + a.row_offset, c, g);
    ct_add(y, a2);
    ct_add(y, rows_to_ct(lds.scr, dn_rows, lane, c, g));
    ct_mask_rows(y, a.ids, row0, a.T, g);
    store_ct(lds.scr, a.o2, a.ld2, y, row0, a.T, lane, c, g);
  }
}

// ---- final: f = LN_last(x) ; pos/neg logits ; [k2, v2] = f Wkv^T + b for up to two decoder layers ---------------
// W0,W1 = Wk2,Wv2 of layer A (-> o2, ld 128) ; W2,W3 = layer B (-> o3) ; nkv = number of layers (0, 1 or 2)
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_final_fwd(FwdChainArgs a) {
  FWD_PROLOGUE(4)
  // compile-time loop bounds: a runtime index into the argument arrays W[] / b[] puts them into scratch memory
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < 2 * a.nkv) stage_wimg<PREC, NW * 64>(lds.w[j], a.W[j], false);
  __syncthreads();
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    int ipr[4] = {0, 0, 0, 0}, inr[4] = {0, 0, 0, 0};      // the four rows' pos / neg ids, requested with the rows (read inside the loop below they were one more round trip in front of the table rows)
    if (a.pos_logits) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        if (row < a.T) { ipr[r] = a.pos[row]; inr[r] = a.neg[row]; }
      }
    }
    wave_fence();
    rows_to_scr(lds.scr, a.x, 64, row0, a.T, lane);
    wave_fence();
    const CT x = scr_to_ct(lds.scr, c, g);
    LnStat st;
    const CT f = ln_apply(ln_xhat(x, a.ln_eps, st), a.gamma, a.beta, c);
    if (a.pos_logits) {
      // pos/neg logits = sum_d f * E[pos|neg]                      (sasrec/model.py:72-76)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        float sp = 0.f, sn = 0.f;
        if (row < a.T) {
          const int ip = ipr[r], in = inr[r];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            sp += f.v[nt][r] * a.E[(size_t)ip * 64 + 16 * nt + c];
            sn += f.v[nt][r] * a.E[(size_t)in * 64 + 16 * nt + c];
          }
        }
        sp = row_sum16(sp);
        sn = row_sum16(sn);
        if (c == 0 && row < a.T) { a.pos_logits[row] = sp; a.neg_logits[row] = sn; }
      }
    }
    wave_fence();
    ct_to_scr(lds.scr, f, c, g);
    wave_fence();
    if (a.o0) scr_to_rows(a.o0, a.ld0, lds.scr, row0, a.T, lane);
    if (a.nkv > 0) {
      const AFrags<PREC> af = scr_to_a<PREC>(lds.scr, c, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= 2 * a.nkv) break;
        CT y = gemm_w<PREC>(af, lds.w[j], c, g);
        ct_add_bias(y, a.b[j], c);
        float* dst = (j < 2 ? a.o2 : a.o3) + 64 * (j & 1);
        store_ct(lds.scr, dst, j < 2 ? a.ld2 : a.ld3, y, row0, a.T, lane, c, g);
      }
    }
  }
}

}  // namespace adt
