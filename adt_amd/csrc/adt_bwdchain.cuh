// Fused backward row chains (wave-local, see adt_wave.cuh): the reverse pass of everything between two
// attention cores of EncoderLayer / DecoderLayer (sasrec/modules.py:644-655, :666-677), i.e. the autograd of
// out_proj + residual + LayerNorm + PointWiseFeedForward + mask, of LayerNorm + packed in-projection, and of
// the decoder's out_proj / q-projection / kv-projection chain.  Weight and bias gradients are accumulated in
// LDS images per workgroup (ds_add_f32) and flushed once with global atomics.
#pragma once
#include "adt_bwdchain_args.h"
#include "adt_wave.cuh"

namespace adt {

template <int PREC, int NW, int NWT>
struct BwdLds {
  typedef typename WImg<PREC>::T WT;
  static constexpr int WIMG = 64 * WImg<PREC>::RS;           // elements per weight image
  static constexpr size_t bytes = NWT * WIMG * sizeof(WT) + NWT * DW_IMG * sizeof(float) + 8 * 64 * sizeof(float) +
                                  NW * WV_SCR * sizeof(float);
  WT* w[4]; float* dw[4]; float* vec[8]; float* scr;
  __device__ BwdLds(unsigned char* base, int wave) {
    WT* pw = reinterpret_cast<WT*>(base);
    for (int i = 0; i < NWT; ++i) w[i] = pw + i * WIMG;
    float* pf = reinterpret_cast<float*>(base + NWT * WIMG * sizeof(WT));
    for (int i = 0; i < NWT; ++i) dw[i] = pf + i * DW_IMG;
    pf += NWT * DW_IMG;
    for (int i = 0; i < 8; ++i) vec[i] = pf + i * 64;
    scr = pf + 8 * 64 + wave * WV_SCR;
  }
  __device__ void zero_acc(int nthreads) {
    for (int i = threadIdx.x; i < NWT * DW_IMG + 8 * 64; i += nthreads) dw[0][i] = 0.f;
  }
};

ADT_DEVICE_INLINE void ct_mask_rows(CT& t, const int* ids, int row0, int T, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    const bool dead = row >= T || ids[row] == 0;
    if (dead) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) t.v[nt][r] = 0.f;
    }
  }
}

ADT_DEVICE_INLINE void ct_dropmask(CT& t, uint32_t key, const DropCfg& d, uint32_t row_base, int c, int g) {
  if (!d.thr) return;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t idx = (row_base + (uint32_t)(4 * g + r)) * 64u + (uint32_t)(16 * nt + c);
      t.v[nt][r] = adt_keep(key, idx, d.thr) ? t.v[nt][r] * d.scale : 0.f;
    }
}

ADT_DEVICE_INLINE void ct_add(CT& a, const CT& b) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) a.v[nt] += b.v[nt];
}

// FFN reverse shared by encoder and decoder:  y = mask(R + drop2(conv2 relu(drop1(conv1 xin)))).
// Returns g = masked upstream gradient and dxin_ffn = gradient reaching xin through the FFN.
template <int PREC>
ADT_DEVICE_INLINE void ffn_bwd_tile(const BwdChainArgs& a, float* scr, const typename WImg<PREC>::T* W2t,
                                    const typename WImg<PREC>::T* W1t, float* dW2, float* db2, float* dW1, float* db1,
                                    const CT& xin, const RowRegs& gy_rows, const RowRegs& u_rows, uint32_t key1, uint32_t key2,
                                    int row0, int lane, int c, int g, CT& gout, CT& dxin) {
  wave_fence();
  rows_put(scr, gy_rows, lane);
  wave_fence();
  gout = scr_to_ct(scr, c, g);
  ct_mask_rows(gout, a.ids, row0, a.T, g);
  CT df = gout;
  ct_dropmask(df, key2, a.drop, (uint32_t)row0 + a.row_offset, c, g);
  wave_fence();
  rows_put(scr, u_rows, lane);
  wave_fence();
  const CT u = scr_to_ct(scr, c, g);
  dw_accum<PREC>(dW2, db2, df, u, c, g);
  CT dt = gemm_w<PREC>(ct_to_a<PREC>(scr, df, c, g), W2t, c, g);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (!(u.v[nt][r] > 0.f)) dt.v[nt][r] = 0.f;
  ct_dropmask(dt, key1, a.drop, (uint32_t)row0 + a.row_offset, c, g);
  dw_accum<PREC>(dW1, db1, dt, xin, c, g);
  dxin = gemm_w<PREC>(ct_to_a<PREC>(scr, dt, c, g), W1t, c, g);
}

// ---- encoder: y = mask(h2 + FFN(h2)), h2 = LN2(h), h = Qn + o Wo^T + bo ------------------------------------
// W0 = conv2, W1 = conv1, W2 = out_proj ; out0 = dh (gradient wrt h == wrt Qn residual), out1 = dO
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_enc_post_bwd(BwdChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  BwdLds<PREC, NW, 3> lds(smem_raw, w);
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W2, true);
  lds.zero_acc(NW * 64);
  __syncthreads();
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  const int ntiles = (a.T + 15) / 16;
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += gridDim.x * NW) {
    const int row0 = tile * 16;
    const RowRegs gy_rows = rows_load(a.gy, 64, row0, a.T, lane);
    const RowRegs u_rows = rows_load(a.u, 64, row0, a.T, lane);
    const RowRegs h_rows = rows_load(a.xin, 64, row0, a.T, lane);
    const RowRegs o_rows = rows_load(a.o, 64, row0, a.T, lane);
    wave_fence();
    rows_put(lds.scr, h_rows, lane);
    wave_fence();
    const CT h = scr_to_ct(lds.scr, c, g);
    LnStat st;
    const CT xhat = ln_xhat(h, a.ln_eps, st);
    const CT h2 = ln_apply(xhat, a.gamma, a.beta, c);
    CT gm, dh2;
    ffn_bwd_tile<PREC>(a, lds.scr, lds.w[0], lds.w[1], lds.dw[0], lds.vec[0], lds.dw[1], lds.vec[1], h2, gy_rows, u_rows, key1, key2,
                       row0, lane, c, g, gm, dh2);
    ct_add(dh2, gm);
    const CT dh = ln_bwd_ct(dh2, xhat, st, a.gamma, lds.vec[3], lds.vec[4], c, g);
    store_ct(lds.scr, a.out0, 64, dh, row0, a.T, lane, c, g);
    wave_fence();
    rows_put(lds.scr, o_rows, lane);
    wave_fence();
    const CT o = scr_to_ct(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[2], lds.vec[2], dh, o, c, g);
    const CT dO = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, dh, c, g), lds.w[2], c, g);
    store_ct(lds.scr, a.out1, 64, dO, row0, a.T, lane, c, g);
  }
  __syncthreads();
  flush_dw<NW * 64>(a.dW0, lds.dw[0]); flush_dw<NW * 64>(a.dW1, lds.dw[1]); flush_dw<NW * 64>(a.dW2, lds.dw[2]);
  flush_vec<NW * 64>(a.db0, lds.vec[0]); flush_vec<NW * 64>(a.db1, lds.vec[1]); flush_vec<NW * 64>(a.db2, lds.vec[2]);
  flush_vec<NW * 64>(a.dgamma, lds.vec[3]); flush_vec<NW * 64>(a.dbeta, lds.vec[4]);
}

// ---- decoder: y = mask(Dn + a2 + FFN(a2)), a2 = o2 Wo2^T + b --------------------------------------------------
// W0 = conv2, W1 = conv1, W2 = enc_attn.out_proj ; out0 = dO2
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_post_bwd(BwdChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  BwdLds<PREC, NW, 3> lds(smem_raw, w);
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W2, true);
  lds.zero_acc(NW * 64);
  __syncthreads();
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  const int ntiles = (a.T + 15) / 16;
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += gridDim.x * NW) {
    const int row0 = tile * 16;
    const RowRegs gy_rows = rows_load(a.gy, 64, row0, a.T, lane);
    const RowRegs u_rows = rows_load(a.u, 64, row0, a.T, lane);
    const RowRegs a2_rows = rows_load(a.xin, 64, row0, a.T, lane);
    const RowRegs o_rows = rows_load(a.o, 64, row0, a.T, lane);
    wave_fence();
    rows_put(lds.scr, a2_rows, lane);
    wave_fence();
    const CT a2 = scr_to_ct(lds.scr, c, g);
    CT gm, da2;
    ffn_bwd_tile<PREC>(a, lds.scr, lds.w[0], lds.w[1], lds.dw[0], lds.vec[0], lds.dw[1], lds.vec[1], a2, gy_rows, u_rows, key1, key2,
                       row0, lane, c, g, gm, da2);
    ct_add(da2, gm);
    wave_fence();
    rows_put(lds.scr, o_rows, lane);
    wave_fence();
    const CT o = scr_to_ct(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[2], lds.vec[2], da2, o, c, g);
    const CT dO = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, da2, c, g), lds.w[2], c, g);
    store_ct(lds.scr, a.out0, 64, dO, row0, a.T, lane, c, g);
  }
  __syncthreads();
  flush_dw<NW * 64>(a.dW0, lds.dw[0]); flush_dw<NW * 64>(a.dW1, lds.dw[1]); flush_dw<NW * 64>(a.dW2, lds.dw[2]);
  flush_vec<NW * 64>(a.db0, lds.vec[0]); flush_vec<NW * 64>(a.db1, lds.vec[1]); flush_vec<NW * 64>(a.db2, lds.vec[2]);
}

// ---- LayerNorm + packed in-projection reverse (encoder: q from LN(x), k/v from x; decoder: all from LN(x)) ----
// W0,W1,W2 = Wq,Wk,Wv ; dqkv = packed gradient (T x 192) ; ENC: dh = residual-path gradient wrt LN output
// out0 = gradient wrt x (acc0: add to what is there) ; DEC: + mask(gy) added to the LN-output gradient
template <int PREC, int NW, bool ENC>
__global__ __launch_bounds__(NW * 64) void k_pre_bwd(BwdChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  BwdLds<PREC, NW, 3> lds(smem_raw, w);
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W2, true);
  lds.zero_acc(NW * 64);
  __syncthreads();
  const int ntiles = (a.T + 15) / 16;
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += gridDim.x * NW) {
    const int row0 = tile * 16;
    const RowRegs dq_rows = rows_load(a.dqkv, a.lddqkv, row0, a.T, lane);
    const RowRegs dk_rows = rows_load(a.dqkv + 64, a.lddqkv, row0, a.T, lane);
    const RowRegs dv_rows = rows_load(a.dqkv + 128, a.lddqkv, row0, a.T, lane);
    const RowRegs x_rows = rows_load(a.xin, 64, row0, a.T, lane);
    const RowRegs r_rows = rows_load(ENC ? a.dh : a.gy, 64, row0, a.T, lane);
    wave_fence();
    rows_put(lds.scr, x_rows, lane);
    wave_fence();
    const CT x = scr_to_ct(lds.scr, c, g);
    LnStat st;
    const CT xhat = ln_xhat(x, a.ln_eps, st);
    const CT xn = ln_apply(xhat, a.gamma, a.beta, c);
    wave_fence();
    rows_put(lds.scr, dq_rows, lane);
    wave_fence();
    const CT dq = scr_to_ct(lds.scr, c, g);
    const AFrags<PREC> aq = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[0], lds.vec[0], dq, xn, c, g);
    CT dn = gemm_w<PREC>(aq, lds.w[0], c, g);          // gradient wrt the LN output
    wave_fence();
    rows_put(lds.scr, dk_rows, lane);
    wave_fence();
    const CT dk = scr_to_ct(lds.scr, c, g);
    const AFrags<PREC> ak = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[1], lds.vec[1], dk, ENC ? x : xn, c, g);
    CT dkv = gemm_w<PREC>(ak, lds.w[1], c, g);
    wave_fence();
    rows_put(lds.scr, dv_rows, lane);
    wave_fence();
    const CT dv = scr_to_ct(lds.scr, c, g);
    const AFrags<PREC> av = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[2], lds.vec[2], dv, ENC ? x : xn, c, g);
    ct_add(dkv, gemm_w<PREC>(av, lds.w[2], c, g));
    wave_fence();
    rows_put(lds.scr, r_rows, lane);
    wave_fence();
    CT res = scr_to_ct(lds.scr, c, g);
    if (!ENC) ct_mask_rows(res, a.ids, row0, a.T, g);
    ct_add(dn, res);
    CT dx;
    if (ENC) {
      dx = ln_bwd_ct(dn, xhat, st, a.gamma, lds.vec[3], lds.vec[4], c, g);
      ct_add(dx, dkv);                                  // k, v read the raw x (sasrec/modules.py:647)
    } else {
      ct_add(dn, dkv);
      dx = ln_bwd_ct(dn, xhat, st, a.gamma, lds.vec[3], lds.vec[4], c, g);
    }
    store_ct(lds.scr, a.out0, 64, dx, row0, a.T, lane, c, g, a.acc0 != 0);
  }
  __syncthreads();
  flush_dw<NW * 64>(a.dW0, lds.dw[0]); flush_dw<NW * 64>(a.dW1, lds.dw[1]); flush_dw<NW * 64>(a.dW2, lds.dw[2]);
  flush_vec<NW * 64>(a.db0, lds.vec[0]); flush_vec<NW * 64>(a.db1, lds.vec[1]); flush_vec<NW * 64>(a.db2, lds.vec[2]);
  flush_vec<NW * 64>(a.dgamma, lds.vec[3]); flush_vec<NW * 64>(a.dbeta, lds.vec[4]);
}

// ---- decoder middle: q2 = a1 Wq^T, a1 = o1 Wo1^T ; [k2, v2] = f Wkv^T -------------------------------------------
// W0 = Wq2, W1 = Wo1 (slf out_proj), W2 = Wk2, W3 = Wv2 ; dqkv = dq2 (ld 64), dkv2 (T x 128)
// out0 = dO1 ; out1 = g_f (+=)
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_mid_bwd(BwdChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  BwdLds<PREC, NW, 4> lds(smem_raw, w);
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W2, true);
  stage_wimg<PREC, NW * 64>(lds.w[3], a.W3, true);
  lds.zero_acc(NW * 64);
  __syncthreads();
  const int ntiles = (a.T + 15) / 16;
  for (int tile = blockIdx.x * NW + w; tile < ntiles; tile += gridDim.x * NW) {
    const int row0 = tile * 16;
    const RowRegs dq_rows = rows_load(a.dqkv, a.lddqkv, row0, a.T, lane);
    const RowRegs a1_rows = rows_load(a.xin, 64, row0, a.T, lane);
    const RowRegs o1_rows = rows_load(a.o, 64, row0, a.T, lane);
    wave_fence();
    rows_put(lds.scr, a1_rows, lane);
    wave_fence();
    const CT a1 = scr_to_ct(lds.scr, c, g);
    wave_fence();
    rows_put(lds.scr, dq_rows, lane);
    wave_fence();
    const CT dq = scr_to_ct(lds.scr, c, g);
    const AFrags<PREC> aq = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[0], lds.vec[0], dq, a1, c, g);
    const CT da1 = gemm_w<PREC>(aq, lds.w[0], c, g);
    wave_fence();
    rows_put(lds.scr, o1_rows, lane);
    wave_fence();
    const CT o1 = scr_to_ct(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[1], lds.vec[1], da1, o1, c, g);
    // second half of the tile's inputs is requested only now: keeps the live register set under 256
    const RowRegs dk_rows = rows_load(a.dkv2, 128, row0, a.T, lane);
    const RowRegs dv_rows = rows_load(a.dkv2 + 64, 128, row0, a.T, lane);
    const RowRegs f_rows = rows_load(a.f, 64, row0, a.T, lane);
    const CT dO1 = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, da1, c, g), lds.w[1], c, g);
    store_ct(lds.scr, a.out0, 64, dO1, row0, a.T, lane, c, g);
    wave_fence();
    rows_put(lds.scr, f_rows, lane);
    wave_fence();
    const CT f = scr_to_ct(lds.scr, c, g);
    wave_fence();
    rows_put(lds.scr, dk_rows, lane);
    wave_fence();
    const CT dk = scr_to_ct(lds.scr, c, g);
    const AFrags<PREC> ak = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[2], lds.vec[2], dk, f, c, g);
    CT df = gemm_w<PREC>(ak, lds.w[2], c, g);
    wave_fence();
    rows_put(lds.scr, dv_rows, lane);
    wave_fence();
    const CT dv = scr_to_ct(lds.scr, c, g);
    const AFrags<PREC> av = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(lds.dw[3], lds.vec[3], dv, f, c, g);
    ct_add(df, gemm_w<PREC>(av, lds.w[3], c, g));
    store_ct(lds.scr, a.out1, 64, df, row0, a.T, lane, c, g, true);
  }
  __syncthreads();
  flush_dw<NW * 64>(a.dW0, lds.dw[0]); flush_dw<NW * 64>(a.dW1, lds.dw[1]); flush_dw<NW * 64>(a.dW2, lds.dw[2]);
  flush_dw<NW * 64>(a.dW3, lds.dw[3]);
  flush_vec<NW * 64>(a.db0, lds.vec[0]); flush_vec<NW * 64>(a.db1, lds.vec[1]); flush_vec<NW * 64>(a.db2, lds.vec[2]);
  flush_vec<NW * 64>(a.db3, lds.vec[3]);
}

}  // namespace adt
