// Fused backward row chains (wave-local, see adt_wave.cuh): the reverse pass of everything between two
// attention cores of EncoderLayer / DecoderLayer (sasrec/modules.py:644-655, :666-677), i.e. the autograd of
// out_proj + residual + LayerNorm + PointWiseFeedForward + mask, of LayerNorm + packed in-projection, and of
// the decoder's out_proj / q-projection / kv-projection chain.  Weight and bias gradients are accumulated in
// LDS images per workgroup (ds_add_f32) and flushed once with global atomics.
#pragma once
#include "adt_bwdchain_args.h"
#include "adt_wave.cuh"

namespace adt {

// LDS carve-up: NWT weight images, then a region shared by the per-wave layout scratch (during the tile loop)
// and the NW per-wave reduction images (after it).
template <int PREC, int NW, int NWT>
struct BwdLds {
  typedef typename WImg<PREC>::T WT;
  static constexpr int WIMG = 64 * WImg<PREC>::RS;           // elements per weight image
  static constexpr size_t bytes = NWT * WIMG * sizeof(WT) + (size_t)NW * DW_IMG * sizeof(float);
  WT* w[4]; float* red; float* scr;
  __device__ BwdLds(unsigned char* base, int wave) {
    WT* pw = reinterpret_cast<WT*>(base);
    for (int i = 0; i < NWT; ++i) w[i] = pw + i * WIMG;
    red = reinterpret_cast<float*>(base + NWT * WIMG * sizeof(WT));
    scr = red + wave * DW_IMG;      // a wave's scratch lives at the head of its own reduction image
  }
};

// FFN reverse shared by encoder and decoder:  y = mask(R + drop2(conv2 relu(drop1(conv1 xin)))).
// Returns g = masked upstream gradient and dxin = gradient reaching xin through the FFN.
template <int PREC>
ADT_DEVICE_INLINE void ffn_bwd_tile(const BwdChainArgs& a, float* scr, const typename WImg<PREC>::T* W2t,
                                    const typename WImg<PREC>::T* W1t, WAcc& dW2, WAcc& dW1, const CT& xin,
                                    const RowRegs& gy_rows, const RowRegs& u_rows, uint32_t key1, uint32_t key2, int row0,
                                    int lane, int c, int g, CT& gout, CT& dxin) {
  gout = rows_to_ct(scr, gy_rows, lane, c, g);
  ct_mask_rows(gout, a.ids, row0, a.T, g);
  CT df = gout;
  ct_dropmask(df, key2, a.drop, (uint32_t)row0 + a.row_offset, c, g);
  const CT u = rows_to_ct(scr, u_rows, lane, c, g);
  dw_accum<PREC>(dW2, df, u);
  CT dt = gemm_w<PREC>(ct_to_a<PREC>(scr, df, c, g), W2t, c, g);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (!(u.v[nt][r] > 0.f)) dt.v[nt][r] = 0.f;
  ct_dropmask(dt, key1, a.drop, (uint32_t)row0 + a.row_offset, c, g);
  dw_accum<PREC>(dW1, dt, xin);
  dxin = gemm_w<PREC>(ct_to_a<PREC>(scr, dt, c, g), W1t, c, g);
}

#define BWD_PROLOGUE(NWT_)                                                                         \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];                         \
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;           \
  BwdLds<PREC, NW, NWT_> lds(smem_raw, w);                                                         \
  const int ntiles = (a.T + 15) / 16;                                                              \
  const int tstride = gridDim.x * NW;                                                              \
  int tile = blockIdx.x * NW + w;

// ---- encoder: y = mask(h2 + FFN(h2)), h2 = LN2(h), h = Qn + o Wo^T + bo ------------------------------------
// W0 = conv2, W1 = conv1, W2 = out_proj ; out0 = dh (gradient wrt h == wrt Qn residual), out1 = dO
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_enc_post_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(3)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W2, true);
  __syncthreads();
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  WAcc dW2, dW1, dWo;
  VAcc dgm, dbt;
  wacc_zero(dW2); wacc_zero(dW1); wacc_zero(dWo); vacc_zero(dgm); vacc_zero(dbt);
  RowRegs gy_rows = rows_load(a.gy, 64, tile * 16, a.T, lane);
  RowRegs u_rows = rows_load(a.u, 64, tile * 16, a.T, lane);
  RowRegs h_rows = rows_load(a.xin, 64, tile * 16, a.T, lane);
  RowRegs o_rows = rows_load(a.o, 64, tile * 16, a.T, lane);
  for (; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    const CT h = rows_to_ct(lds.scr, h_rows, lane, c, g);
    LnStat st;
    const CT xhat = ln_xhat(h, a.ln_eps, st);
    const CT h2 = ln_apply(xhat, a.gamma, a.beta, c);
    CT gm, dh2;
    ffn_bwd_tile<PREC>(a, lds.scr, lds.w[0], lds.w[1], dW2, dW1, h2, gy_rows, u_rows, key1, key2, row0, lane, c, g, gm, dh2);
    const CT o = rows_to_ct(lds.scr, o_rows, lane, c, g);
    // the next tile's inputs are requested now and consumed one iteration later
    const int nrow0 = (tile + tstride) * 16;
    gy_rows = rows_load(a.gy, 64, nrow0, a.T, lane);
    u_rows = rows_load(a.u, 64, nrow0, a.T, lane);
    h_rows = rows_load(a.xin, 64, nrow0, a.T, lane);
    o_rows = rows_load(a.o, 64, nrow0, a.T, lane);
    ct_add(dh2, gm);
    const CT dh = ln_bwd_ct(dh2, xhat, st, a.gamma, dgm, dbt, c, g);
    store_ct(lds.scr, a.out0, 64, dh, row0, a.T, lane, c, g);
    dw_accum<PREC>(dWo, dh, o);
    const CT dO = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, dh, c, g), lds.w[2], c, g);
    store_ct(lds.scr, a.out1, 64, dO, row0, a.T, lane, c, g);
  }
  wacc_flush<NW>(dW2, lds.red, a.dW0, a.db0, w, c, g);
  wacc_flush<NW>(dW1, lds.red, a.dW1, a.db1, w, c, g);
  wacc_flush<NW>(dWo, lds.red, a.dW2, a.db2, w, c, g);
  vacc_flush(dgm, a.dgamma, c, g);
  vacc_flush(dbt, a.dbeta, c, g);
}

// ---- decoder: y = mask(Dn + a2 + FFN(a2)), a2 = o2 Wo2^T + b --------------------------------------------------
// W0 = conv2, W1 = conv1, W2 = enc_attn.out_proj ; out0 = dO2
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_post_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(3)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W2, true);
  __syncthreads();
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  WAcc dW2, dW1, dWo;
  wacc_zero(dW2); wacc_zero(dW1); wacc_zero(dWo);
  RowRegs gy_rows = rows_load(a.gy, 64, tile * 16, a.T, lane);
  RowRegs u_rows = rows_load(a.u, 64, tile * 16, a.T, lane);
  RowRegs a2_rows = rows_load(a.xin, 64, tile * 16, a.T, lane);
  RowRegs o_rows = rows_load(a.o, 64, tile * 16, a.T, lane);
  for (; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    const CT a2 = rows_to_ct(lds.scr, a2_rows, lane, c, g);
    CT gm, da2;
    ffn_bwd_tile<PREC>(a, lds.scr, lds.w[0], lds.w[1], dW2, dW1, a2, gy_rows, u_rows, key1, key2, row0, lane, c, g, gm, da2);
    const CT o = rows_to_ct(lds.scr, o_rows, lane, c, g);
    const int nrow0 = (tile + tstride) * 16;
    gy_rows = rows_load(a.gy, 64, nrow0, a.T, lane);
    u_rows = rows_load(a.u, 64, nrow0, a.T, lane);
    a2_rows = rows_load(a.xin, 64, nrow0, a.T, lane);
    o_rows = rows_load(a.o, 64, nrow0, a.T, lane);
    ct_add(da2, gm);
    dw_accum<PREC>(dWo, da2, o);
    const CT dO = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, da2, c, g), lds.w[2], c, g);
    store_ct(lds.scr, a.out0, 64, dO, row0, a.T, lane, c, g);
  }
  wacc_flush<NW>(dW2, lds.red, a.dW0, a.db0, w, c, g);
  wacc_flush<NW>(dW1, lds.red, a.dW1, a.db1, w, c, g);
  wacc_flush<NW>(dWo, lds.red, a.dW2, a.db2, w, c, g);
}

// ---- LayerNorm + packed in-projection reverse (encoder: q from LN(x), k/v from x; decoder: all from LN(x)) ----
// W0,W1,W2 = Wq,Wk,Wv ; dqkv = packed gradient (T x 192) ; ENC: dh = residual-path gradient wrt LN output
// out0 = gradient wrt x (acc0: add to what is there) ; DEC: + mask(gy) added to the LN-output gradient
template <int PREC, int NW, bool ENC>
__global__ __launch_bounds__(NW * 64) void k_pre_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(3)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  stage_wimg<PREC, NW * 64>(lds.w[2], a.W2, true);
  __syncthreads();
  WAcc dWq, dWk, dWv;
  VAcc dgm, dbt;
  wacc_zero(dWq); wacc_zero(dWk); wacc_zero(dWv); vacc_zero(dgm); vacc_zero(dbt);
  const float* rsrc = ENC ? a.dh : a.gy;
  RowRegs dq_rows = rows_load(a.dqkv, a.lddqkv, tile * 16, a.T, lane);
  RowRegs dk_rows = rows_load(a.dqkv + 64, a.lddqkv, tile * 16, a.T, lane);
  RowRegs dv_rows = rows_load(a.dqkv + 128, a.lddqkv, tile * 16, a.T, lane);
  RowRegs x_rows = rows_load(a.xin, 64, tile * 16, a.T, lane);
  for (; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    // requested at the top of the iteration, consumed at its end (not carried across iterations: registers)
    const RowRegs r_rows = rows_load(rsrc, 64, row0, a.T, lane);
    const CT x = rows_to_ct(lds.scr, x_rows, lane, c, g);
    LnStat st;
    const CT xhat = ln_xhat(x, a.ln_eps, st);
    const CT xn = ln_apply(xhat, a.gamma, a.beta, c);
    const CT dq = rows_to_ct(lds.scr, dq_rows, lane, c, g);
    const AFrags<PREC> aq = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(dWq, dq, xn);
    CT dn = gemm_w<PREC>(aq, lds.w[0], c, g);          // gradient wrt the LN output
    const CT dk = rows_to_ct(lds.scr, dk_rows, lane, c, g);
    const AFrags<PREC> ak = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(dWk, dk, ENC ? x : xn);
    CT dkv = gemm_w<PREC>(ak, lds.w[1], c, g);
    const CT dv = rows_to_ct(lds.scr, dv_rows, lane, c, g);
    const AFrags<PREC> av = scr_to_a<PREC>(lds.scr, c, g);
    dw_accum<PREC>(dWv, dv, ENC ? x : xn);
    ct_add(dkv, gemm_w<PREC>(av, lds.w[2], c, g));
    CT res = rows_to_ct(lds.scr, r_rows, lane, c, g);
    const int nrow0 = (tile + tstride) * 16;
    dq_rows = rows_load(a.dqkv, a.lddqkv, nrow0, a.T, lane);
    dk_rows = rows_load(a.dqkv + 64, a.lddqkv, nrow0, a.T, lane);
    dv_rows = rows_load(a.dqkv + 128, a.lddqkv, nrow0, a.T, lane);
    x_rows = rows_load(a.xin, 64, nrow0, a.T, lane);
    if (!ENC) ct_mask_rows(res, a.ids, row0, a.T, g);
    ct_add(dn, res);
    CT dx;
    if (ENC) {
      dx = ln_bwd_ct(dn, xhat, st, a.gamma, dgm, dbt, c, g);
      ct_add(dx, dkv);                                  // k, v read the raw x (sasrec/modules.py:647)
    } else {
      ct_add(dn, dkv);
      dx = ln_bwd_ct(dn, xhat, st, a.gamma, dgm, dbt, c, g);
    }
    store_ct(lds.scr, a.out0, 64, dx, row0, a.T, lane, c, g, a.acc0 != 0);
  }
  wacc_flush<NW>(dWq, lds.red, a.dW0, a.db0, w, c, g);
  wacc_flush<NW>(dWk, lds.red, a.dW1, a.db1, w, c, g);
  wacc_flush<NW>(dWv, lds.red, a.dW2, a.db2, w, c, g);
  vacc_flush(dgm, a.dgamma, c, g);
  vacc_flush(dbt, a.dbeta, c, g);
}

// ---- decoder middle, part 1: q2 = a1 Wq^T + b, a1 = o1 Wo1^T + b --------------------------------------------------
// W0 = Wq2, W1 = Wo1 (slf out_proj) ; dqkv = dq2 (ld 64) ; xin = a1, o = o1 ; out0 = dO1
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_mid_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(2)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  __syncthreads();
  WAcc dWq, dWo;
  wacc_zero(dWq); wacc_zero(dWo);
  RowRegs dq_rows = rows_load(a.dqkv, a.lddqkv, tile * 16, a.T, lane);
  RowRegs a1_rows = rows_load(a.xin, 64, tile * 16, a.T, lane);
  RowRegs o1_rows = rows_load(a.o, 64, tile * 16, a.T, lane);
  for (; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    const CT a1 = rows_to_ct(lds.scr, a1_rows, lane, c, g);
    const CT dq = rows_to_ct(lds.scr, dq_rows, lane, c, g);
    const AFrags<PREC> aq = scr_to_a<PREC>(lds.scr, c, g);
    const CT o1 = rows_to_ct(lds.scr, o1_rows, lane, c, g);
    const int nrow0 = (tile + tstride) * 16;
    dq_rows = rows_load(a.dqkv, a.lddqkv, nrow0, a.T, lane);
    a1_rows = rows_load(a.xin, 64, nrow0, a.T, lane);
    o1_rows = rows_load(a.o, 64, nrow0, a.T, lane);
    dw_accum<PREC>(dWq, dq, a1);
    const CT da1 = gemm_w<PREC>(aq, lds.w[0], c, g);
    dw_accum<PREC>(dWo, da1, o1);
    const CT dO1 = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, da1, c, g), lds.w[1], c, g);
    store_ct(lds.scr, a.out0, 64, dO1, row0, a.T, lane, c, g);
  }
  wacc_flush<NW>(dWq, lds.red, a.dW0, a.db0, w, c, g);
  wacc_flush<NW>(dWo, lds.red, a.dW1, a.db1, w, c, g);
}

// ---- decoder middle, part 2: [k2, v2] = f Wkv^T + b --------------------------------------------------------------
// W0 = Wk2, W1 = Wv2 ; dkv2 (T x 128) ; f = log_feats ; out0 = g_f (+=)
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_kv_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(2)
  stage_wimg<PREC, NW * 64>(lds.w[0], a.W0, true);
  stage_wimg<PREC, NW * 64>(lds.w[1], a.W1, true);
  __syncthreads();
  WAcc dWk, dWv;
  wacc_zero(dWk); wacc_zero(dWv);
  RowRegs dk_rows = rows_load(a.dkv2, 128, tile * 16, a.T, lane);
  RowRegs dv_rows = rows_load(a.dkv2 + 64, 128, tile * 16, a.T, lane);
  RowRegs f_rows = rows_load(a.f, 64, tile * 16, a.T, lane);
  for (; tile < ntiles; tile += tstride) {
    const int row0 = tile * 16;
    const CT f = rows_to_ct(lds.scr, f_rows, lane, c, g);
    const CT dk = rows_to_ct(lds.scr, dk_rows, lane, c, g);
    const AFrags<PREC> ak = scr_to_a<PREC>(lds.scr, c, g);
    const CT dv = rows_to_ct(lds.scr, dv_rows, lane, c, g);
    const AFrags<PREC> av = scr_to_a<PREC>(lds.scr, c, g);
    const int nrow0 = (tile + tstride) * 16;
    dk_rows = rows_load(a.dkv2, 128, nrow0, a.T, lane);
    dv_rows = rows_load(a.dkv2 + 64, 128, nrow0, a.T, lane);
    f_rows = rows_load(a.f, 64, nrow0, a.T, lane);
    dw_accum<PREC>(dWk, dk, f);
    dw_accum<PREC>(dWv, dv, f);
    CT df = gemm_w<PREC>(ak, lds.w[0], c, g);
    ct_add(df, gemm_w<PREC>(av, lds.w[1], c, g));
    store_ct(lds.scr, a.out0, 64, df, row0, a.T, lane, c, g, true);
  }
  wacc_flush<NW>(dWk, lds.red, a.dW0, a.db0, w, c, g);
  wacc_flush<NW>(dWv, lds.red, a.dW1, a.db1, w, c, g);
}

}  // namespace adt
