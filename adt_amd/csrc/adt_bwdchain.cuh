// Fused backward row chains (wave-local, see adt_wave.cuh): the reverse pass of everything between two
// attention cores of EncoderLayer / DecoderLayer (sasrec/modules.py:644-655, :666-677), i.e. the autograd of
// out_proj + residual + LayerNorm + PointWiseFeedForward + mask, of LayerNorm + packed in-projection, and of
// the decoder's out_proj / q-projection / kv-projection chain.  Weight and bias gradients are accumulated in
// LDS images per workgroup (ds_add_f32) and flushed once with global atomics.
#pragma once
#include "adt_bwdchain_args.h"
#include "adt_misc.cuh"
#include "adt_wave.cuh"

namespace adt {

// LDS carve-up: NWT weight images | two cooperative exchange areas (Coop) | per-wave layout scratch.
template <int PREC, int NW, int NWT>
struct BwdLds {
  typedef typename WImg<PREC>::T WT;
  static constexpr int WIMG = 64 * WImg<PREC>::RS;           // elements per weight image
  static constexpr size_t wbytes = NWT * WIMG * sizeof(WT);
  static constexpr int CLS = 2 * 16 * 16;                     // per wave: rec + drec rows of a tile (H <= 4)
  static constexpr size_t bytes = wbytes + Coop<PREC, NW>::bytes + (size_t)NW * (WV_SCR + CLS) * sizeof(float);
  WT* w[4]; void* coop; float* scr; float* cls;
  __device__ BwdLds(unsigned char* base, int wave) {
    WT* pw = reinterpret_cast<WT*>(base);
    for (int i = 0; i < NWT; ++i) w[i] = pw + i * WIMG;
    coop = base + wbytes;
    scr = reinterpret_cast<float*>(base + wbytes + Coop<PREC, NW>::bytes) + wave * WV_SCR;
    cls = reinterpret_cast<float*>(base + wbytes + Coop<PREC, NW>::bytes) + NW * WV_SCR + wave * CLS;
  }
};

template <int N>
ADT_DEVICE_INLINE void acc_zero(f32x4 (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

ADT_DEVICE_INLINE void bwd_replica(BwdChainArgs& a) {
  if (a.nrep <= 1) return;
  const size_t off = (size_t)(blockIdx.x % a.nrep) * a.rep_stride;
  float** const ptrs[] = {&a.dW0, &a.dW1, &a.dW2, &a.dW3, &a.db0, &a.db1, &a.db2, &a.db3, &a.dgamma, &a.dbeta, &a.dWs, &a.dbs};
#pragma unroll
  for (int i = 0; i < 12; ++i)
    if (*ptrs[i]) *ptrs[i] += off;
}

#define BWD_PROLOGUE(NWT_)                                                                         \
  bwd_replica(a);                                                                                  \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];                         \
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;           \
  BwdLds<PREC, NW, NWT_> lds(smem_raw, w);                                                         \
  Coop<PREC, NW> coop(lds.coop, w, lane, a.ablate);                                                        \
  constexpr int NOWN = Coop<PREC, NW>::NOWN;                                                       \
  const int ntiles = (a.T + 15) / 16;                                                              \
  const int tstride = gridDim.x * NW;                                                              \
  const int nrounds = (ntiles + tstride - 1) / tstride;   /* identical for every wave: barriers inside */ \
  int tile = blockIdx.x * NW + w;                                                                   \
  const WPack wpk{a.wp_base, reinterpret_cast<const __bf16*>(a.wp_img)};

// FFN reverse shared by encoder and decoder:  y = mask(R + drop2(conv2 relu(drop1(conv1 xin)))).
// Returns g = masked upstream gradient and dxin = gradient reaching xin through the FFN.
template <int PREC, int NW>
ADT_DEVICE_INLINE void ffn_bwd_tile(const BwdChainArgs& a, float* scr, Coop<PREC, NW>& coop,
                                    const typename WImg<PREC>::T* W2t, const typename WImg<PREC>::T* W1t,
                                    f32x4 (&dW2)[Coop<PREC, NW>::NOWN], f32x4 (&dW1)[Coop<PREC, NW>::NOWN], VAcc& db2, VAcc& db1,
                                    const CT& xin, const RowRegs& gy_rows, const RowRegs& u_rows, uint32_t key1, uint32_t key2,
                                    int row0, int lane, int c, int g, CT& gout, CT& dxin) {
  gout = rows_to_ct(scr, gy_rows, lane, c, g);
  ct_mask_rows(gout, a.ids, row0, a.T, g);
  CT df = gout;
  ct_dropmask(df, key2, a.drop, (uint32_t)row0 + a.row_offset, c, g);
  const CT u = rows_to_ct(scr, u_rows, lane, c, g);
  coop.product(dW2, df, u);
  colsum_accum(db2, df);
  CT dt = gemm_w<PREC>(ct_to_a<PREC>(scr, df, c, g), W2t, c, g);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (!(u.v[nt][r] > 0.f)) dt.v[nt][r] = 0.f;
  ct_dropmask(dt, key1, a.drop, (uint32_t)row0 + a.row_offset, c, g);
  coop.product(dW1, dt, xin);
  colsum_accum(db1, dt);
  dxin = gemm_w<PREC>(ct_to_a<PREC>(scr, dt, c, g), W1t, c, g);
}

// ---- encoder: y = mask(h2 + FFN(h2)), h2 = LN2(h), h = Qn + o Wo^T + bo ------------------------------------
// W0 = conv2, W1 = conv1, W2 = out_proj ; out0 = dh (gradient wrt h == wrt Qn residual), out1 = dO
template <int PREC, int NW, int HC>   // HC: compile-time cap on the number of heads of the fused classifier (0 = none)
__global__ __launch_bounds__(NW * 64) void k_enc_post_bwd(BwdChainArgs a) {
  constexpr int HCM = HC > 0 ? HC : 1;
  BWD_PROLOGUE(3)
  {
    typename WImg<PREC>::T* const im[3] = {lds.w[0], lds.w[1], lds.w[2]};
    const float* const wsrc[3] = {a.W0, a.W1, a.W2};
    stage_w_set<PREC, NW * 64, 3>(im, wsrc, true, wpk);
  }
  __syncthreads();
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  f32x4 dW2[NOWN], dW1[NOWN], dWo[NOWN];
  VAcc db2, db1, dbo, dgm, dbt;
  acc_zero(dW2); acc_zero(dW1); acc_zero(dWo);
  vacc_zero(db2); vacc_zero(db1); vacc_zero(dbo); vacc_zero(dgm); vacc_zero(dbt);
  // head classifier: column 16nt+c of this lane is element jcol of head hcol[nt]
  const bool cls = HC > 0 && a.drec != nullptr;
  const int hd = 64 / (a.H > 0 ? a.H : 1);
  int hcol[4];
  float wcls[4][HCM];
  VAcc dws[HCM];          // dws[cc].v[nt]: partial of dWs[cc][jcol(nt)] over this lane's rows
  float dbs_acc[HCM];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    hcol[nt] = (16 * nt + c) / hd;
    const int j = 16 * nt + c - hcol[nt] * hd;
#pragma unroll
    for (int cc = 0; cc < HCM; ++cc) wcls[nt][cc] = (cls && cc < a.H) ? a.Ws[cc * hd + j] : 0.f;
  }
#pragma unroll
  for (int cc = 0; cc < HCM; ++cc) { vacc_zero(dws[cc]); dbs_acc[cc] = 0.f; }
  for (int rnd = 0; rnd < nrounds; ++rnd, tile += tstride) {
    const int row0 = tile * 16;    // may lie beyond T: a phantom tile of zeros that only takes part in the barriers
    // every input of the round is requested at its top; carrying the next round's gy / h rows across the round cost 340 B of
    // scratch per lane (the kernel sits at the 256-register budget of two waves per SIMD)
    const RowRegs gy_cur = rows_load(a.gy, 64, row0, a.T, lane);
    const RowRegs h_rows = rows_load_saved(a.xin, row0, a.T, lane, a.saved_bf16);
    const RowRegs u_cur = rows_load_saved(a.u, row0, a.T, lane, a.saved_bf16);    // requested now, consumed later in the round
    // classifier log-probs and their upstream gradient for the 16 tile rows (H*H floats per row, reference row order
    // l*B + b): requested now as one element per lane and step, parked in LDS when needed
    float cls_rec[4], cls_drec[4];
    const int HH = a.H * a.H;
    if (cls) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = q * 64 + lane, rr = e / HH, row = row0 + rr;
        cls_rec[q] = 0.f; cls_drec[q] = 0.f;
        if (e < 16 * HH && row < a.T) {
          const int b = row / a.L, l = row - b * a.L;
          const size_t off = (size_t)(l * a.B + b) * HH + (e - rr * HH);
          cls_rec[q] = a.rec[off];
          cls_drec[q] = a.drec[off];
        }
      }
    }
    const CT h = rows_to_ct(lds.scr, h_rows, lane, c, g);
    LnStat st;
    const CT xhat = ln_xhat(h, a.ln_eps, st);
    const CT h2 = ln_apply(xhat, a.gamma, a.beta, c);
    CT gm, dh2;
    ffn_bwd_tile<PREC, NW>(a, lds.scr, coop, lds.w[0], lds.w[1], dW2, dW1, db2, db1, h2, gy_cur, u_cur, key1, key2, row0, lane, c, g,
                           gm, dh2);
    const RowRegs o_rows = rows_load_saved(a.o, row0, a.T, lane, a.saved_bf16);   // requested after the FFN reverse (register budget), before the store below
    ct_add(dh2, gm);
    const CT dh = ln_bwd_ct(dh2, xhat, st, a.gamma, dgm, dbt, c, g);
    store_ct(lds.scr, a.out0, 64, dh, row0, a.T, lane, c, g);
    const CT o = rows_to_ct(lds.scr, o_rows, lane, c, g);
    coop.product(dWo, dh, o);
    colsum_accum(dbo, dh);
    CT dO = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, dh, c, g), lds.w[2], c, g);
    if (cls) {
      // dz = drec - softmax(z) * sum(drec), softmax = exp(rec); dO += dz Ws; dWs += dz^T o_head; dbs += dz
      wave_fence();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = q * 64 + lane;
        if (e < 16 * HH) { lds.cls[e] = cls_rec[q]; lds.cls[256 + e] = cls_drec[q]; }
      }
      wave_fence();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = 4 * g + r;
        if (row0 + rl < a.T) {
#pragma unroll 1
          for (int h = 0; h < a.H; ++h) {
            float dz[HCM];
            float sd = 0.f;
#pragma unroll
            for (int cc = 0; cc < HCM; ++cc) {
              dz[cc] = (cc < a.H) ? lds.cls[256 + rl * HH + h * a.H + cc] : 0.f;
              sd += dz[cc];
            }
#pragma unroll
            for (int cc = 0; cc < HCM; ++cc)
              if (cc < a.H) {
                dz[cc] -= __expf(lds.cls[rl * HH + h * a.H + cc]) * sd;
                if (c == 0) dbs_acc[cc] += dz[cc];
              }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              if (hcol[nt] == h) {
                float add = 0.f;
#pragma unroll
                for (int cc = 0; cc < HCM; ++cc)
                  if (cc < a.H) {
                    add += dz[cc] * wcls[nt][cc];
                    dws[cc].v[nt] += dz[cc] * o.v[nt][r];
                  }
                dO.v[nt][r] += add;
              }
          }
        }
      }
    }
    store_ct(lds.scr, a.out1, 64, dO, row0, a.T, lane, c, g);
  }
  if (cls) {
    // classifier weight gradient: H column-sum vectors folded modulo hd; bias: one value per (wave, g) through LDS
    float* red = reinterpret_cast<float*>(lds.coop);
#pragma unroll
    for (int cc = 0; cc < HCM; ++cc) {
      if (cc < a.H) {
        const VAcc* const accs[1] = {&dws[cc]};
        float* const dst[1] = {a.dWs + cc * hd};
        vacc_flush_wg<NW, 1>(accs, dst, red, w, c, g, hd);
      }
    }
    __syncthreads();
    if (c == 0) {
#pragma unroll
      for (int cc = 0; cc < HCM; ++cc) red[(w * 4 + g) * HCM + cc] = dbs_acc[cc];
    }
    __syncthreads();
    if ((int)threadIdx.x < a.H) {
      float sb = 0.f;
      for (int i = 0; i < NW * 4; ++i) sb += red[i * HCM + threadIdx.x];
      atomicAdd(a.dbs + threadIdx.x, sb);
    }
  }
  coop.flush(dW2, a.dW0, c, g); coop.flush(dW1, a.dW1, c, g); coop.flush(dWo, a.dW2, c, g);
  {
    const VAcc* const accs[5] = {&db2, &db1, &dbo, &dgm, &dbt};
    float* const dst[5] = {a.db0, a.db1, a.db2, a.dgamma, a.dbeta};
    vacc_flush_wg<NW, 5>(accs, dst, reinterpret_cast<float*>(lds.coop), w, c, g);
  }
}

// ---- decoder: y = mask(Dn + a2 + FFN(a2)), a2 = o2 Wo2^T + b --------------------------------------------------
// W0 = conv2, W1 = conv1, W2 = enc_attn.out_proj ; out0 = dO2
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_post_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(3)
  {
    typename WImg<PREC>::T* const im[3] = {lds.w[0], lds.w[1], lds.w[2]};
    const float* const wsrc[3] = {a.W0, a.W1, a.W2};
    stage_w_set<PREC, NW * 64, 3>(im, wsrc, true, wpk);
  }
  __syncthreads();
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  f32x4 dW2[NOWN], dW1[NOWN], dWo[NOWN];
  VAcc db2, db1, dbo;
  acc_zero(dW2); acc_zero(dW1); acc_zero(dWo);
  vacc_zero(db2); vacc_zero(db1); vacc_zero(dbo);
  for (int rnd = 0; rnd < nrounds; ++rnd, tile += tstride) {
    const int row0 = tile * 16;
    // requested at the top of the round (no carry across rounds: register budget, see k_enc_post_bwd)
    const RowRegs gy_cur = rows_load(a.gy, 64, row0, a.T, lane);
    const RowRegs u_cur = rows_load_saved(a.u, row0, a.T, lane, a.saved_bf16);
    const RowRegs a2_rows = rows_load_saved(a.xin, row0, a.T, lane, a.saved_bf16);
    const RowRegs o_rows = rows_load_saved(a.o, row0, a.T, lane, a.saved_bf16);
    const CT a2 = rows_to_ct(lds.scr, a2_rows, lane, c, g);
    const CT o = rows_to_ct(lds.scr, o_rows, lane, c, g);
    CT gm, da2;
    ffn_bwd_tile<PREC, NW>(a, lds.scr, coop, lds.w[0], lds.w[1], dW2, dW1, db2, db1, a2, gy_cur, u_cur, key1, key2, row0, lane, c, g,
                           gm, da2);
    ct_add(da2, gm);
    coop.product(dWo, da2, o);
    colsum_accum(dbo, da2);
    const CT dO = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, da2, c, g), lds.w[2], c, g);
    store_ct(lds.scr, a.out0, 64, dO, row0, a.T, lane, c, g);
  }
  coop.flush(dW2, a.dW0, c, g); coop.flush(dW1, a.dW1, c, g); coop.flush(dWo, a.dW2, c, g);
  {
    const VAcc* const accs[3] = {&db2, &db1, &dbo};
    float* const dst[3] = {a.db0, a.db1, a.db2};
    vacc_flush_wg<NW, 3>(accs, dst, reinterpret_cast<float*>(lds.coop), w, c, g);
  }
}

// ---- LayerNorm + packed in-projection reverse (encoder: q from LN(x), k/v from x; decoder: all from LN(x)) ----
// W0,W1,W2 = Wq,Wk,Wv ; dqkv = packed gradient (T x 192) ; ENC: dh = residual-path gradient wrt LN output
// out0 = gradient wrt x (acc0: add to what is there) ; DEC: + mask(gy) added to the LN-output gradient
template <int PREC, int NW, bool ENC>
__global__ __launch_bounds__(NW * 64) void k_pre_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(3)
  {
    typename WImg<PREC>::T* const im[3] = {lds.w[0], lds.w[1], lds.w[2]};
    const float* const wsrc[3] = {a.W0, a.W1, a.W2};
    stage_w_set<PREC, NW * 64, 3>(im, wsrc, true, wpk);
  }
  __syncthreads();
  f32x4 dWq[NOWN], dWk[NOWN], dWv[NOWN];
  VAcc dbq, dbk, dbv, dgm, dbt;
  acc_zero(dWq); acc_zero(dWk); acc_zero(dWv);
  vacc_zero(dbq); vacc_zero(dbk); vacc_zero(dbv); vacc_zero(dgm); vacc_zero(dbt);
  const float* rsrc = ENC ? a.dh : a.gy;
  for (int rnd = 0; rnd < nrounds; ++rnd, tile += tstride) {
    const int row0 = tile * 16;
    // everything is requested at the top of the round: carrying two row sets across rounds cost 224 B of scratch per lane
    RowRegs dq_rows = rows_load(a.dqkv, a.lddqkv, row0, a.T, lane);
    RowRegs x_rows = rows_load(a.xin, 64, row0, a.T, lane);
    const RowRegs dk_rows = rows_load(a.dqkv + 64, a.lddqkv, row0, a.T, lane);
    const RowRegs dv_rows = rows_load(a.dqkv + 128, a.lddqkv, row0, a.T, lane);
    const RowRegs r_rows = rows_load(rsrc, 64, row0, a.T, lane);
    const CT x = rows_to_ct(lds.scr, x_rows, lane, c, g);
    LnStat st;
    const CT xhat = ln_xhat(x, a.ln_eps, st);
    const CT xn = ln_apply(xhat, a.gamma, a.beta, c);
    CT dn;
    {
      const CT dq = rows_to_ct(lds.scr, dq_rows, lane, c, g);
      const AFrags<PREC> aq = scr_to_a<PREC>(lds.scr, c, g);
      coop.product(dWq, dq, xn);
      colsum_accum(dbq, dq);
      dn = gemm_w<PREC>(aq, lds.w[0], c, g);          // gradient wrt the LN output
    }
    CT dkv;
    {
      const CT dk = rows_to_ct(lds.scr, dk_rows, lane, c, g);
      const AFrags<PREC> ak = scr_to_a<PREC>(lds.scr, c, g);
      coop.product(dWk, dk, ENC ? x : xn);
      colsum_accum(dbk, dk);
      dkv = gemm_w<PREC>(ak, lds.w[1], c, g);
    }
    {
      const CT dv = rows_to_ct(lds.scr, dv_rows, lane, c, g);
      const AFrags<PREC> av = scr_to_a<PREC>(lds.scr, c, g);
      coop.product(dWv, dv, ENC ? x : xn);
      colsum_accum(dbv, dv);
      ct_add(dkv, gemm_w<PREC>(av, lds.w[2], c, g));
    }
    RowRegs acc_rows;
    if (a.acc0) acc_rows = rows_load(a.out0, 64, row0, a.T, lane);
    CT res = rows_to_ct(lds.scr, r_rows, lane, c, g);
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
    if (a.acc0) ct_add(dx, rows_to_ct(lds.scr, acc_rows, lane, c, g));
    store_ct(lds.scr, a.out0, 64, dx, row0, a.T, lane, c, g);
  }
  coop.flush(dWq, a.dW0, c, g); coop.flush(dWk, a.dW1, c, g); coop.flush(dWv, a.dW2, c, g);
  {
    const VAcc* const accs[5] = {&dbq, &dbk, &dbv, &dgm, &dbt};
    float* const dst[5] = {a.db0, a.db1, a.db2, a.dgamma, a.dbeta};
    vacc_flush_wg<NW, 5>(accs, dst, reinterpret_cast<float*>(lds.coop), w, c, g);
  }
}

// ---- decoder middle, part 1: q2 = a1 Wq^T + b, a1 = o1 Wo1^T + b --------------------------------------------------
// W0 = Wq2, W1 = Wo1 (slf out_proj) ; dqkv = dq2 (ld 64) ; xin = a1, o = o1 ; out0 = dO1
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_dec_mid_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(2)
  {
    typename WImg<PREC>::T* const im[2] = {lds.w[0], lds.w[1]};
    const float* const wsrc[2] = {a.W0, a.W1};
    stage_w_set<PREC, NW * 64, 2>(im, wsrc, true, wpk);
  }
  __syncthreads();
  f32x4 dWq[NOWN], dWo[NOWN];
  VAcc dbq, dbo;
  acc_zero(dWq); acc_zero(dWo); vacc_zero(dbq); vacc_zero(dbo);
  RowRegs dq_rows = rows_load(a.dqkv, a.lddqkv, tile * 16, a.T, lane);
  RowRegs a1_rows = rows_load_saved(a.xin, tile * 16, a.T, lane, a.saved_bf16);
  RowRegs o1_rows = rows_load_saved(a.o, tile * 16, a.T, lane, a.saved_bf16);
  for (int rnd = 0; rnd < nrounds; ++rnd, tile += tstride) {
    const int row0 = tile * 16;
    const CT a1 = rows_to_ct(lds.scr, a1_rows, lane, c, g);
    const CT dq = rows_to_ct(lds.scr, dq_rows, lane, c, g);
    const AFrags<PREC> aq = scr_to_a<PREC>(lds.scr, c, g);
    const CT o1 = rows_to_ct(lds.scr, o1_rows, lane, c, g);
    const int nrow0 = (tile + tstride) * 16;
    dq_rows = rows_load(a.dqkv, a.lddqkv, nrow0, a.T, lane);
    a1_rows = rows_load_saved(a.xin, nrow0, a.T, lane, a.saved_bf16);
    o1_rows = rows_load_saved(a.o, nrow0, a.T, lane, a.saved_bf16);
    coop.product(dWq, dq, a1);
    colsum_accum(dbq, dq);
    const CT da1 = gemm_w<PREC>(aq, lds.w[0], c, g);
    coop.product(dWo, da1, o1);
    colsum_accum(dbo, da1);
    const CT dO1 = gemm_w<PREC>(ct_to_a<PREC>(lds.scr, da1, c, g), lds.w[1], c, g);
    store_ct(lds.scr, a.out0, 64, dO1, row0, a.T, lane, c, g);
  }
  coop.flush(dWq, a.dW0, c, g); coop.flush(dWo, a.dW1, c, g);
  {
    const VAcc* const accs[2] = {&dbq, &dbo};
    float* const dst[2] = {a.db0, a.db1};
    vacc_flush_wg<NW, 2>(accs, dst, reinterpret_cast<float*>(lds.coop), w, c, g);
  }
}

// ---- decoder middle, part 2: [k2, v2] = f Wkv^T + b --------------------------------------------------------------
// W0 = Wk2, W1 = Wv2 ; dkv2 (T x 128) ; f = log_feats ; out0 = g_f (+=)
template <int PREC, int NW>
__global__ __launch_bounds__(NW * 64) void k_kv_bwd(BwdChainArgs a) {
  BWD_PROLOGUE(2)
  {
    typename WImg<PREC>::T* const im[2] = {lds.w[0], lds.w[1]};
    const float* const wsrc[2] = {a.W0, a.W1};
    stage_w_set<PREC, NW * 64, 2>(im, wsrc, true, wpk);
  }
  __syncthreads();
  f32x4 dWk[NOWN], dWv[NOWN];
  VAcc dbk, dbv;
  acc_zero(dWk); acc_zero(dWv); vacc_zero(dbk); vacc_zero(dbv);
  RowRegs dk_rows = rows_load(a.dkv2, 128, tile * 16, a.T, lane);
  RowRegs dv_rows = rows_load(a.dkv2 + 64, 128, tile * 16, a.T, lane);
  RowRegs f_rows = rows_load(a.f, 64, tile * 16, a.T, lane);
  RowRegs acc_rows = rows_load(a.out0, 64, tile * 16, a.T, lane);
  for (int rnd = 0; rnd < nrounds; ++rnd, tile += tstride) {
    const int row0 = tile * 16;
    const CT f = rows_to_ct(lds.scr, f_rows, lane, c, g);
    const CT dk = rows_to_ct(lds.scr, dk_rows, lane, c, g);
    const AFrags<PREC> ak = scr_to_a<PREC>(lds.scr, c, g);
    const CT dv = rows_to_ct(lds.scr, dv_rows, lane, c, g);
    const AFrags<PREC> av = scr_to_a<PREC>(lds.scr, c, g);
    CT df = rows_to_ct(lds.scr, acc_rows, lane, c, g);
    const int nrow0 = (tile + tstride) * 16;
    dk_rows = rows_load(a.dkv2, 128, nrow0, a.T, lane);
    dv_rows = rows_load(a.dkv2 + 64, 128, nrow0, a.T, lane);
    f_rows = rows_load(a.f, 64, nrow0, a.T, lane);
    acc_rows = rows_load(a.out0, 64, nrow0, a.T, lane);
    coop.product(dWk, dk, f);
    colsum_accum(dbk, dk);
    coop.product(dWv, dv, f);
    colsum_accum(dbv, dv);
    ct_add(df, gemm_w<PREC>(ak, lds.w[0], c, g));
    ct_add(df, gemm_w<PREC>(av, lds.w[1], c, g));
    store_ct(lds.scr, a.out0, 64, df, row0, a.T, lane, c, g);
  }
  coop.flush(dWk, a.dW0, c, g); coop.flush(dWv, a.dW1, c, g);
  {
    const VAcc* const accs[2] = {&dbk, &dbv};
    float* const dst[2] = {a.db0, a.db1};
    vacc_flush_wg<NW, 2>(accs, dst, reinterpret_cast<float*>(lds.coop), w, c, g);
  }
}

}  // namespace adt
