// Per-sequence BACKWARD of the token-wise chains of a layer on the transposed tiles of adt_tt.cuh (bf16 MFMA operands):
//   k_seqtt_post_bwd<HD, true>   encoder: mask + PointWiseFeedForward + forward_layernorm + residual + out_proj (+ head classifier)
//                                (the autograd of sasrec/modules.py:648-654, :629-633)          -> dh (residual path), dO
//   k_seqtt_post_bwd<HD, false>  decoder: mask + PointWiseFeedForward + residual + enc_attn.out_proj (modules.py:673-676) -> dO2
//   k_seqtt_mid_bwd              decoder: enc_attn q-projection + slf_attn.out_proj -> dO1 ; enc_attn k / v projections -> d log_feats +=
// They replace k_enc_post_bwd / k_dec_post_bwd / k_dec_mid_bwd + k_kv_bwd of adt_bwdchain.cuh (same argument block, same outputs) when the
// batch is whole sequences.  One workgroup = one user sequence; the gradient of a token tile runs through the chain in registers (no LDS
// layout changes), and every weight gradient dW = G^T X is ONE product over all tokens of the sequence from two LDS row images
// (sb_dw_product of adt_seqbwd_tt.cuh; the ones column of the X image yields the bias gradient) instead of per-wave register
// accumulators: the row-major kernels held 3 x 64 accumulator registers per wave and sat at the 256-register limit.
#pragma once
#include "adt_bwdchain_args.h"
#include "adt_seqbwd_tt.cuh"

namespace adt {

template <int NWT>
struct SeqPostLds {
  static constexpr size_t wbytes = (size_t)NWT * TT_WIMG * 2, ibytes = (size_t)SB_R * TT_RS * 2;
  static constexpr size_t rbytes = (448 + 256) * 4;             // sRed: dgamma, dbeta, dWs, dbs, 3-4 bias vectors ; sVec: gamma, beta, Ws, last-LN gamma
  static constexpr int wvrow = 384;                             // per wave: dgamma, dbeta, dWs, dbs [0, 196) ; last LayerNorm's dgamma, dbeta [256, 384)
  static constexpr size_t wvbytes = 8 * wvrow * 4;              // per-wave sums (joined in wave order: no LDS atomics)
  static constexpr size_t bytes = wbytes + 2 * ibytes + 64 + rbytes + wvbytes;
};

ADT_DEVICE_INLINE void sp_replica(BwdChainArgs& a) {
  if (a.nrep <= 1) return;
  const size_t off = (size_t)(blockIdx.x % a.nrep) * a.rep_stride;
  float** const ptrs[] = {&a.dW0, &a.dW1, &a.dW2, &a.dW3, &a.db0, &a.db1, &a.db2, &a.db3, &a.dgamma, &a.dbeta, &a.dWs, &a.dbs, &a.lnl_dgamma, &a.lnl_dbeta};
#pragma unroll
  for (int i = 0; i < 14; ++i)
    if (*ptrs[i]) *ptrs[i] += off;
}

// transposed slot-ordered images (the operand of dX^T = W^T dY^T) of N weights, global -> LDS by LDS-DMA (no staging registers; the caller
// waits with adt_wait_vm0() in front of the barrier that publishes them)
template <int N, int NW>
ADT_DEVICE_INLINE void sp_wdma(const BwdChainArgs& a, const float* const (&W)[N], __bf16* wimg) {
#pragma unroll
  for (int j = 0; j < N; ++j)
    adt_glds_block<NW>(reinterpret_cast<const __bf16*>(a.wp_img) + 6 * (W[j] - a.wp_base) + 3 * WPACK_IMG, wimg + j * TT_WIMG, TT_WIMG * 2);
}
template <int NW>
ADT_DEVICE_INLINE void sp_zero_images(__bf16* img0) {
  constexpr size_t ibytes = (size_t)SB_R * TT_RS * 2;
  uint4* z = reinterpret_cast<uint4*>(img0);
  for (int i = threadIdx.x; i < (int)((2 * ibytes + 64) / 16); i += NW * 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
}
// Wave count of the token-chain backward kernels.  Every tile costs the same here (no causal weights), so with 13 tiles at L = 200 the eight-wave
// form is paced by the five waves that own two tiles; sixteen waves (<= 128 VGPRs, one tile slot per wave) give every tile its own wave and
// four waves per SIMD to cover each other's LDS and MFMA latency.
constexpr int SP_NW = 8;
constexpr int SP_NS = (14 + SP_NW - 1) / SP_NW;          // tile slots per wave (L <= 224)

template <int HD, bool ENC>
__global__ __launch_bounds__(SP_NW * 64) void k_seqtt_post_bwd(BwdChainArgs a) {
  adt_prefetch_kernargs<(sizeof(BwdChainArgs) + 63) / 64 * 64 <= 512 ? sizeof(BwdChainArgs) : 512>();      // adt_common.cuh
  constexpr int H = 64 / HD, NT = HD / 16, NW = SP_NW, NS = SP_NS;
  typedef SeqPostLds<3> Lds;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* wimg = reinterpret_cast<__bf16*>(smem_raw);                       // conv2^T, conv1^T, out_proj^T
  __bf16* img0 = reinterpret_cast<__bf16*>(smem_raw + Lds::wbytes);
  __bf16* img1 = reinterpret_cast<__bf16*>(smem_raw + Lds::wbytes + Lds::ibytes);
  float* sRed = reinterpret_cast<float*>(smem_raw + Lds::wbytes + 2 * Lds::ibytes + 64);   // [0,64) dgamma [64,128) dbeta [128,192) dWs [192,196) dbs [256,448) db0, db1, db2
  float* sVec = sRed + 448;                                                  // gamma, beta, Ws
  float* sWave = sVec + 256;                                                 // [8 waves][wvrow]: the sums of each wave
  constexpr int WVR = Lds::wvrow;
  const float *vgamma = sVec, *vbeta = sVec + 64, *vws = sVec + 128, *vgl = sVec + 192;
  // encoder, last block only: the upstream gradient is d log_feats and the model's last LayerNorm (sasrec/model.py:44, reversed) is undone
  // HERE, per tile, in front of everything else -- its own kernel was a 13.5 us streaming pass + a 6 us queue hop in the middle of the backward
  const bool lnl = ENC && a.lnl_x != nullptr;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nsp = a.nsplit > 1 ? a.nsplit : 1, b = blockIdx.x / nsp, part = blockIdx.x % nsp;
  const int L = a.L, ntiles = (L + 15) / 16, npair = (L + 31) / 32;
  auto tileof = [&](int s) { const int t = tq_tile(s, w, ntiles, NW); return (t >= 0 && t % nsp == part) ? t : -1; };      // this workgroup's tiles only
  const bool cls = ENC && H > 1 && a.drec != nullptr;
  sp_replica(a);
  SB_STAMP(0);
  // every workgroup of the launch runs the same phases at the same time: the activations of phase A are requested right behind the
  // weight images, so that HBM streams them while the prologue runs (requested at the top of phase A they arrived ~8k cycles after it)
  const float* const ws3[3] = {a.W0, a.W1, a.W2};
  sp_wdma<3, NW>(a, ws3, wimg);
  TT dy[NS], xl[NS];
  TTSaved uraw[NS], hreq[NS];                            // requested here, converted where they are first needed (tt_saved_value)
  int idv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = tile >= 0 && l < L;
    dy[s] = tt_load(a.gy + (size_t)row * 64, valid, g);
    if (ENC) xl[s] = tt_load(a.lnl_x + (size_t)row * 64, valid && lnl, g);
    uraw[s] = tt_saved_request(a.u, row, valid, g, a.saved_bf16);
    hreq[s] = tt_saved_request(a.xin, row, valid, g, a.saved_bf16);
    idv[s] = tt_load_id(a.ids, row, valid);
  }
  if (a.gy_scale != 0.f) {                                 // after every load of the prologue has been issued: a use is a wait
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dy[s].v[nt] *= a.gy_scale;
  }
  {
    sp_zero_images<NW>(img0);
    if (threadIdx.x < 448) sRed[threadIdx.x] = 0.f;
    if (ENC) { tt_stage_vec<NW * 64>(sVec, a.gamma, 64); tt_stage_vec<NW * 64>(sVec + 64, a.beta, 64); }
    if (cls) tt_stage_vec<NW * 64>(sVec + 128, a.Ws, 64);
    if (lnl) tt_stage_vec<NW * 64>(sVec + 192, a.lnl_gamma, 64);
    adt_wait_vm0();
  }
  const uint32_t seedv = a.drop.thr ? *a.drop.seed : 0u;
  const uint32_t key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  SB_STAMP(1);
  __syncthreads();
  SB_STAMP(2);
  // ---- A: masked upstream gradient through dropout2 and conv2 ; dW(conv2) = df^T u ----------------------------------------------------
  TT dt[NS];
  if (ENC) {
    if (lnl) {      // (workgroup-uniform) every row of the tile, padded positions too: the mask below is the layer's, not the LayerNorm's
      TT dgl = tt_zero(), dbl = tt_zero();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (tileof(s) < 0) continue;
        dy[s] = tt_ln_bwd(dy[s], tt_ln_stats(xl[s], a.lnl_eps), vgl, dgl, dbl, g);
      }
      float* mine = sWave + w * WVR + 256;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sg = tt_rowsum16(dgl.v[nt][r]), sb = tt_rowsum16(dbl.v[nt][r]);
          if (c == 0) { mine[16 * nt + 4 * g + r] = sg; mine[64 + 16 * nt + 4 * g + r] = sb; }
        }
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    const TT u = tt_saved_value(uraw[s], a.saved_bf16);
    if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (s == 0) SB_STAMP(3); }
    if (idv[s] == 0) dy[s] = tt_zero();
    TT gyv = dy[s];
    tt_dropout(gyv, key2, a.drop, (uint32_t)row + a.row_offset, g);         // the forward's keep decisions and scale, applied to the gradient
    tt_put_rows(img0, l, gyv, valid, g);
    tt_put_rows(img1, l, u, valid, g);
    TT t = tt_gemm(tt_bfrags(gyv), wimg, c, g);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!(u.v[nt][r] > 0.f)) t.v[nt][r] = 0.f;
    tt_dropout(t, key1, a.drop, (uint32_t)row + a.row_offset, g);
    dt[s] = t;
    if (s == 0) SB_STAMP(4);
  }
  SB_STAMP(5);
  __syncthreads();
  SB_STAMP(6);
  sb_dw_product16<NW>(img0, img1, npair, a.dW0, a.part[0] ? a.part[0] + (size_t)blockIdx.x * a.part_stride : nullptr, sRed + 256, w, c, g);
  SB_STAMP(7);
  __syncthreads();
  SB_STAMP(8);
  // ---- B: conv1 ; encoder: forward_layernorm backward -> dh ; dW(conv1) = dt^T LN2(h) (decoder: dt^T a2) --------------------------------
  TT dh[NS];
  TTSaved oreq[NS];
  f32x4 drq[NS], rcq[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) drq[s] = rcq[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  TT dgm = tt_zero(), dbt = tt_zero();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    oreq[s] = tt_saved_request(a.o, row, valid, g, a.saved_bf16);           // consumed in C
    if constexpr (H == 2) {                                                  // the head classifier's H x H values of this token, consumed in C
      if (cls) {                                                             // (read there they were two exposed round trips per tile)
        typedef const f32x4 __attribute__((address_space(1))) * gf4;
        const size_t off = (size_t)(l * a.B + b) * (H * H);
        drq[s] = *(valid ? (gf4)(a.drec + off) : (gf4)tt_zero_row);
        rcq[s] = *(valid ? (gf4)(a.rec + off) : (gf4)tt_zero_row);
      }
    }
    const TT hraw_s = tt_saved_value(hreq[s], a.saved_bf16);
    tt_put_rows(img0, l, dt[s], valid, g);
    TT d = tt_gemm(tt_bfrags(dt[s]), wimg + TT_WIMG, c, g);
    tt_add(d, dy[s]);                                                       // the residual around the feed-forward
    if (ENC) {
      const TTLn st = tt_ln_stats(hraw_s, a.ln_eps);
      tt_put_rows(img1, l, tt_ln_apply(st.xhat, vgamma, vbeta, g), valid, g);
      dh[s] = tt_ln_bwd(d, st, vgamma, dgm, dbt, g);
      tt_store(a.out0 + (size_t)row * 64, dh[s], valid, g);                 // gradient wrt h == wrt the LN1 output on the residual path
    } else {
      tt_put_rows(img1, l, hraw_s, valid, g);
      dh[s] = d;                                                            // gradient wrt a2 (the Dn residual is handled by the pre chain)
    }
  }
  SB_STAMP(9);
  __syncthreads();
  sb_dw_product16<NW>(img0, img1, npair, a.dW1, a.part[1] ? a.part[1] + (size_t)blockIdx.x * a.part_stride : nullptr, sRed + 320, w, c, g);
  SB_STAMP(10);
  __syncthreads();
  SB_STAMP(11);
  // ---- C: out_proj ; dW(out_proj) = dh^T o ; encoder: head classifier reverse joins dO ---------------------------------------------------
  float dws[H][NT][4], dbs_acc[H];
#pragma unroll
  for (int cc = 0; cc < H; ++cc) {
    dbs_acc[cc] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dws[cc][nt][r] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    const TT oraw_s = tt_saved_value(oreq[s], a.saved_bf16);
    tt_put_rows(img0, l, dh[s], valid, g);
    tt_put_rows(img1, l, oraw_s, valid, g);
    TT dO = tt_gemm(tt_bfrags(dh[s]), wimg + 2 * TT_WIMG, c, g);
    if (cls) {
      // z[h][cc] = o_h . Ws[cc] + bs[cc], rec = log_softmax_cc(z): dz = drec - exp(rec) * sum_cc drec ; dO_h += dz Ws ; dWs += dz^T o_h ; dbs += dz
      const size_t off = (size_t)(l * a.B + b) * (H * H);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float dz[H], rcv[H], sd = 0.f;
        if constexpr (H == 2) {
          const float dr4[4] = {drq[s][0], drq[s][1], drq[s][2], drq[s][3]}, rc4[4] = {rcq[s][0], rcq[s][1], rcq[s][2], rcq[s][3]};
#pragma unroll
          for (int cc = 0; cc < H; ++cc) { dz[cc] = valid ? dr4[h * H + cc] : 0.f; rcv[cc] = rc4[h * H + cc]; sd += dz[cc]; }
        } else {
#pragma unroll
          for (int cc = 0; cc < H; ++cc) { dz[cc] = valid ? a.drec[off + h * H + cc] : 0.f; rcv[cc] = valid ? a.rec[off + h * H + cc] : 0.f; sd += dz[cc]; }
        }
#pragma unroll
        for (int cc = 0; cc < H; ++cc) {
          dz[cc] -= valid ? __expf(rcv[cc]) * sd : 0.f;
          if (g == 0) dbs_acc[cc] += dz[cc];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int cc = 0; cc < H; ++cc) {
            const float4 wv = *reinterpret_cast<const float4*>(vws + cc * HD + 16 * nt + 4 * g);
            const f32x4 wq = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dO.v[h * NT + nt][r] += dz[cc] * wq[r];
              dws[cc][nt][r] += dz[cc] * oraw_s.v[h * NT + nt][r];
            }
          }
        }
      }
    }
    tt_store((ENC ? a.out1 : a.out0) + (size_t)row * 64, dO, valid, g);
  }
  SB_STAMP(12);
  __syncthreads();
  sb_dw_product16<NW>(img0, img1, npair, a.dW2, a.part[2] ? a.part[2] + (size_t)blockIdx.x * a.part_stride : nullptr, sRed + 384, w, c, g);
  SB_STAMP(13);
  if (ENC) {
    // per-lane partials over this wave's tokens -> this wave's sums in its own LDS row (16-lane DPP sums, plain stores) -> joined below in
    // wave order: the workgroup's sums do not depend on which wave arrives first (LDS float atomics added in arrival order)
    float* mine = sWave + w * WVR;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sg = tt_rowsum16(dgm.v[nt][r]), sb = tt_rowsum16(dbt.v[nt][r]);
        if (c == 0) { mine[16 * nt + 4 * g + r] = sg; mine[64 + 16 * nt + 4 * g + r] = sb; }
      }
    if (cls) {
#pragma unroll
      for (int cc = 0; cc < H; ++cc) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sw = tt_rowsum16(dws[cc][nt][r]);
            if (c == 0) mine[128 + cc * HD + 16 * nt + 4 * g + r] = sw;
          }
        const float sb = tt_rowsum16(dbs_acc[cc]);
        if (c == 0 && g == 0) mine[192 + cc] = sb;
      }
    }
  }
  __syncthreads();
  {
    const int t = threadIdx.x;
    if (ENC && t < 196) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) sum += sWave[k * WVR + t];
      sRed[t] = sum;
    }
    if (ENC) {
      if (lnl && t >= 256 && t < 384) {      // the last LayerNorm's dgamma | dbeta of this workgroup, waves in order
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) sum += sWave[k * WVR + t];
        if (a.vpart2) a.vpart2[(size_t)blockIdx.x * 512 + (t - 256)] = sum;
        else atomicAdd((t < 320 ? a.lnl_dgamma : a.lnl_dbeta - 64) + (t - 256), sum);
      }
    }
    // (each thread flushes the element it has just formed, or a bias sum of the weight-gradient sweep that the last barrier published)
    const bool live = t < 128 ? ENC : (t < 192 + H ? cls : (t >= 256 && t < 448));
    if (a.vpart) {
      if (t < 448) a.vpart[(size_t)blockIdx.x * 512 + t] = live ? sRed[t] : 0.f;
    } else if (live) {
      float* dst = t < 64 ? a.dgamma + t : t < 128 ? a.dbeta + t - 64 : t < 192 ? a.dWs + t - 128 : t < 256 ? a.dbs + t - 192
                 : t < 320 ? a.db0 + t - 256 : t < 384 ? a.db1 + t - 320 : a.db2 + t - 384;
      atomicAdd(dst, sRed[t]);
    }
  }
  SB_STAMP(14);
}

// W0 = enc_attn Wq, W1 = slf_attn.out_proj, W2 = enc_attn Wk, W3 = enc_attn Wv ; dqkv = dq2 (ld lddqkv), xin = a1, o = o1, dkv2 (B*L x 128),
// f = log_feats ; out0 = dO1, out1 = d log_feats (acc1: add to what is there)
constexpr int SP_MID_NW = 8;       // 12 waves (156 VGPRs, three per SIMD) measured 30.0 us against 28.3: with 13 tiles one wave still carries two, and its chain sets the time
__global__ __launch_bounds__(SP_MID_NW * 64) void k_seqtt_mid_bwd(BwdChainArgs a) {
  adt_prefetch_kernargs<(sizeof(BwdChainArgs) + 63) / 64 * 64 <= 512 ? sizeof(BwdChainArgs) : 512>();      // adt_common.cuh
  constexpr int NW = SP_MID_NW, NS = (14 + NW - 1) / NW;
  typedef SeqPostLds<4> Lds;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* wimg = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* img0 = reinterpret_cast<__bf16*>(smem_raw + Lds::wbytes);
  __bf16* img1 = reinterpret_cast<__bf16*>(smem_raw + Lds::wbytes + Lds::ibytes);
  float* sRed = reinterpret_cast<float*>(smem_raw + Lds::wbytes + 2 * Lds::ibytes + 64);   // the four bias gradients
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int nsp = a.nsplit > 1 ? a.nsplit : 1, b = blockIdx.x / nsp, part = blockIdx.x % nsp;
  const int L = a.L, ntiles = (L + 15) / 16, npair = (L + 31) / 32;
  auto tileof = [&](int s) { const int t = tq_tile(s, w, ntiles, NW); return (t >= 0 && t % nsp == part) ? t : -1; };      // this workgroup's tiles only
  sp_replica(a);
  const float* const ws4[4] = {a.W0, a.W1, a.W2, a.W3};
  sp_wdma<4, NW>(a, ws4, wimg);
  TT dqa[NS];
  TTSaved a1req[NS], oreq[NS];             // phase A's activations are requested right behind the weight images (see k_seqtt_post_bwd)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = tile >= 0 && l < L;
    dqa[s] = a.grad_bf16 ? tt_load_bf16(reinterpret_cast<const __bf16*>(a.dqkv) + (size_t)row * a.lddqkv, valid, g) : tt_load(a.dqkv + (size_t)row * a.lddqkv, valid, g);
    a1req[s] = tt_saved_request(a.xin, row, valid, g, a.saved_bf16);
    oreq[s] = tt_saved_request(a.o, row, valid, g, a.saved_bf16);
  }
  sp_zero_images<NW>(img0);
  if (threadIdx.x < 256) sRed[threadIdx.x] = 0.f;
  adt_wait_vm0();
  __syncthreads();
  // ---- A: cross-attention query projection: da1 = dq2 Wq ; dWq = dq2^T a1 ----------------------------------------------------------
  TT da1[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c;
    const bool valid = l < L;
    tt_put_rows(img0, l, dqa[s], valid, g);
    tt_put_rows(img1, l, tt_saved_value(a1req[s], a.saved_bf16), valid, g);
    da1[s] = tt_gemm(tt_bfrags(dqa[s]), wimg, c, g);
  }
  __syncthreads();
  sb_dw_product16<NW>(img0, img1, npair, a.dW0, a.part[0] ? a.part[0] + (size_t)blockIdx.x * a.part_stride : nullptr, sRed, w, c, g);
  __syncthreads();
  // ---- B: self-attention out_proj: dO1 = da1 Wo1 ; dWo1 = da1^T o1 --------------------------------------------------------------------
  TT dk[NS], fx[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    dk[s] = a.grad_bf16 ? tt_load_bf16(reinterpret_cast<const __bf16*>(a.dkv2) + (size_t)row * 128, valid, g) : tt_load(a.dkv2 + (size_t)row * 128, valid, g);
    fx[s] = tt_load(a.f + (size_t)row * 64, valid, g);
    tt_put_rows(img0, l, da1[s], valid, g);
    tt_put_rows(img1, l, tt_saved_value(oreq[s], a.saved_bf16), valid, g);
    tt_store(a.out0 + (size_t)row * 64, tt_gemm(tt_bfrags(da1[s]), wimg + TT_WIMG, c, g), valid, g);
  }
  __syncthreads();
  sb_dw_product16<NW>(img0, img1, npair, a.dW1, a.part[1] ? a.part[1] + (size_t)blockIdx.x * a.part_stride : nullptr, sRed + 64, w, c, g);
  __syncthreads();
  // ---- C: cross-attention keys: df = dk2 Wk ; dWk = dk2^T f -----------------------------------------------------------------------------
  TT df[NS], dv[NS], acc1v[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    dv[s] = a.grad_bf16 ? tt_load_bf16(reinterpret_cast<const __bf16*>(a.dkv2) + (size_t)row * 128 + 64, valid, g) : tt_load(a.dkv2 + (size_t)row * 128 + 64, valid, g);
    acc1v[s] = tt_load(a.out1 + (size_t)row * 64, valid && a.acc1, g);       // consumed in D (requested there it was an exposed round trip)
    tt_put_rows(img0, l, dk[s], valid, g);
    tt_put_rows(img1, l, fx[s], valid, g);
    df[s] = tt_gemm(tt_bfrags(dk[s]), wimg + 2 * TT_WIMG, c, g);
  }
  __syncthreads();
  sb_dw_product16<NW>(img0, img1, npair, a.dW2, a.part[2] ? a.part[2] + (size_t)blockIdx.x * a.part_stride : nullptr, sRed + 128, w, c, g);
  __syncthreads();
  // ---- D: cross-attention values (the X image still holds f) --------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tile = tileof(s);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    tt_put_rows(img0, l, dv[s], valid, g);
    tt_add(df[s], tt_gemm(tt_bfrags(dv[s]), wimg + 3 * TT_WIMG, c, g));
    float* dst = a.out1 + (size_t)row * 64;
    tt_add(df[s], acc1v[s]);                                                // zeros unless a.acc1
    tt_store(dst, df[s], valid, g);
  }
  __syncthreads();
  sb_dw_product16<NW>(img0, img1, npair, a.dW3, a.part[3] ? a.part[3] + (size_t)blockIdx.x * a.part_stride : nullptr, sRed + 192, w, c, g);
  __syncthreads();          // the last product's bias sums
  {
    const int t = threadIdx.x;
    float* const dst[4] = {a.db0, a.db1, a.db2, a.db3};
    if (t < 256) {
      if (a.vpart) a.vpart[(size_t)blockIdx.x * 512 + t] = sRed[t];
      else atomicAdd(dst[t >> 6] + (t & 63), sRed[t]);
    }
  }
}

}  // namespace adt
