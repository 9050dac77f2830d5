// Per-sequence fused layer forward on the register-resident transposed chains of adt_tt.cuh (bf16 MFMA operands).
// Same contract and argument block as adt_seqfwd.cuh (one workgroup = one sequence = one EncoderLayer / DecoderLayer forward,
// sasrec/modules.py:644-655, :666-677; the same tensors reach HBM), different data path: activations never touch LDS -- the only
// LDS traffic is the read-only weight images and the key / value images of the attention (both row-major [token][64], bf16).
#pragma once
#include "adt_seq_args.h"
#include "adt_tt.cuh"

namespace adt {

#define TQ_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && lane == 0) a.stamps[w * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

constexpr int TQ_NW = 8;                       // waves per workgroup of the backward kernels that share tq_tile
constexpr int TQ_FWD_NW = 12;                  // forward kernels: 12 waves = 3 per SIMD (<= 168 VGPRs): the 13 tiles of L = 200 are one per wave but one (tq_tile12)
constexpr int TQ_MAXKT = 14;                   // 16-token tiles per sequence: L <= 224
constexpr int TQ_LP = TQ_MAXKT * 16;

template <int NWT>
struct SeqTtLds {
  static constexpr size_t wbytes = (size_t)NWT * TT_WIMG * 2, ibytes = (size_t)TQ_LP * TT_RS * 2;
  static constexpr size_t vbytes = 1152 * sizeof(float);            // per-feature vectors (SeqVec offsets)
  static constexpr size_t bytes = wbytes + 2 * ibytes + vbytes;
  __bf16* w[NWT]; __bf16* sK; __bf16* sV; float* vec;
  __device__ SeqTtLds(unsigned char* base) {
    __bf16* pw = reinterpret_cast<__bf16*>(base);
    for (int i = 0; i < NWT; ++i) w[i] = pw + i * TT_WIMG;
    sK = reinterpret_cast<__bf16*>(base + wbytes);
    sV = reinterpret_cast<__bf16*>(base + wbytes + ibytes);
    vec = reinterpret_cast<float*>(base + wbytes + 2 * ibytes);
  }
};

// offsets (floats) of the per-feature vectors in the LDS vector area
enum SeqVec { SV_GAMMA = 0, SV_BETA = 64, SV_BIN = 128, SV_BO = 320, SV_GAMMA2 = 384, SV_BETA2 = 448, SV_B1 = 512, SV_B2 = 576, SV_BIN2 = 640,
              SV_BO2 = 832, SV_WS = 896, SV_BS = 960, SV_LNL_G = 1024, SV_LNL_B = 1088 };
constexpr int SV_FLOATS = 1152;

// all vectors in one pass: element i of the 1,152-float area comes from segment i >> 6.  Split into the global loads (issued together
// with the weight-image loads at kernel entry) and the LDS stores (after the zero-fill), so the prologue pays ONE global round trip.
struct TqVecRegs { float v[2]; };
template <int NTHREADS>
ADT_DEVICE_INLINE TqVecRegs tq_vec_load(const SeqFwdArgs& a, int H, int HD) {
  TqVecRegs r;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * NTHREADS;
    const int seg = i >> 6, j = i & 63;
    const float* src = nullptr;
    switch (seg) {
      case 0: src = a.gamma; break;
      case 1: src = a.beta; break;
      case 2: case 3: case 4: src = a.bin ? a.bin + 64 * (seg - 2) : nullptr; break;
      case 5: src = a.bo; break;
      case 6: src = a.gamma2; break;
      case 7: src = a.beta2; break;
      case 8: src = a.b1; break;
      case 9: src = a.b2; break;
      case 10: case 11: case 12: src = a.bin2 ? a.bin2 + 64 * (seg - 10) : nullptr; break;
      case 13: src = a.bo2; break;
      case 14: src = (a.rec && j < H * HD) ? a.Ws : nullptr; break;
      case 15: src = (a.rec && j < H) ? a.bs : nullptr; break;
      case 16: src = a.lnl_gamma; break;
      case 17: src = a.lnl_gamma ? a.lnl_beta : nullptr; break;
    }
    r.v[it] = (i < SV_FLOATS && src) ? src[j] : 0.f;
  }
  return r;
}
template <int NTHREADS>
ADT_DEVICE_INLINE void tq_vec_store(float* vec, const TqVecRegs& r) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * NTHREADS;
    if (i < SV_FLOATS) vec[i] = r.v[it];
  }
}

// the layer output, optionally scaled and accumulated (supernet mixing epilogue)
ADT_DEVICE_INLINE void tq_store_y(const SeqFwdArgs& a, size_t row, TT y, bool valid, int g) {
  float* dst = a.y + row * 64;
  if (a.y_scale != 0.f) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) y.v[nt] *= a.y_scale;
    if (a.y_acc) tt_add(y, tt_load(dst, valid, g));
  }
  tt_store(dst, y, valid, g);
}

// Tile of wave w in slot s.  Slot 0: the heaviest causal tiles, n-1-w.  Slot 1 in SNAKE order, tile n-16+w (waves 16-n .. 7): waves w and
// w + 4 share a SIMD, and with the straight order (tile w for w < n-8) SIMD 0 carried causal weight 13+9+1+5 = 28 of 91 at n = 13
// against 20 for SIMD 3 -- its waves finished the attention phases 5k cycles after the others (profiles/r02_stamps_fwd.txt);
// the snake gives 24 / 23 / 22 / 22.
ADT_DEVICE_INLINE int tq_tile(int s, int w, int ntiles, int nw = TQ_NW) {
  if (s == 0) return ntiles - 1 - w;          // < 0: no tile for this wave in this slot
  const int t = ntiles - 2 * nw + w;
  return t >= 0 ? t : -1;
}
// Forward kernels, 12 waves: waves w, w + 4 and w + 8 share a SIMD, and the attention phase is bound by each SIMD's vector-instruction
// issue (profiles/r03_stamps_fwd12.txt: with tile 12 - w on wave w, SIMD 0 carried causal weight 13 + 9 + 5 = 27 of 91 against 19 on
// SIMD 3 and its waves left the phase 7k cycles after the others).  The tiles, heaviest first (j = 0 is tile ntiles - 1), are dealt to the
// four SIMD classes in snake order 0 1 2 3 3 2 1 0 0 1 2 3 3 ..: wave w takes round k = w / 4 in slot 0 and waves 8 .. 11 take round 3 in
// slot 1 -- 24 / 23 / 22 / 22 at 13 tiles.
// At 13 tiles (L = 200) one SIMD carries FOUR tiles, and most of a tile's cost does not depend on its position: in-projection, out_proj and
// feed-forward are ~11k cycles per tile against ~1.1k per causal key tile in the attention (profiles/r03_stamps_fwd12.txt).  The snake gave
// that SIMD tiles 9, 8, 1, 0 -- 66k of work against 55-57k on the others, and its wave with two tiles ended the kernel 5k cycles after
// everyone else.  Fitted to the stamps (a tile ~7.2k + 0.81k per causal key tile) the table deals 12 9 1 | 11 8 2 | 10 7 5 | 3 6 4 + 0:
// 25 / 24 / 25 causal units on the three-tile SIMDs, 17 on the four-tile one.  The two tiles of one wave
// are a serial chain, so they go to wave 3, the OLDEST wave of its SIMD (the issue arbiter prefers older waves: as the youngest, wave 11
// finished its two tiles last even with the lightest load).
ADT_DEVICE_INLINE int tq_tile12(int s, int w, int ntiles) {
  if (ntiles == 13) {
    const int t13 = (0xC << 0) | (0xB << 4) | (0xA << 8) | (0x3 << 12) | (0x9 << 16) | (0x8 << 20) | (0x7 << 24) | (0x6 << 28);      // waves 0 .. 7
    const int t13b = (0x1 << 0) | (0x2 << 4) | (0x5 << 8) | (0x4 << 12);                                                              // waves 8 .. 11
    if (s == 0) return w < 8 ? (int)(((unsigned)t13 >> (4 * w)) & 15u) : ((t13b >> (4 * (w - 8))) & 15);
    return w == 3 ? 0 : -1;
  }
  const int cl = w & 3, k = s == 0 ? (w >> 2) : ((w >> 2) == 2 ? 3 : 4);
  const int j = 4 * k + ((k & 1) ? 3 - cl : cl);
  return (k < 4 && j < ntiles) ? ntiles - 1 - j : -1;
}

// The wave with two tiles used to run both in-projections in front of the barrier behind which every wave's attention starts: the other
// eleven waited ~3k cycles for its second one (profiles/r03_stamps_fwd12.txt: pre1 17.0k against 11.5-14k).  At 13 tiles the k / v rows of
// that second tile (tile 0, which every query attends to) are computed by wave 0 instead -- an oldest wave, done with its own tile first -- in
// its idle slot 1; the owner computes the tile's query operands behind the barrier.
constexpr int TQ_HELPER_WAVE = 0;
ADT_DEVICE_INLINE int tq_helper_tile(int w, int ntiles) { return (ntiles == 13 && w == TQ_HELPER_WAVE) ? 0 : -1; }
ADT_DEVICE_INLINE bool tq_split_second(int ntiles) { return ntiles == 13; }

// slot-ordered packed image of the 64 x 64 block at W (adt_seq.hip: k_pack_wimg writes it at + 2 images, transposed at + 3)
template <int NTHREADS>
ADT_DEVICE_INLINE void tq_stage(__bf16* img, const float* W, bool transposed, const SeqFwdArgs& a) {
  const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.wp_img) + 6 * (W - a.wp_base) + (transposed ? 3 : 2) * WPACK_IMG);
  uint4* dst = reinterpret_cast<uint4*>(img);
  for (int i = threadIdx.x; i < TT_WIMG * 2 / 16; i += NTHREADS) dst[i] = src[i];
}

ADT_DEVICE_INLINE const uint4* tq_img_src(const SeqFwdArgs& a, const float* W, bool transposed) {
  return reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.wp_img) + 6 * (W - a.wp_base) + (transposed ? 3 : 2) * WPACK_IMG);
}
// one weight image per statement, in named registers: the array-of-registers form (tq_img_load) was kept in scratch by the compiler in
// these two kernels, with a wait for the loads right behind their issue
// (the tail load is unconditional -- lanes without a tail chunk re-read their first chunk: a load inside `if (tq_tail)` drew a vmcnt(0) behind it)
// (TQW = the kernel's wave count; an image is TQ_CH 16-byte chunks: with more threads than chunks the surplus threads re-read chunk 0 and store nothing)
constexpr int TQ_CH = TT_WIMG * 2 / 16;
#define TQ_IMG_LOAD(k, Wk) const uint4* tqp##k = tq_img_src(a, Wk, false); const uint4 tqr##k = tqp##k[tq_body ? threadIdx.x : 0]; \
  const uint4 tqt##k = tqp##k[tq_tail ? threadIdx.x + TQW * 64 : 0];
#define TQ_IMG_STORE(k, dstk) if (tq_body) reinterpret_cast<uint4*>(dstk)[threadIdx.x] = tqr##k; if (tq_tail) reinterpret_cast<uint4*>(dstk)[threadIdx.x + TQW * 64] = tqt##k;

template <int NTHREADS>
ADT_DEVICE_INLINE void tq_zero(__bf16* p, size_t nbytes) {
  uint4* q = reinterpret_cast<uint4*>(p);
  for (int i = threadIdx.x; i < (int)(nbytes / 16); i += NTHREADS) q[i] = make_uint4(0u, 0u, 0u, 0u);
}

// layer input of this lane's token: a load, or the embedding gather x = dropout(E[id] * sqrt(d) + P[l]) * (id != 0)   (model.py:34-41).
// In two steps, so that the prologue can put the requests of BOTH of a wave's tiles in flight beside the weight images (they used to be
// issued inside the tile loop behind the first barrier: one exposed round trip per tile, two for the gather, whose row address needs the id):
// tq_x_request issues the loads (lanes without a token read the zero row, tt_load), tq_x_finish forms x where the tile is processed.
struct TqX { TT e; TT p; int id; };
ADT_DEVICE_INLINE TqX tq_x_request(const SeqFwdArgs& a, int row, int l, bool valid, int id, int g) {
  TqX r;
  r.id = id;
  if (a.x) {
    r.e = tt_load(a.x + (size_t)row * 64, valid, g);
    r.p = tt_zero();
  } else {
    r.e = tt_load(a.E + (size_t)id * 64, valid, g);        // id == 0 (padding, or no token): row 0 of the table, discarded below
    r.p = tt_load(a.P + (size_t)l * 64, valid, g);
  }
  return r;
}
ADT_DEVICE_INLINE TT tq_x_finish(const SeqFwdArgs& a, const TqX& r, int row, bool valid, uint32_t key0, int g) {
  if (a.x) return r.e;
  TT x = tt_zero();
  if (r.id != 0) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) x.v[nt][e] = r.e.v[nt][e] * a.emb_scale + r.p.v[nt][e];
    tt_dropout(x, key0, a.drop, (uint32_t)row + a.b_offset * (uint32_t)a.L, g);
  }
  if (a.x_out) tt_store(a.x_out + (size_t)row * 64, x, valid, g);
  return x;
}

// LayerNorm + packed in-projection of one tile: query operands to registers, k / v into the LDS images (and to HBM for the backward)
// PART 0: everything.  The tile of the one wave that carries TWO tiles (13 tiles on 12 waves) is split: PART 1 = the k / v rows only, by a
// helper wave in front of the barrier every wave's attention waits at; PART 2 = x / LN(x) stores and the query operands, by the owner behind
// that barrier (tq_helper_tile).  Both recompute x and LN(x) from the same inputs (same hash dropout): identical values.
template <int HD, bool ENC, int PART = 0>
ADT_DEVICE_INLINE void tq_pre_tile(const SeqFwdArgs& a, const float* vec, const __bf16* wq, const __bf16* wk, const __bf16* wv, __bf16* sK, __bf16* sV,
                                   int tile, int b, uint32_t key0, float qmul, int c, int g,
                                   bf16x8 (&fq)[(64 / HD) * ((HD + 31) / 32)], TT& xn_out, const TqX& xr) {
  const int l = tile * 16 + c, row = b * a.L + l;
  const bool valid = l < a.L;
  TT x;
  if (PART == 1) {
    SeqFwdArgs a1 = a;
    a1.x_out = nullptr;                            // the owner stores x
    x = tq_x_finish(a1, xr, row, valid, key0, g);
  } else {
    x = tq_x_finish(a, xr, row, valid, key0, g);
  }
  const TT xn = tt_layernorm(x, vec + SV_GAMMA, vec + SV_BETA, a.ln_eps, g);
  if (PART != 1 && a.xn) tt_store(a.xn + (size_t)row * 64, xn, valid, g);
  if (PART != 1) xn_out = xn;
  const TTB bn = tt_bfrags(xn);
  TTB bx;
  if (ENC) bx = tt_bfrags(x);
  if (PART != 1) {
    TT q = tt_gemm(bn, wq, c, g);
    tt_add_vec(q, vec + SV_BIN, g);
    if (a.qkv && valid) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<float4*>(a.qkv + (size_t)row * 192 + 16 * nt + 4 * g) = make_float4(q.v[nt][0], q.v[nt][1], q.v[nt][2], q.v[nt][3]);
    }
    tt_qfrags<HD>(q, qmul, fq);
  }
  if (PART == 2) return;
  {
    TT k = tt_gemm(ENC ? bx : bn, wk, c, g);
    tt_add_vec(k, vec + SV_BIN + 64, g);
    if (a.qkv && valid) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<float4*>(a.qkv + (size_t)row * 192 + 64 + 16 * nt + 4 * g) = make_float4(k.v[nt][0], k.v[nt][1], k.v[nt][2], k.v[nt][3]);
    }
    tt_put_slot(sK, l, k, valid, g);
  }
  {
    TT v = tt_gemm(ENC ? bx : bn, wv, c, g);
    tt_add_vec(v, vec + SV_BIN + 128, g);
    if (a.qkv && valid) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<float4*>(a.qkv + (size_t)row * 192 + 128 + 16 * nt + 4 * g) = make_float4(v.v[nt][0], v.v[nt][1], v.v[nt][2], v.v[nt][3]);
    }
    tt_put_rows(sV, l, v, valid, g);
  }
}

// u = relu(dropout1(xin W1^T + b1)) -> stored ; returns dropout2(u W2^T + b2)           (PointWiseFeedForward, modules.py:629-633)
ADT_DEVICE_INLINE TT tq_ffn(const SeqFwdArgs& a, const float* vec, const __bf16* w1, const __bf16* w2, const TT& xin, uint32_t key1, uint32_t key2, int row,
                            bool valid, int c, int g) {
  const uint32_t rg = (uint32_t)row + a.b_offset * (uint32_t)a.L;
  TT u = tt_gemm(tt_bfrags(xin), w1, c, g);
  tt_add_vec(u, vec + SV_B1, g);
  tt_dropout(u, key1, a.drop, rg, g);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) u.v[nt][r] = fmaxf(u.v[nt][r], 0.f);
  tt_save(a.u, row, u, valid, g, a.saved_bf16);
  TT y = tt_gemm(tt_bfrags(u), w2, c, g);
  tt_add_vec(y, vec + SV_B2, g);
  tt_dropout(y, key2, a.drop, rg, g);
  return y;
}

// ---- encoder layer: weight images 0 Wq, 1 Wk, 2 Wv, 3 out_proj, 4 conv1, 5 conv2 ------------------------------------------------
template <int HD>
__global__ __launch_bounds__(TQ_FWD_NW * 64) void k_seqtt_enc_fwd(SeqFwdArgs a) {
  constexpr int TQW = TQ_FWD_NW;
  adt_prefetch_kernargs<(sizeof(SeqFwdArgs) + 63) / 64 * 64 <= 512 ? sizeof(SeqFwdArgs) : 512>();      // adt_common.cuh
  constexpr int H = 64 / HD, KB = (HD + 31) / 32, NF = H * KB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  SeqTtLds<6> lds(smem_raw);
  const int nsp = a.nsplit > 1 ? a.nsplit : 1, b = blockIdx.x / nsp, part = blockIdx.x % nsp;      // adt_seq_args.h: several workgroups per sequence
  const int L = a.L, ntiles = (L + 15) / 16;
  const bool own[2] = {tq_tile12(0, w, ntiles) >= 0 && tq_tile12(0, w, ntiles) % nsp == part, tq_tile12(1, w, ntiles) >= 0 && tq_tile12(1, w, ntiles) % nsp == part};
  TQ_STAMP(0);
  uint32_t seedv = 0u;
  TqX xr[2];
  const int htile = tq_helper_tile(w, ntiles);      // >= 0: this wave computes the k / v rows of that tile in its slot 1
  const bool split2 = tq_split_second(ntiles);      // the owner of a second tile leaves its k / v rows to the helper
  int idm[2] = {0, 0};                        // ids of this wave's OWN tiles (the pad mask of the layer output: read at the end of a tile it was an exposed round trip)
  {
    int idr[2] = {0, 0};                      // the ids first: vector-memory results return in issue order, and the gather's row loads wait for them
    if (!a.x) {                               // embedding layer: the gather's row addresses need the ids
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int tile = tq_tile12(s, w, ntiles), l = tile * 16 + c;
        idm[s] = tt_load_id(a.ids, b * L + l, tile >= 0 && l < L);
        idr[s] = idm[s];
      }
      if (htile >= 0) idr[1] = tt_load_id(a.ids, b * L + htile * 16 + c, htile * 16 + c < L);
    }      // the helper's slot 1 is another wave's tile
    const bool tq_body = TQW * 64 <= TQ_CH || threadIdx.x < TQ_CH;
    const bool tq_tail = (int)threadIdx.x < TQ_CH - TQW * 64;        // chunks beyond the first TQW * 64 (wave-uniform: whole waves)
    // only what the in-projection needs is requested in front of the first barrier (the prologue is a bandwidth burst: every workgroup of
    // the launch pulls its inputs at once, ~11 B per cycle and CU); out_proj / conv1 / conv2 follow behind it by LDS-DMA
    TQ_IMG_LOAD(0, a.Win) TQ_IMG_LOAD(1, a.Win + 4096) TQ_IMG_LOAD(2, a.Win + 8192)
    const TqVecRegs vr = tq_vec_load<TQW * 64>(a, H, HD);
    if (a.drop.thr) seedv = *a.drop.seed;     // behind the image requests: read first, its round trip preceded every other load of the kernel
#pragma unroll
    for (int s = 0; s < 2; ++s) {             // the layer input of both tiles, in flight while the images are stored
      const int tile = (s == 1 && htile >= 0) ? htile : tq_tile12(s, w, ntiles), l = tile * 16 + c;
      xr[s] = tq_x_request(a, b * L + l, l, tile >= 0 && l < L, idr[s], g);
    }
    if (a.x) {                                // later layers: the ids are only the output's pad mask -- behind every other request
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int tile = tq_tile12(s, w, ntiles), l = tile * 16 + c;
        idm[s] = tt_load_id(a.ids, b * L + l, tile >= 0 && l < L);
      }
    }
    TQ_STAMP(11);
    tq_zero<TQW * 64>(lds.sK, 2 * SeqTtLds<6>::ibytes);
    TQ_STAMP(12);
    TQ_IMG_STORE(0, lds.w[0]) TQ_IMG_STORE(1, lds.w[1]) TQ_IMG_STORE(2, lds.w[2])
    TQ_STAMP(13);
    tq_vec_store<TQW * 64>(lds.vec, vr);
    TQ_STAMP(14);
  }
  __syncthreads();
  TQ_STAMP(1);
  adt_glds_block<TQW>(tq_img_src(a, a.Wo, false), lds.w[3], TT_WIMG * 2);
  adt_glds_block<TQW>(tq_img_src(a, a.W1, false), lds.w[4], TT_WIMG * 2);
  adt_glds_block<TQW>(tq_img_src(a, a.W2, false), lds.w[5], TT_WIMG * 2);
  const uint32_t key0 = adt_site_key(seedv, a.site_emb), key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  const float qmul = a.scale * 1.4426950408889634f;          // scores in log2 units: the softmax is exp2
  bf16x8 fq[2][NF];
  TT xn[2];
  {
    const int tile = tq_tile12(0, w, ntiles);
    if (tile >= 0 && own[0]) tq_pre_tile<HD, true>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile, b, key0, qmul, c, g, fq[0], xn[0], xr[0]);
    else if (tile >= 0) tq_pre_tile<HD, true, 1>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile, b, key0, qmul, c, g, fq[0], xn[0], xr[0]);      // another workgroup's tile: k / v rows only
    TQ_STAMP(2);
    const int tile1 = tq_tile12(1, w, ntiles);
    if (htile >= 0) tq_pre_tile<HD, true, 1>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, htile, b, key0, qmul, c, g, fq[1], xn[1], xr[1]);
    else if (tile1 >= 0 && !split2 && own[1]) tq_pre_tile<HD, true>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile1, b, key0, qmul, c, g, fq[1], xn[1], xr[1]);
    else if (tile1 >= 0 && !split2) tq_pre_tile<HD, true, 1>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile1, b, key0, qmul, c, g, fq[1], xn[1], xr[1]);
    TQ_STAMP(3);
  }
  adt_wait_vm0();            // the three images requested behind the first barrier have landed: published by this one
  __syncthreads();
  TQ_STAMP(4);
  {      // the owner's part of a split second tile, first (its layer-input registers die here, as they did in front of the barrier)
    const int tile1 = tq_tile12(1, w, ntiles);
    if (split2 && tile1 >= 0 && own[1]) tq_pre_tile<HD, true, 2>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile1, b, key0, qmul, c, g, fq[1], xn[1], xr[1]);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = tq_tile12(s, w, ntiles);
    if (tile < 0 || !own[s]) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    const TT o = tt_attn_heads<HD, TQ_MAXKT>(lds.sK, lds.sV, fq[s], tile, L, b, a.b_offset, a.drop, a.site_attn, seedv, a.lse, a.mask, lane, c, g);
    TQ_STAMP(5 + 3 * s);
    tt_save(a.o, row, o, valid, g, a.saved_bf16);
    if (a.rec) {
      // head classifier (sasrec/modules.py:648-649): z[h][cc] = sum_j o[h hd + j] Ws[cc][j] + bs[cc]; log-softmax over cc.  The features of
      // a token are spread over this lane's 16 registers and the 4 lanes g: in-lane products, two cross-lane steps per class.
      constexpr int NT = HD / 16;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float z[H];
#pragma unroll
        for (int cc = 0; cc < H; ++cc) {
          float acc = 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float4 ws = *reinterpret_cast<const float4*>(lds.vec + SV_WS + cc * HD + 16 * nt + 4 * g);
            const f32x4& ov = o.v[h * NT + nt];
            acc += ov[0] * ws.x + ov[1] * ws.y + ov[2] * ws.z + ov[3] * ws.w;
          }
          z[cc] = tt_colsum(acc) + lds.vec[SV_BS + cc];
        }
        float m = z[0];
#pragma unroll
        for (int cc = 1; cc < H; ++cc) m = fmaxf(m, z[cc]);
        float se = 0.f;
#pragma unroll
        for (int cc = 0; cc < H; ++cc) se += __expf(z[cc] - m);
        const float lz = m + __logf(se);
        if (g == 0 && valid) {
          float* dst = a.rec + ((size_t)(l * a.B + b) * H + h) * H;
#pragma unroll
          for (int cc = 0; cc < H; ++cc) dst[cc] = z[cc] - lz;
        }
      }
    }
    TT hh = tt_gemm(tt_bfrags(o), lds.w[3], c, g);
    tt_add_vec(hh, lds.vec + SV_BO, g);
    tt_add(hh, xn[s]);                                        // the residual adds LN1(x)                         (modules.py:651)
    TQ_STAMP(6 + 3 * s);
    tt_save(a.h, row, hh, valid, g, a.saved_bf16);
    const TT h2 = tt_layernorm(hh, lds.vec + SV_GAMMA2, lds.vec + SV_BETA2, a.ln_eps, g);
    TT y = tq_ffn(a, lds.vec, lds.w[4], lds.w[5], h2, key1, key2, row, valid, c, g);
    tt_add(y, h2);
    if (!valid || idm[s] == 0) y = tt_zero();
    tq_store_y(a, row, y, valid, g);
    if (a.f_out) tt_store(a.f_out + (size_t)row * 64, tt_layernorm(y, lds.vec + SV_LNL_G, lds.vec + SV_LNL_B, a.ln_eps, g), valid, g);   // log_feats (model.py:48)
    TQ_STAMP(7 + 3 * s);
  }
}

// ---- decoder layer: set A: 0 Wq, 1 Wk, 2 Wv (slf_attn), 3 slf out_proj, 4 enc_attn Wq ; set B: 0 enc_attn Wk, 1 Wv, 2 enc_attn out_proj,
// 3 conv1, 4 conv2 -------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(TQ_FWD_NW * 64) void k_seqtt_dec_fwd(SeqFwdArgs a) {
  constexpr int TQW = TQ_FWD_NW;
  adt_prefetch_kernargs<(sizeof(SeqFwdArgs) + 63) / 64 * 64 <= 512 ? sizeof(SeqFwdArgs) : 512>();      // adt_common.cuh
  constexpr int H = 64 / HD, KB = (HD + 31) / 32, NF = H * KB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  SeqTtLds<5> lds(smem_raw);
  const int nsp = a.nsplit > 1 ? a.nsplit : 1, b = blockIdx.x / nsp, part = blockIdx.x % nsp;      // adt_seq_args.h: several workgroups per sequence
  const int L = a.L, ntiles = (L + 15) / 16;
  const bool own[2] = {tq_tile12(0, w, ntiles) >= 0 && tq_tile12(0, w, ntiles) % nsp == part, tq_tile12(1, w, ntiles) >= 0 && tq_tile12(1, w, ntiles) % nsp == part};
  uint32_t seedv = 0u;
  const bool tq_body = TQW * 64 <= TQ_CH || threadIdx.x < TQ_CH;
  const bool tq_tail = (int)threadIdx.x < TQ_CH - TQW * 64;        // chunks beyond the first TQW * 64 (wave-uniform: whole waves)
  TqX xr[2];
  const int htile = tq_helper_tile(w, ntiles);      // see k_seqtt_enc_fwd: k / v rows (both attentions) of the two-tile wave's second tile
  const bool split2 = tq_split_second(ntiles);
  int idm[2] = {0, 0};                        // see k_seqtt_enc_fwd
  {
    int idr[2] = {0, 0};                      // ids first, layer inputs of both tiles behind the images (see k_seqtt_enc_fwd)
    if (!a.x) {                               // embedding layer: the gather's row addresses need the ids
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int tile = tq_tile12(s, w, ntiles), l = tile * 16 + c;
        idm[s] = tt_load_id(a.ids, b * L + l, tile >= 0 && l < L);
        idr[s] = idm[s];
      }
      if (htile >= 0) idr[1] = tt_load_id(a.ids, b * L + htile * 16 + c, htile * 16 + c < L);
    }
    TQ_IMG_LOAD(0, a.Win) TQ_IMG_LOAD(1, a.Win + 4096) TQ_IMG_LOAD(2, a.Win + 8192)      // out_proj and Wq2 follow behind the barrier (see k_seqtt_enc_fwd)
    const TqVecRegs vr = tq_vec_load<TQW * 64>(a, H, HD);
    if (a.drop.thr) seedv = *a.drop.seed;     // behind the image requests (see k_seqtt_enc_fwd)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int tile = (s == 1 && htile >= 0) ? htile : tq_tile12(s, w, ntiles), l = tile * 16 + c;
      xr[s] = tq_x_request(a, b * L + l, l, tile >= 0 && l < L, idr[s], g);
    }
    if (a.x) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int tile = tq_tile12(s, w, ntiles), l = tile * 16 + c;
        idm[s] = tt_load_id(a.ids, b * L + l, tile >= 0 && l < L);
      }
    }
    tq_zero<TQW * 64>(lds.sK, 2 * SeqTtLds<5>::ibytes);
    TQ_IMG_STORE(0, lds.w[0]) TQ_IMG_STORE(1, lds.w[1]) TQ_IMG_STORE(2, lds.w[2])
    tq_vec_store<TQW * 64>(lds.vec, vr);
  }
  __syncthreads();
  adt_glds_block<TQW>(tq_img_src(a, a.Wo, false), lds.w[3], TT_WIMG * 2);
  adt_glds_block<TQW>(tq_img_src(a, a.Win2, false), lds.w[4], TT_WIMG * 2);
  const uint32_t key0 = adt_site_key(seedv, a.site_emb), key1 = adt_site_key(seedv, a.site1), key2 = adt_site_key(seedv, a.site2);
  const float qmul = a.scale * 1.4426950408889634f;
  bf16x8 fq[2][NF];
  TT dn[2];
  // self attention: D = LN(x); q, k, v = D Win^T + b                                                    (sasrec/modules.py:668-670)
  {
    const int tile = tq_tile12(0, w, ntiles), tile1 = tq_tile12(1, w, ntiles);
    if (tile >= 0 && own[0]) tq_pre_tile<HD, false>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile, b, key0, qmul, c, g, fq[0], dn[0], xr[0]);
    else if (tile >= 0) tq_pre_tile<HD, false, 1>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile, b, key0, qmul, c, g, fq[0], dn[0], xr[0]);      // another workgroup's tile: k / v rows only
    if (htile >= 0) tq_pre_tile<HD, false, 1>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, htile, b, key0, qmul, c, g, fq[1], dn[1], xr[1]);
    else if (tile1 >= 0 && !split2 && own[1]) tq_pre_tile<HD, false>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile1, b, key0, qmul, c, g, fq[1], dn[1], xr[1]);
    else if (tile1 >= 0 && !split2) tq_pre_tile<HD, false, 1>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile1, b, key0, qmul, c, g, fq[1], dn[1], xr[1]);
  }
  adt_wait_vm0();
  __syncthreads();
  {
    const int tile1 = tq_tile12(1, w, ntiles);
    if (split2 && tile1 >= 0 && own[1]) tq_pre_tile<HD, false, 2>(a, lds.vec, lds.w[0], lds.w[1], lds.w[2], lds.sK, lds.sV, tile1, b, key0, qmul, c, g, fq[1], dn[1], xr[1]);
  }
  // a1 = out_proj(o1) ; q2 = a1 Wq2^T + b : the cross attention's queries replace the self attention's in the registers
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = tq_tile12(s, w, ntiles);
    if (tile < 0 || !own[s]) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    const TT o1 = tt_attn_heads<HD, TQ_MAXKT>(lds.sK, lds.sV, fq[s], tile, L, b, a.b_offset, a.drop, a.site_attn, seedv, a.lse, a.mask, lane, c, g);
    tt_save(a.o, row, o1, valid, g, a.saved_bf16);
    TT a1 = tt_gemm(tt_bfrags(o1), lds.w[3], c, g);
    tt_add_vec(a1, lds.vec + SV_BO, g);
    tt_save(a.a1, row, a1, valid, g, a.saved_bf16);
    TT q2 = tt_gemm(tt_bfrags(a1), lds.w[4], c, g);
    tt_add_vec(q2, lds.vec + SV_BIN2, g);
    tt_save(a.q2, row, q2, valid, g, a.saved_bf16);
    tt_qfrags<HD>(q2, qmul, fq[s]);
  }
  // weight set B is requested before the barrier ...
  TQ_IMG_LOAD(5, a.Win2 + 4096) TQ_IMG_LOAD(6, a.Win2 + 8192) TQ_IMG_LOAD(7, a.Wo2) TQ_IMG_LOAD(8, a.W1) TQ_IMG_LOAD(9, a.W2)
  TT fa[2];                             // the encoder's log_feats rows of both tiles, behind the images (inside the tile loop each was an exposed round trip)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = (s == 1 && split2) ? htile : tq_tile12(s, w, ntiles), l = tile * 16 + c;      // the helper projects the split tile's k2 / v2 as well
    fa[s] = tt_load(a.f + (size_t)(b * L + l) * 64, tile >= 0 && l < L, g);
  }
  __syncthreads();                      // every wave is done with the self-attention images and with weight set A
  TQ_IMG_STORE(5, lds.w[0]) TQ_IMG_STORE(6, lds.w[1]) TQ_IMG_STORE(7, lds.w[2]) TQ_IMG_STORE(8, lds.w[3]) TQ_IMG_STORE(9, lds.w[4])      // ... and lands after it
  __syncthreads();
  // cross attention keys / values from the encoder's log_feats: [k2, v2] = f Wkv^T + b              (memory = log_feats, model.py:69-70)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = (s == 1 && split2) ? htile : tq_tile12(s, w, ntiles);
    if (tile < 0) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    const TTB bf = tt_bfrags(fa[s]);
    TT k2 = tt_gemm(bf, lds.w[0], c, g);
    tt_add_vec(k2, lds.vec + SV_BIN2 + 64, g);
    TT v2 = tt_gemm(bf, lds.w[1], c, g);
    tt_add_vec(v2, lds.vec + SV_BIN2 + 128, g);
    const bool mine = tile % nsp == part;      // every workgroup of a split sequence computes these rows; one of them stores them
    if (!mine) {
    } else if (a.kv2 && a.saved_bf16) {          // rows of 128 bf16: k2 | v2
      __bf16* kvrow = reinterpret_cast<__bf16*>(a.kv2) + (size_t)row * 128;
      tt_store_bf16(kvrow, k2, valid, g);
      tt_store_bf16(kvrow + 64, v2, valid, g);
    } else if (a.kv2 && valid) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        *reinterpret_cast<float4*>(a.kv2 + (size_t)row * 128 + 16 * nt + 4 * g) = make_float4(k2.v[nt][0], k2.v[nt][1], k2.v[nt][2], k2.v[nt][3]);
        *reinterpret_cast<float4*>(a.kv2 + (size_t)row * 128 + 64 + 16 * nt + 4 * g) = make_float4(v2.v[nt][0], v2.v[nt][1], v2.v[nt][2], v2.v[nt][3]);
      }
    }
    tt_put_slot(lds.sK, l, k2, valid, g);
    tt_put_rows(lds.sV, l, v2, valid, g);
  }
  __syncthreads();
  // a2 = out_proj(o2) ; y = (D + a2 + FFN(a2)) * mask                                                (sasrec/modules.py:673-676)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tile = tq_tile12(s, w, ntiles);
    if (tile < 0 || !own[s]) continue;
    const int l = tile * 16 + c, row = b * L + l;
    const bool valid = l < L;
    const TT o2 = tt_attn_heads<HD, TQ_MAXKT>(lds.sK, lds.sV, fq[s], tile, L, b, a.b_offset, a.drop, a.site_attn2, seedv, a.lse2, a.mask2, lane, c, g);
    tt_save(a.o2, row, o2, valid, g, a.saved_bf16);
    TT a2 = tt_gemm(tt_bfrags(o2), lds.w[2], c, g);
    tt_add_vec(a2, lds.vec + SV_BO2, g);
    tt_save(a.h, row, a2, valid, g, a.saved_bf16);
    TT y = tq_ffn(a, lds.vec, lds.w[3], lds.w[4], a2, key1, key2, row, valid, c, g);
    tt_add(y, a2);
    tt_add(y, dn[s]);
    if (!valid || idm[s] == 0) y = tt_zero();
    tq_store_y(a, row, y, valid, g);
  }
}

}  // namespace adt
