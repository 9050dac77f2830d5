// Fused all-item logits + cross-entropy for BERT4Rec-ADT's masked rows (bert4rec/model/bert.py:80-90: logits = h E^T + bias over
// V + 100 items; bert4rec/trainer.py:113-115: CrossEntropyLoss(ignore_index=0)).  The (M, V) logits / probabilities are never
// written to memory: the forward keeps an online log-sum-exp, the backward recomputes each 32 x 32 score tile on the matrix cores
// and feeds exp(score - lse) straight back as the operand of the gradient product.
//
// One kernel template serves the three passes.  "X" is the operand a wave keeps in registers (32 or 64 rows of KD bf16), "Y" the
// operand that streams through LDS in 32-row tiles shared by the eight waves of the workgroup:
//   LCE_FWD  X = masked rows h (64 per wave), Y = item table E.  S^T = E h^T + bias; per-lane online (max, sum).
//   LCE_DH   X = h rows (32 per wave),        Y = E.              P = exp(S + bias - lse) / n;  dh^T += E^T P.
//   LCE_DE   X = item rows (32 per wave),     Y = h.              P as above (transposed roles); dE^T += h^T P;  dbias += rowsum P.
// S is computed with Y as the A operand (v_mfma_f32_32x32x16_bf16: Y row on the accumulator rows, X row on the lane), so the 16
// accumulator registers of a lane belong to ONE X row: softmax statistics are in-lane, and the converted accumulator is already the B
// operand of the second product, which contracts over the Y rows (cdna_hip_programming.md, "An accumulator tile as the next MFMA's
// operand"); its A operand, Y^T, comes from the same LDS image through ds_read_b64_tr_b16.  The "- onehot(label)" term of the
// gradient is rank-one per row and is applied by the small kernels (k_lce_combine, k_lce_reduce), not inside the tiles.
//
// Y tiles: a 32 x KD bf16 image in 8-row x 32-column subtiles of 512 B with the 16-byte chunks of a row XOR-ed by (row >> 2) & 3 --
// conflict-free for the ds_read_b128 row reads and for the transposed reads (one image serves both).  Tiles are fetched by LDS-DMA
// (global_load_lds_dwordx4: the swizzle goes on the per-lane SOURCE address, the LDS side is linear) into a three-slot ring, two
// tiles ahead, with a counted s_waitcnt vmcnt and one raw s_barrier per tile.  The LDS-DMA is issued from inline asm: hipcc would
// otherwise wait vmcnt(0) in front of every ds_read that follows a pending global_load_lds.
//
// Work split: an item = (X block of the workgroup, one of C ranges of Y tiles); C is derived on the device from the live row count
// so that the items fill the grid once.  FWD writes (max, sum) partials, DH / DE write their fp32 partial products in register order
// (fully coalesced 16-byte stores); k_lce_combine / k_lce_reduce fold them.
#pragma once
#include "adt_common.cuh"

namespace adt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short lce_s4 __attribute__((ext_vector_type(4)));

enum { LCE_FWD = 0, LCE_DH = 1, LCE_DE = 2 };
constexpr int LCE_NW = 8;                 // waves per workgroup (two per SIMD, 256 registers each)
constexpr int LCE_NTH = LCE_NW * 64;
constexpr int LCE_NBUF = 3;
constexpr int LCE_CMAX = 64;              // at most this many Y ranges per X block
constexpr int LCE_XR_FWD = LCE_NW * 64;   // X rows per workgroup
constexpr int LCE_XR_BWD = LCE_NW * 32;
constexpr float LCE_L2E = 1.4426950408889634f;

template <int KD> struct LceGeo {
  static_assert(KD == 128 || KD == 256, "contraction width 128 or 256");
  static constexpr int NCH = KD / 8;                 // 16-byte chunks per row
  static constexpr int NKS = KD / 16;                // k-steps of the score product
  static constexpr int NFT = KD / 32;                // 32-feature tiles of the gradient product
  static constexpr int SUB = NCH / 4 * 512;          // bytes of an 8-row group of the image
  static constexpr int IMG = 4 * SUB;                // 32 rows
  static constexpr int YV = LCE_NW * 256;            // one copy per wave of the tile's 32 row constants (filled by that wave's own LDS-DMA)
  static constexpr int BUF = IMG + YV;
  static constexpr int PIECES = IMG / 1024;          // 1 KiB LDS-DMA pieces per tile
  static constexpr int PPW = PIECES / LCE_NW;        // per wave
  static constexpr int LDS = LCE_NBUF * BUF;
  static constexpr int VM = PPW + 1;                 // vector-memory operations a wave issues per tile
};

struct LceArgs {
  const __bf16* Xb;        // [>= nx rounded up to the block][KD]; rows beyond nx are zero
  const __bf16* Yb;        // [>= ny rounded up to 32][KD]; rows beyond ny are zero
  const float* xv;         // per X row, natural-log units: DH ln(1/n) - lse[row], DE bias[item] (-inf beyond nx); FWD unused
  const float* yv;         // per Y row: FWD / DH bias[item], DE ln(1/n) - lse[row]; -inf beyond ny up to the next multiple of 32
  const int32_t* nx_dev;   // live counts in device memory (NULL: the host value)
  const int32_t* ny_dev;
  int nx, ny;
  float* part_m;           // FWD: per item, XR_FWD (max, sum) pairs
  float* part_s;
  float* dpart;            // DH / DE: per item, XR_BWD x KD floats in register order
  float* dsum;             // DE: dbias (atomic)
  int slots;               // workgroups the split aims at (the grid size)
};

struct LceSplit { int nxb, C, items; };
__host__ __device__ __forceinline__ LceSplit lce_split(int nx, int xr, int slots, int ntiles) {
  LceSplit s;
  s.nxb = (nx + xr - 1) / xr;
  int C = s.nxb > 0 ? slots / s.nxb : 0;
  C = C < 1 ? 1 : C;
  C = C > LCE_CMAX ? LCE_CMAX : C;
  C = C > ntiles ? ntiles : C;
  s.C = C;
  s.items = s.nxb * C;
  return s;
}

template <int KD>
ADT_DEVICE_INLINE int lce_off(int row, int ch) {
  return LceGeo<KD>::SUB * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// LDS-DMA from inline asm (see the header comment).  lds_dst: wave-uniform LDS byte address; the hardware adds lane * size.
ADT_DEVICE_INLINE void lce_glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
ADT_DEVICE_INLINE void lce_glds4(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> ADT_DEVICE_INLINE void lce_wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else static_assert(N < 0, "add the count");
}

ADT_DEVICE_INLINE bf16x8 lce_pack8(const f32x16& v, int base) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[base + j];
  return b;
}

template <int KD, int MODE>
__global__ __launch_bounds__(LCE_NTH) void k_lce(LceArgs a) {
  adt_prefetch_kernargs<sizeof(LceArgs) <= 512 ? sizeof(LceArgs) : 512>();      // every kernarg line in one scalar-cache round trip (adt_common.cuh)
  using G = LceGeo<KD>;
  constexpr int NX = MODE == LCE_FWD ? 2 : 1;
  constexpr int XR = LCE_NW * 32 * NX;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lce_smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
  const int nx = a.nx_dev ? min(*a.nx_dev, a.nx) : a.nx;
  const int ny = a.ny_dev ? min(*a.ny_dev, a.ny) : a.ny;
  const int ntiles = (ny + 31) / 32;
  const LceSplit sp = lce_split(nx, XR, a.slots, ntiles);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lce_smem;

  // LDS-DMA source offsets (elements, relative to the tile's first row) of this lane's chunks: piece pc = w * PPW + i covers the linear
  // 16-byte positions pc * 64 + lane of the image
  int srcoff[G::PPW];
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) {
    const int p = (w * G::PPW + i) * 64 + lane, grp = p / (G::SUB / 16), q = p % (G::SUB / 16);
    const int row = 8 * grp + ((q & 31) >> 2), ch = 4 * (q >> 5) + ((q & 3) ^ ((row >> 2) & 3));
    srcoff[i] = row * KD + ch * 8;
  }
  // row-read bases (A operand of the score product: row r, chunk 2 ks + h) and transposed-read bases (A operand of the gradient product)
  int rb[2], tb[2];
  {
    const int v = (r >> 2) & 3;
    rb[0] = G::SUB * (r >> 3) + 64 * (r & 7) + 16 * (h ^ v);
    rb[1] = G::SUB * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ v);
    const int li = lane & 15, q = li >> 2, p = li & 3, gi = (lane >> 4) & 1;
    tb[0] = 64 * (4 * h + q) + 16 * ((2 * gi + (p >> 1)) ^ h) + 8 * (p & 1);
    tb[1] = 64 * (4 * h + q) + 16 * ((2 * gi + (p >> 1)) ^ (2 + h)) + 8 * (p & 1);
  }

  const int per_xcd = gridDim.x >> 3;                     // workgroups b and b + 8 share an XCD: give an XCD consecutive items (same Y range)
  const int j0 = gridDim.x % 8 == 0 ? (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
  for (int j = j0; j < sp.items; j += gridDim.x) {
    const int c = j / sp.nxb, xb = j - c * sp.nxb;
    const int t0 = (int)((long long)c * ntiles / sp.C), t1 = (int)((long long)(c + 1) * ntiles / sp.C);
    const int x0 = xb * XR + w * 32 * NX;
    __builtin_amdgcn_s_barrier();                         // the ring of the previous item is no longer read

    auto issue = [&](int t, int slot) {
      const int tt = t < t1 ? t : t1 - 1;                 // the tail re-fetches the last tile: the per-tile operation count stays constant
      const uint32_t base = lds0 + slot * G::BUF;
      lce_glds4(a.yv + (size_t)tt * 32 + r, base + G::IMG + w * 256);
      const __bf16* src = a.Yb + (size_t)tt * 32 * KD;
#pragma unroll
      for (int i = 0; i < G::PPW; ++i) lce_glds16(src + srcoff[i], base + (w * G::PPW + i) * 1024);
    };
    issue(t0, 0);
    issue(t0 + 1, 1);

    // X operand: lane (r, h) holds X[x0 + r][16 ks + 8 h .. + 8]
    bf16x8 xf[NX][G::NKS];
#pragma unroll
    for (int n = 0; n < NX; ++n) {
      const bf16x8* xp = reinterpret_cast<const bf16x8*>(a.Xb + (size_t)(x0 + 32 * n + r) * KD + 8 * h);
#pragma unroll
      for (int ks = 0; ks < G::NKS; ++ks) xf[n][ks] = xp[2 * ks];
    }
    float xvl = 0.f;
    if constexpr (MODE != LCE_FWD) xvl = a.xv[x0 + r] * LCE_L2E;      // log2 units: added inside the exponent
    // a "use" of every register the loads above fill, in front of the tile loop: hipcc then waits for them HERE.  Left to itself it puts
    // its vmcnt(15) ... vmcnt(0) in front of their first use inside the loop, where they would drain the LDS-DMA ring on every tile.
#pragma unroll
    for (int n = 0; n < NX; ++n)
#pragma unroll
      for (int ks = 0; ks < G::NKS; ++ks) asm volatile("" : "+v"(xf[n][ks]));
    asm volatile("" : "+v"(xvl));
    float sm[NX], ss[NX];                                  // FWD: running max (natural units) and sum
#pragma unroll
    for (int n = 0; n < NX; ++n) { sm[n] = -INFINITY; ss[n] = 0.f; }
    f32x16 dacc[MODE == LCE_FWD ? 1 : G::NFT];
    if constexpr (MODE != LCE_FWD) {
#pragma unroll
      for (int ft = 0; ft < G::NFT; ++ft)
#pragma unroll
        for (int e = 0; e < 16; ++e) dacc[ft][e] = 0.f;
    }
    float bsum = 0.f;

    int slot = 0;
    for (int t = t0; t < t1; ++t) {
      lce_wait_vm<G::VM>();                               // tile t has landed (tile t + 1 may still be in flight)
      __builtin_amdgcn_s_barrier();                       // ... for every wave; and every wave is done with tile t - 1
      issue(t + 2, slot >= 1 ? slot - 1 : LCE_NBUF - 1);
      const unsigned char* buf = lce_smem + slot * G::BUF;
      const float4* yvp = reinterpret_cast<const float4*>(buf + G::IMG + w * 256);
      f32x16 acc[NX];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 y = yvp[2 * i + h];                   // rows 8 i + 4 h .. + 3 of the tile = accumulator registers 4 i .. 4 i + 3
#pragma unroll
        for (int n = 0; n < NX; ++n) {                     // the accumulator starts at the Y-side constant; the X-side one joins in the exponent
          acc[n][4 * i + 0] = y.x; acc[n][4 * i + 1] = y.y; acc[n][4 * i + 2] = y.z; acc[n][4 * i + 3] = y.w;
        }
      }
#pragma unroll
      for (int ks = 0; ks < G::NKS; ++ks) {
        const bf16x8 ya = *reinterpret_cast<const bf16x8*>(buf + rb[ks & 1] + 512 * (ks >> 1));
#pragma unroll
        for (int n = 0; n < NX; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, xf[n][ks], acc[n], 0, 0, 0);
      }
      if constexpr (MODE == LCE_FWD) {
#pragma unroll
        for (int n = 0; n < NX; ++n) {
          float mx = acc[n][0];
#pragma unroll
          for (int e = 1; e < 16; ++e) mx = fmaxf(mx, acc[n][e]);
          const float mn = fmaxf(fmaxf(sm[n], mx), -1e30f);
          const float mb = -mn * LCE_L2E;
          float s = ss[n] * __builtin_amdgcn_exp2f(fmaf(sm[n], LCE_L2E, mb));
#pragma unroll
          for (int e = 0; e < 16; ++e) s += __builtin_amdgcn_exp2f(fmaf(acc[n][e], LCE_L2E, mb));
          sm[n] = mn;
          ss[n] = s;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[0][e] = __builtin_amdgcn_exp2f(fmaf(acc[0][e], LCE_L2E, xvl));
          if constexpr (MODE == LCE_DE) bsum += acc[0][e];
        }
        const bf16x8 pf0 = lce_pack8(acc[0], 0), pf1 = lce_pack8(acc[0], 8);
#pragma unroll
        for (int ft = 0; ft < G::NFT; ++ft) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            union { struct { lce_s4 lo, hi; } p; bf16x8 v; } u;
            u.p.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lce_s4 __attribute__((address_space(3)))*)(buf + tb[0] + G::SUB * (2 * s) + 512 * ft));
            u.p.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lce_s4 __attribute__((address_space(3)))*)(buf + tb[1] + G::SUB * (2 * s + 1) + 512 * ft));
            dacc[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, s ? pf1 : pf0, dacc[ft], 0, 0, 0);
          }
        }
      }
      slot = slot + 1 == LCE_NBUF ? 0 : slot + 1;
    }
    lce_wait_vm<0>();                                      // the two tail fetches

    if constexpr (MODE == LCE_FWD) {
#pragma unroll
      for (int n = 0; n < NX; ++n) {
        const float mo = __shfl_xor(sm[n], 32, 64), so = __shfl_xor(ss[n], 32, 64);
        const float mn = fmaxf(sm[n], mo);
        const float s = ss[n] * __builtin_amdgcn_exp2f((sm[n] - mn) * LCE_L2E) + so * __builtin_amdgcn_exp2f((mo - mn) * LCE_L2E);
        if (h == 0) {
          const size_t o = (size_t)j * XR + w * 64 + 32 * n + r;
          a.part_m[o] = mn;
          a.part_s[o] = s;
        }
      }
    } else {
      float4* dp = reinterpret_cast<float4*>(a.dpart) + ((size_t)j * LCE_NW + w) * G::NFT * 256 + lane;
#pragma unroll
      for (int ft = 0; ft < G::NFT; ++ft)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          dp[(ft * 4 + i) * 64] = make_float4(dacc[ft][4 * i], dacc[ft][4 * i + 1], dacc[ft][4 * i + 2], dacc[ft][4 * i + 3]);
      if constexpr (MODE == LCE_DE) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (h == 0 && x0 + r < nx) atomicAdd(a.dsum + x0 + r, bsum);
      }
    }
  }
}

// ---- rows -> bf16 images ----------------------------------------------------------------------------------------------------------------
// dst[i] = bf16(src[idx ? idx[i] : i]) for i < n, zero rows for n <= i < n rounded up to `padto`; cpad (optional): a per-row constant
// copied for i < n and set to -inf on the zero rows (the bias of the item image).
struct LcePackArgs {
  const float* src; int lds; const int32_t* idx; const int32_t* n_dev; int n; int padto; int K; __bf16* dst;
  const float* csrc; float* cdst;
};
__global__ __launch_bounds__(256) void k_lce_pack(LcePackArgs a) {
  const int n = a.n_dev ? min(*a.n_dev, a.n) : a.n;
  const int npad = (n + a.padto - 1) / a.padto * a.padto;
  const int k8 = a.K / 8;
  const size_t total = (size_t)npad * k8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / k8), c = (int)(i % k8);
    bf16x8 o;
    if (row < n) {
      const float* s = a.src + (size_t)(a.idx ? a.idx[row] : row) * a.lds + 8 * c;
      const float4 u = *reinterpret_cast<const float4*>(s), v = *reinterpret_cast<const float4*>(s + 4);
      o[0] = (__bf16)u.x; o[1] = (__bf16)u.y; o[2] = (__bf16)u.z; o[3] = (__bf16)u.w;
      o[4] = (__bf16)v.x; o[5] = (__bf16)v.y; o[6] = (__bf16)v.z; o[7] = (__bf16)v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
    }
    *reinterpret_cast<bf16x8*>(a.dst + (size_t)row * a.K + 8 * c) = o;
    if (a.cdst && c == 0) a.cdst[row] = row < n ? a.csrc[row] : -INFINITY;
  }
}

// ---- forward combine: lse, loss, and the rank-one label terms of dE / dbias ------------------------------------------------------------
struct LceCombineArgs {
  const float* part_m; const float* part_s;
  const __bf16* Hb; const __bf16* Eb; const float* bias; const int32_t* labels;
  const int32_t* m_dev; int mcap; int V; int K; int slots;
  const float* inv_count; float* loss64; float* nlse; float* lse_out;
  float* dE; int lddE; float* dbias;
};
template <int KD>
__global__ __launch_bounds__(256) void k_lce_combine(LceCombineArgs a) {
  const int M = a.m_dev ? min(*a.m_dev, a.mcap) : a.mcap;
  const int Mpad = (M + LCE_XR_FWD - 1) / LCE_XR_FWD * LCE_XR_FWD;
  const LceSplit sp = lce_split(M, LCE_XR_FWD, a.slots, (a.V + 31) / 32);
  const float wn = *a.inv_count;
  const int lane = threadIdx.x & 63;
  __shared__ float lce_wsum[4];
  float lsum = 0.f;                                       // lane 0: this wave's share of the loss
  for (int m0 = blockIdx.x * 16; m0 < Mpad; m0 += gridDim.x * 16)        // a block owns 16 consecutive rows: one loss atomic per 16 rows
  for (int m = m0 + (threadIdx.x >> 6); m < m0 + 16; m += 4) {
    if (m >= M) {
      if (lane == 0) a.nlse[m] = -INFINITY;
      continue;
    }
    const int xb = m / LCE_XR_FWD, xr = m - xb * LCE_XR_FWD;
    const int lab = a.labels[m];
    float pm = -INFINITY, ps = 0.f;
    if (lane < sp.C) {
      const size_t o = (size_t)(lane * sp.nxb + xb) * LCE_XR_FWD + xr;
      pm = a.part_m[o];
      ps = a.part_s[o];
    }
    // lane <-> consecutive columns: every atomic wave-instruction covers 256 contiguous bytes of the table row (the shape that runs at the
    // full float-atomic rate); all loads of the row are issued ahead of the atomics (they may alias for the compiler)
    float dot = 0.f;
    float hv[KD / 64], ev[KD / 64];
#pragma unroll
    for (int j = 0; j < KD / 64; ++j) {                    // unconditional: the loads issue back to back (one round trip, not KD / 32)
      hv[j] = (float)a.Hb[(size_t)m * KD + lane + 64 * j];
      ev[j] = (float)a.Eb[(size_t)lab * KD + lane + 64 * j];
    }
#pragma unroll
    for (int j = 0; j < KD / 64; ++j) dot = fmaf(hv[j], ev[j], dot);
#pragma unroll
    for (int j = 0; j < KD / 64; ++j) atomicAdd(a.dE + (size_t)lab * a.lddE + lane + 64 * j, -wn * hv[j]);
    const float mx = wave_max(pm);
    const float s = wave_sum(ps * __builtin_amdgcn_exp2f((pm - mx) * LCE_L2E));
    const float lse = mx + __logf(s);
    dot = wave_sum(dot);
    if (lane == 0) {
      lsum += wn * (lse - (dot + a.bias[lab]));
      a.nlse[m] = __logf(wn) - lse;
      if (a.lse_out) a.lse_out[m] = lse;
      atomicAdd(a.dbias + lab, -wn);
    }
  }
  // one loss atomic per BLOCK: an atomic per row (5.9k adds on 64 addresses within a few microseconds) was a ~90-deep same-address chain
  // across XCDs and cost 34 of this kernel's 44 us
  if (lane == 0) lce_wsum[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (lce_wsum[0] + lce_wsum[1]) + (lce_wsum[2] + lce_wsum[3]);
    if (t != 0.f) atomicAdd(a.loss64 + (blockIdx.x & 63), t);
  }
}

// ---- partial products -> gradients ----------------------------------------------------------------------------------------------------
// One block per 32 X rows (one wave's share of an item): sums the C partials in register order (coalesced), transposes through LDS and
// writes whole rows.  DH: dh[rows[x]] = sum - (1/n) E[label[x]] (plain store, the masked rows are distinct); DE: dE[x] += sum.
struct LceReduceArgs {
  const float* dpart; const int32_t* nx_dev; int nx; const int32_t* ny_dev; int ny; int slots; int K;
  const int32_t* rows; const int32_t* labels; const __bf16* Eb; const float* inv_count;   // DH only (rows != NULL)
  float* out; int ldo;
};
template <int KD>
__global__ __launch_bounds__(256) void k_lce_reduce(LceReduceArgs a) {
  using G = LceGeo<KD>;
  constexpr int RS = KD + 4;
  extern __shared__ __attribute__((aligned(16))) float lce_red[];        // [32][KD + 4]
  const int nx = a.nx_dev ? min(*a.nx_dev, a.nx) : a.nx;
  const int ny = a.ny_dev ? min(*a.ny_dev, a.ny) : a.ny;
  const LceSplit sp = lce_split(nx, LCE_XR_BWD, a.slots, (ny + 31) / 32);
  const int nb = (nx + 31) / 32;
  for (int xb32 = blockIdx.x; xb32 < nb; xb32 += gridDim.x) {
    const int xb = xb32 / LCE_NW, w = xb32 % LCE_NW;
    for (int idx = threadIdx.x; idx < G::NFT * 256; idx += 256) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int c = 0; c < sp.C; ++c) {
        const float4 v = reinterpret_cast<const float4*>(a.dpart)[((size_t)(c * sp.nxb + xb) * LCE_NW + w) * G::NFT * 256 + idx];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int ft = idx >> 8, i = (idx >> 6) & 3, lane = idx & 63;
      *reinterpret_cast<float4*>(lce_red + (lane & 31) * RS + 32 * ft + 8 * i + 4 * (lane >> 5)) = s;
    }
    __syncthreads();
    const float wn = a.rows ? *a.inv_count : 0.f;
    for (int idx = threadIdx.x; idx < 32 * (KD / 4); idx += 256) {
      const int xl = idx / (KD / 4), f4 = idx % (KD / 4), x = xb32 * 32 + xl;
      if (x < nx) {
        float4 s = *reinterpret_cast<const float4*>(lce_red + xl * RS + 4 * f4);
        if (a.rows) {
          const __bf16* e = a.Eb + (size_t)a.labels[x] * KD + 4 * f4;
          s.x -= wn * (float)e[0]; s.y -= wn * (float)e[1]; s.z -= wn * (float)e[2]; s.w -= wn * (float)e[3];
          *reinterpret_cast<float4*>(a.out + (size_t)a.rows[x] * a.ldo + 4 * f4) = s;
        } else {
          float4* o = reinterpret_cast<float4*>(a.out + (size_t)x * a.ldo + 4 * f4);
          float4 v = *o;
          v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
          *o = v;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace adt
