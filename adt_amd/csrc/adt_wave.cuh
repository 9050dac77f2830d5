// Wave-local row-chain toolkit (d == 64).
//
// One wave owns a 16-token tile end to end: every activation of the tile lives in registers in the MFMA C
// layout ("CT": lane (c, g) holds rows 4g+r, columns 16nt+c), every product is a 16x64x64 MFMA chain whose
// B operand comes from a weight image that the workgroup staged into LDS once, and LDS is otherwise used only
// as a per-wave 16x64 scratch to change layout (C layout -> A fragments, or <-> full 256-byte rows so that all
// global traffic is float4-coalesced).  There is NO workgroup barrier inside a chain: waves only share the
// read-only weight images and the ds_add_f32-accumulated weight-gradient images, so a CU hides latency by
// running 8-16 independent waves.
//
// Weight-gradient products dW = dY^T X contract over the 16 tile rows; both operands are taken straight from
// CT registers (accumulator-as-operand identity: slot (g, j<4) of the k-step is row 4g+j), the upper 16 slots
// are zero.
#pragma once
#include "adt_common.cuh"

namespace adt {

struct CT {            // 16 x 64 fp32 tile in C layout
  f32x4 v[4];          // v[nt][r] = element (row 4g + r, col 16 nt + c)
};

constexpr int WV_RS = 68;              // per-wave scratch row stride (floats)
constexpr int WV_SCR = 16 * WV_RS;     // floats per wave scratch
constexpr int DW_RS = 68;              // fp32 weight-gradient image row stride
constexpr int DW_IMG = 64 * DW_RS;

template <int PREC> struct WImg;       // weight image element type / row stride
template <> struct WImg<PREC_F32> { typedef float T; static constexpr int RS = 68; };
template <> struct WImg<PREC_BF16> { typedef __bf16 T; static constexpr int RS = 72; };

template <int PREC> struct OpFrag;     // MFMA operand fragment (8 k-slots)
template <> struct OpFrag<PREC_F32> { Frag8 f; };
template <> struct OpFrag<PREC_BF16> { bf16x8 f; };

ADT_DEVICE_INLINE void wave_fence() {
  // LDS operations of one wave execute in program order; this only stops the compiler from reordering them
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

template <int PREC>
ADT_DEVICE_INLINE OpFrag<PREC> to_op(const Frag8& a) {
  OpFrag<PREC> o;
  if constexpr (PREC == PREC_BF16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o.f[j] = (__bf16)a.v[j];
  } else {
    o.f = a;
  }
  return o;
}

template <int PREC>
ADT_DEVICE_INLINE f32x4 mma_op(f32x4 acc, const OpFrag<PREC>& a, const OpFrag<PREC>& b) {
  if constexpr (PREC == PREC_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.f, b.f, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f.v[j], b.f.v[j], acc, 0, 0, 0);
    return acc;
  }
}

// ---- weight images ---------------------------------------------------------------------------------
// plain image: img[n][k] = W[n][k]  (forward products y = x W^T: rows = output column n, contraction k)
// transposed : img[k][n] = W[n][k]  (backward products dx = dy W: rows = output column k, contraction n)
template <int PREC, int NTHREADS>
ADT_DEVICE_INLINE void stage_wimg(typename WImg<PREC>::T* img, const float* W, bool transposed) {
  constexpr int RS = WImg<PREC>::RS;
  typedef typename WImg<PREC>::T T;
  for (int i = threadIdx.x; i < 64 * 16; i += NTHREADS) {
    const int n = i >> 4, k4 = (i & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(W + n * 64 + k4);
    if (!transposed) {
      img[n * RS + k4 + 0] = (T)v.x; img[n * RS + k4 + 1] = (T)v.y; img[n * RS + k4 + 2] = (T)v.z; img[n * RS + k4 + 3] = (T)v.w;
    } else {
      img[(k4 + 0) * RS + n] = (T)v.x; img[(k4 + 1) * RS + n] = (T)v.y; img[(k4 + 2) * RS + n] = (T)v.z; img[(k4 + 3) * RS + n] = (T)v.w;
    }
  }
}

// Pre-packed weight images (bf16 mode): adt_pack_wimg writes, once per step, both LDS images of every 64 x 64 weight block of the
// flat parameter buffer -- plain and transposed, [64][RS = 72] bf16 each, exactly the bytes stage_wimg would produce -- so a
// kernel's staging is a straight 16-byte copy instead of 1,024 float4 loads, 4,096 conversions and 4,096 two-byte LDS stores per
// image (14,000 cycles for the six images of an encoder layer, measured with s_memtime stamps).  The block whose fp32 weights
// start at `base + off` keeps its images at img + 6 * off (bf16 elements): plain first, transposed WPACK_IMG elements later.
constexpr int WPACK_IMG = 64 * 72;

// ---- pre-packed weight images: the 64 x 64 fp32 block at base + off as four bf16 images of 64 x 72 (plain, transposed, and their slot-ordered
// forms for the transposed chains of adt_tt.cuh) at img + 6 * off.  One workgroup of 256 threads per block (k_pack_wimg, or the trailing
// blocks of k_step_begin).
struct PackArgs {
  const float* base;      // start of the packed parameter range
  __bf16* img;
  int n;
  int off[256];           // float offsets (relative to base) of the 64 x 64 blocks
};
// PARTS workgroups share one block (quarter q packs rows 16 q .. 16 q + 15 in one iteration per thread): as one workgroup per block the
// four dependent iterations (a load, sixteen 2-byte stores each) made the packing the long pole of k_step_begin
template <int PARTS = 1>
__device__ __forceinline__ void pack_wimg_block(const PackArgs& a, int blk_part) {
  const int blk = blk_part / PARTS, part = blk_part % PARTS;
  const int off = a.off[blk];
  const float* W = a.base + off;
  __bf16* plain = a.img + 6 * (size_t)off;
  __bf16* trans = plain + WPACK_IMG;
  __bf16* splain = plain + 2 * WPACK_IMG;       // slot-ordered forms for the transposed chains (adt_tt.cuh): column 32 kb + 4 g' + q + 16 s
  __bf16* strans = plain + 3 * WPACK_IMG;       // of a row sits at 32 kb + 8 g' + 4 s + q
  for (int i = threadIdx.x + part * (64 * 16 / PARTS); i < (part + 1) * (64 * 16 / PARTS); i += 256) {
    const int n = i >> 4, k4 = (i & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(W + n * 64 + k4);
    const float x[4] = {v.x, v.y, v.z, v.w};
    const int ns = (n & 32) + 8 * ((n >> 2) & 3) + 4 * ((n >> 4) & 1) + (n & 3);      // slot position of column n in a transposed row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k4 + j;
      const int ks = (k & 32) + 8 * ((k >> 2) & 3) + 4 * ((k >> 4) & 1) + (k & 3);
      plain[n * 72 + k] = (__bf16)x[j];
      trans[k * 72 + n] = (__bf16)x[j];
      splain[n * 72 + ks] = (__bf16)x[j];
      strans[k * 72 + ns] = (__bf16)x[j];
    }
  }
}
// The same four images of ONE block by one workgroup of 256 threads through LDS (sp: 4 * WPACK_IMG bf16): the fp32 block is read with 16-byte
// loads, the images are assembled in LDS (2-byte LDS stores) and leave with 16-byte global stores.  pack_wimg_block writes the images with
// 16k scattered 2-byte global stores per block, the slowest store form there is (MI355X_MICROARCH.md, "stores of each flavour").
__device__ __forceinline__ void pack_wimg_block_lds(const PackArgs& a, int blk, __bf16* sp) {
  const int off = a.off[blk];
  const float* W = a.base + off;
  float4 v[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const float4*>(W + (threadIdx.x + it * 256) * 4);
  for (int i = threadIdx.x; i < 4 * WPACK_IMG / 8; i += 256) reinterpret_cast<uint4*>(sp)[i] = make_uint4(0u, 0u, 0u, 0u);      // the row padding
  __syncthreads();
  __bf16 *plain = sp, *trans = sp + WPACK_IMG, *splain = sp + 2 * WPACK_IMG, *strans = sp + 3 * WPACK_IMG;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = threadIdx.x + it * 256;
    const int n = i >> 4, k4 = (i & 15) * 4;
    const float x[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
    const int ns = (n & 32) + 8 * ((n >> 2) & 3) + 4 * ((n >> 4) & 1) + (n & 3);      // slot position of column n in a transposed row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k4 + j;
      const int ks = (k & 32) + 8 * ((k >> 2) & 3) + 4 * ((k >> 4) & 1) + (k & 3);
      plain[n * 72 + k] = (__bf16)x[j];
      trans[k * 72 + n] = (__bf16)x[j];
      splain[n * 72 + ks] = (__bf16)x[j];
      strans[k * 72 + ns] = (__bf16)x[j];
    }
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(a.img + 6 * (size_t)off);
  for (int i = threadIdx.x; i < 4 * WPACK_IMG / 8; i += 256) dst[i] = reinterpret_cast<const uint4*>(sp)[i];
}
struct WPack {
  const float* base;       // start of the packed parameter range (the positional table: everything after the item table)
  const __bf16* img;       // nullptr: not packed (fp32-exact mode, or a caller that did not pack)
};

template <int PREC, int NTHREADS>
ADT_DEVICE_INLINE void stage_w(typename WImg<PREC>::T* img, const float* W, bool transposed, const WPack& wp) {
  if constexpr (PREC == PREC_BF16) {
    if (wp.img) {
      const uint4* src = reinterpret_cast<const uint4*>(wp.img + 6 * (W - wp.base) + (transposed ? WPACK_IMG : 0));
      uint4* dst = reinterpret_cast<uint4*>(img);
      for (int i = threadIdx.x; i < WPACK_IMG * 2 / 16; i += NTHREADS) dst[i] = src[i];
      return;
    }
  }
  stage_wimg<PREC, NTHREADS>(img, W, transposed);
}

// N images.  (Issuing every global load of the set before the first LDS store -- 42 x 16 B in flight per thread for six images --
// was measured SLOWER than image after image: 18,700 against 10,300 cycles for an encoder layer's six images with all 256
// workgroups staging at once; converting from fp32 in the kernel took 14,000.)
template <int PREC, int NTHREADS, int N>
ADT_DEVICE_INLINE void stage_w_set(typename WImg<PREC>::T* const (&img)[N], const float* const (&W)[N], bool transposed, const WPack& wp) {
#pragma unroll
  for (int k = 0; k < N; ++k) stage_w<PREC, NTHREADS>(img[k], W[k], transposed, wp);
}

template <int PREC>
ADT_DEVICE_INLINE OpFrag<PREC> wfrag(const typename WImg<PREC>::T* img, int nt, int kb, int c, int g) {
  constexpr int RS = WImg<PREC>::RS;
  OpFrag<PREC> o;
  const typename WImg<PREC>::T* p = img + (16 * nt + c) * RS + kb * 32 + 8 * g;
  if constexpr (PREC == PREC_BF16) {
    o.f = *reinterpret_cast<const bf16x8*>(p);
  } else {
    o.f = frag_contig(p);
  }
  return o;
}

// ---- layout changes through the per-wave scratch ------------------------------------------------------
// global rows -> scratch (4 float4 per lane; one instruction covers 4 full 256-byte rows)
ADT_DEVICE_INLINE void rows_to_scr(float* scr, const float* g, int ld, int row0, int T, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = q * 64 + lane, r = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < T) v = *reinterpret_cast<const float4*>(g + (size_t)(row0 + r) * ld + c4);
    *reinterpret_cast<float4*>(scr + r * WV_RS + c4) = v;
  }
}

struct RowRegs { float4 v[4]; };   // a tile as 4 full-row float4 pieces per lane (prefetch form)

ADT_DEVICE_INLINE RowRegs rows_load(const float* g, int ld, int row0, int T, int lane) {
  RowRegs x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = q * 64 + lane, r = i >> 4, c4 = (i & 15) * 4;
    x.v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < T) x.v[q] = *reinterpret_cast<const float4*>(g + (size_t)(row0 + r) * ld + c4);
  }
  return x;
}

// the same tile from a tensor SAVED by the forward: fp32 rows, or -- when the transposed-chain forward (adt_seqfwd_tt.cuh) wrote it --
// bf16 rows of 64 (128 bytes; the buffer keeps its fp32-sized slot in the workspace) in the register order of a transposed tile:
// features 4q .. 4q+3 (q = 4 nt + g) sit at elements 16 g + 4 nt .. + 3 (adt_tt.cuh: tt_store_bf16)
ADT_DEVICE_INLINE RowRegs rows_load_saved(const float* g, int row0, int T, int lane, int saved_bf16) {
  if (!saved_bf16) return rows_load(g, 64, row0, T, lane);
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const __bf16* gb = reinterpret_cast<const __bf16*>(g);
  RowRegs x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = q * 64 + lane, r = i >> 4, c4 = (i & 15) * 4;
    x.v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < T) {
      const bf16x4_t b = *reinterpret_cast<const bf16x4_t*>(gb + (size_t)(row0 + r) * 64 + 16 * ((c4 >> 2) & 3) + 4 * (c4 >> 4));
      x.v[q] = make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]);
    }
  }
  return x;
}

ADT_DEVICE_INLINE void rows_put(float* scr, const RowRegs& x, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = q * 64 + lane, r = i >> 4, c4 = (i & 15) * 4;
    *reinterpret_cast<float4*>(scr + r * WV_RS + c4) = x.v[q];
  }
}

ADT_DEVICE_INLINE void scr_to_rows(float* g, int ld, const float* scr, int row0, int T, int lane, bool accumulate = false) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = q * 64 + lane, r = i >> 4, c4 = (i & 15) * 4;
    if (row0 + r < T) {
      float4 v = *reinterpret_cast<const float4*>(scr + r * WV_RS + c4);
      float* dst = g + (size_t)(row0 + r) * ld + c4;
      if (accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(dst);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<float4*>(dst) = v;
    }
  }
}

ADT_DEVICE_INLINE CT scr_to_ct(const float* scr, int c, int g) {
  CT t;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) t.v[nt][r] = scr[(4 * g + r) * WV_RS + 16 * nt + c];
  return t;
}

ADT_DEVICE_INLINE void ct_to_scr(float* scr, const CT& t, int c, int g) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[(4 * g + r) * WV_RS + 16 * nt + c] = t.v[nt][r];
}

template <int PREC>
struct AFrags { OpFrag<PREC> kb[2]; };

template <int PREC>
ADT_DEVICE_INLINE AFrags<PREC> scr_to_a(const float* scr, int c, int g) {
  AFrags<PREC> a;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) a.kb[kb] = to_op<PREC>(frag_contig(scr + c * WV_RS + kb * 32 + 8 * g));
  return a;
}

// convenience: global tile -> CT (through scratch)
ADT_DEVICE_INLINE CT load_ct(float* scr, const float* gptr, int ld, int row0, int T, int lane, int c, int g) {
  wave_fence();
  rows_to_scr(scr, gptr, ld, row0, T, lane);
  wave_fence();
  return scr_to_ct(scr, c, g);
}

ADT_DEVICE_INLINE void store_ct(float* scr, float* gptr, int ld, const CT& t, int row0, int T, int lane, int c, int g,
                                bool accumulate = false) {
  wave_fence();
  ct_to_scr(scr, t, c, g);
  wave_fence();
  scr_to_rows(gptr, ld, scr, row0, T, lane, accumulate);
}

// CT -> A fragments (through scratch)
template <int PREC>
ADT_DEVICE_INLINE AFrags<PREC> ct_to_a(float* scr, const CT& t, int c, int g) {
  wave_fence();
  ct_to_scr(scr, t, c, g);
  wave_fence();
  return scr_to_a<PREC>(scr, c, g);
}

// out[16 x 64] = A[16 x 64] * img^T  (img rows = output columns)
template <int PREC>
ADT_DEVICE_INLINE CT gemm_w(const AFrags<PREC>& a, const typename WImg<PREC>::T* img, int c, int g) {
  // The image is loop-invariant, so the compiler would hoist all 8 fragments per weight (32 VGPRs each) out of the
  // tile loop and spill; re-reading them from LDS per tile is far cheaper than the lost occupancy.  Only the OFFSET
  // is made opaque: an opaque pointer loses its LDS address space and turns the reads into flat loads, whose
  // vmcnt(0) waits would also drain every prefetch in flight.
  int opaque_zero = 0;
  asm volatile("" : "+v"(opaque_zero));
  img += opaque_zero;
  CT o;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    o.v[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) o.v[nt] = mma_op<PREC>(o.v[nt], a.kb[kb], wfrag<PREC>(img, nt, kb, c, g));
  }
  return o;
}

// Weight-gradient accumulator of one 64x64 weight, resident in registers (AGPR-able) for the whole kernel:
// t[nt*4 + kt] is the 16x16 tile (rows n = 16nt + 4g + r, cols k = 16kt + c); bias[nt] = this lane's partial
// column sums of dY (rows 4g..4g+3 of every tile it has seen).
struct WAcc {
  f32x4 t[16];
  float bias[4];
};

ADT_DEVICE_INLINE void wacc_zero(WAcc& a) {
#pragma unroll
  for (int i = 0; i < 16; ++i) a.t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) a.bias[i] = 0.f;
}

// acc += dY^T X over the 16 tile rows (accumulator-as-operand: both operands straight from CT registers)
template <int PREC>
ADT_DEVICE_INLINE void dw_accum(WAcc& acc, const CT& dy, const CT& x) {
  OpFrag<PREC> fb[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    Frag8 b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { b.v[j] = x.v[kt][j]; b.v[4 + j] = 0.f; }
    fb[kt] = to_op<PREC>(b);
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    Frag8 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.v[j] = dy.v[nt][j]; a.v[4 + j] = 0.f; }
    const OpFrag<PREC> fa = to_op<PREC>(a);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc.t[nt * 4 + kt] = mma_op<PREC>(acc.t[nt * 4 + kt], fa, fb[kt]);
    acc.bias[nt] += dy.v[nt][0] + dy.v[nt][1] + dy.v[nt][2] + dy.v[nt][3];
  }
}


// ---- cooperative weight gradients -------------------------------------------------------------------------
// dW = dY^T X contracts over ALL rows.  Each wave publishes the C-layout registers of its 16-row dY and X tiles
// to an LDS exchange area ("register image": [nt][lane][r], so the consumer lane reads the SAME lane index with
// one 8/16-byte load), the workgroup synchronises once, and every wave then accumulates only ITS output tiles
// (16 tiles of 16x16 per 64x64 weight, dealt round-robin over the NW waves) over the NW published row tiles with
// full K=32 MFMAs: slots j<4 of a k-step are rows 4g+j of tile 2s, slots j>=4 rows 4g+j-4 of tile 2s+1
// (accumulator-as-operand identity).  Accumulators are 4 registers per owned tile instead of 64 per weight, so the
// chain keeps a high wave count.  Two exchange areas alternate => ONE barrier per product.
template <int PREC, int NW>
struct Coop {
  typedef typename WImg<PREC>::T XT;
  static constexpr int TILE = 4 * 64 * 4;                 // elements per published CT
  static constexpr int AREA = NW * 2 * TILE;              // elements per exchange area (dY and X of every wave)
  static constexpr int NOWN = (16 + NW - 1) / NW;         // output tiles owned by a wave
  static constexpr size_t bytes = 2 * (size_t)AREA * sizeof(XT);
  XT* base;
  int w, lane;
  int parity;
  int skip;
  __device__ Coop(void* p, int wave, int lane_, int skip_ = 0) : base(reinterpret_cast<XT*>(p)), w(wave), lane(lane_), parity(0), skip(skip_) {}

  ADT_DEVICE_INLINE void put(XT* dst, const CT& t) const {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      XT* q = dst + (nt * 64 + lane) * 4;
      if constexpr (PREC == PREC_BF16) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (__bf16)t.v[nt][r];
        *reinterpret_cast<bf16x4*>(q) = v;
      } else {
        *reinterpret_cast<f32x4*>(q) = t.v[nt];
      }
    }
  }

  ADT_DEVICE_INLINE OpFrag<PREC> frag(const XT* area, int which, int s, int tile_idx) const {
    // k-step s: tiles 2s and 2s+1 of the published set, sub-tile (nt or kt) = tile_idx
    const XT* p0 = area + ((2 * s) * 2 + which) * TILE + (tile_idx * 64 + lane) * 4;
    const XT* p1 = area + ((2 * s + 1) * 2 + which) * TILE + (tile_idx * 64 + lane) * 4;
    OpFrag<PREC> o;
    if constexpr (PREC == PREC_BF16) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      const bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
      const bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { o.f[j] = a[j]; o.f[4 + j] = b[j]; }
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p0);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { o.f.v[j] = a[j]; o.f.v[4 + j] = b[j]; }
    }
    return o;
  }

  // acc[i] (i < NOWN) accumulates output tile id = w + i*NW  (nt = id >> 2, kt = id & 3).  ALL waves must call.
  ADT_DEVICE_INLINE void product(f32x4 (&acc)[NOWN], const CT& dy, const CT& x) {
    if (skip & 1) return;
    XT* area = base + parity * AREA;
    parity ^= 1;
    put(area + (w * 2 + 0) * TILE, dy);
    put(area + (w * 2 + 1) * TILE, x);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
      const int id = w + i * NW;
      if (id < 16) {
        const int nt = id >> 2, kt = id & 3;
#pragma unroll
        for (int s = 0; s < NW / 2; ++s) acc[i] = mma_op<PREC>(acc[i], frag(area, 0, s, nt), frag(area, 1, s, kt));
      }
    }
  }

  // add the owned tiles of one finished weight gradient to global memory (row-major 64 x 64)
  ADT_DEVICE_INLINE void flush(const f32x4 (&acc)[NOWN], float* gW, int c, int g) const {
    if (skip & 2) return;
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
      const int id = w + i * NW;
      if (id < 16) {
        const int nt = id >> 2, kt = id & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(gW + (16 * nt + 4 * g + r) * 64 + 16 * kt + c, acc[i][r]);
      }
    }
  }
};

// per-lane partial column sums (bias / LayerNorm dgamma / dbeta gradients); reduced at kernel end
struct VAcc { float v[4]; };
ADT_DEVICE_INLINE void vacc_zero(VAcc& a) { a.v[0] = a.v[1] = a.v[2] = a.v[3] = 0.f; }
ADT_DEVICE_INLINE void colsum_accum(VAcc& a, const CT& t) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) a.v[nt] += t.v[nt][0] + t.v[nt][1] + t.v[nt][2] + t.v[nt][3];
}

// End-of-kernel reduction of one weight accumulator over the NW waves of the workgroup: every wave parks its
// tiles in a private LDS image, then all threads sum the NW images and add the result to global memory with
// one float atomic per element (coalesced 256-byte rows).  red: NW * DW_IMG floats.  Must be called by all
// threads (contains workgroup barriers).
template <int NW>
ADT_DEVICE_INLINE void wacc_flush(const WAcc& acc, float* red, float* gW, float* gb, int w, int c, int g) {
  __syncthreads();
  float* mine = red + w * DW_IMG;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(16 * nt + 4 * g + r) * DW_RS + 16 * kt + c] = acc.t[nt * 4 + kt][r];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += NW * 64) {
    const int o = (i >> 6) * DW_RS + (i & 63);
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) s += red[ww * DW_IMG + o];
    atomicAdd(gW + i, s);
  }
  if (gb) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float s = acc.bias[nt];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) atomicAdd(gb + 16 * nt + c, s);
    }
  }
}

// Reduce up to NV per-lane column-sum accumulators over the whole workgroup and add them to global memory with ONE
// float atomic per column per workgroup.  (Every wave flushing on its own means thousands of serialised atomics on
// the same 64 addresses -- tens of microseconds.)  red: LDS, NV * NW * 64 floats, free at this point; all threads call.
template <int NW, int NV>
ADT_DEVICE_INLINE void vacc_flush_wg(const VAcc* const (&accs)[NV], float* const (&dst)[NV], float* red, int w, int c, int g,
                                     int fold = 64) {
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float s = accs[v]->v[nt];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) red[(v * NW + w) * 64 + 16 * nt + c] = s;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NV * 64; i += NW * 64) {
    const int v = i >> 6, col = i & 63;
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) s += red[(v * NW + ww) * 64 + col];
    if (dst[v]) atomicAdd(dst[v] + (col % fold), s);   // fold < 64: columns with equal col % fold share a destination
  }
}

// ---- LayerNorm in C layout ---------------------------------------------------------------------------
// row r of the tile lives on the 16 lanes with the same g (one value per nt per lane)
ADT_DEVICE_INLINE float row_sum16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

struct LnStat { float rstd[4]; };   // per row r of this lane's group

// xhat (normalised, before gamma/beta) and rstd of a CT
ADT_DEVICE_INLINE CT ln_xhat(const CT& x, float eps, LnStat& st) {
  CT h;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = x.v[0][r] + x.v[1][r] + x.v[2][r] + x.v[3][r];
    s = row_sum16(s);
    const float mu = s * (1.0f / 64);
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { const float d = x.v[nt][r] - mu; h.v[nt][r] = d; q += d * d; }
    q = row_sum16(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 64) + eps);
    st.rstd[r] = rstd;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) h.v[nt][r] *= rstd;
  }
  return h;
}

ADT_DEVICE_INLINE CT ln_apply(const CT& xhat, const float* gamma, const float* beta, int c) {
  CT y;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float gm = gamma[16 * nt + c], bt = beta[16 * nt + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) y.v[nt][r] = xhat.v[nt][r] * gm + bt;
  }
  return y;
}

// dx = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)), dxh = dy * gamma; also accumulates dgamma/dbeta
ADT_DEVICE_INLINE CT ln_bwd_ct(const CT& dy, const CT& xhat, const LnStat& st, const float* gamma, VAcc& dg_acc, VAcc& db_acc,
                               int c, int g) {
  CT t;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) t.v[nt][r] = dy.v[nt][r] * xhat.v[nt][r];
  colsum_accum(dg_acc, t);
  colsum_accum(db_acc, dy);
  CT dx;
  float gm[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) gm[nt] = gamma[16 * nt + c];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float dxh = dy.v[nt][r] * gm[nt];
      dx.v[nt][r] = dxh;
      m1 += dxh;
      m2 += dxh * xhat.v[nt][r];
    }
    m1 = row_sum16(m1) * (1.0f / 64);
    m2 = row_sum16(m2) * (1.0f / 64);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dx.v[nt][r] = st.rstd[r] * (dx.v[nt][r] - m1 - xhat.v[nt][r] * m2);
  }
  return dx;
}

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

ADT_DEVICE_INLINE void ct_add_bias(CT& t, const float* b, int c) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float bv = b[16 * nt + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) t.v[nt][r] += bv;
  }
}

ADT_DEVICE_INLINE void ct_add(CT& a, const CT& b) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) a.v[nt] += b.v[nt];
}

ADT_DEVICE_INLINE CT rows_to_ct(float* scr, const RowRegs& x, int lane, int c, int g) {
  wave_fence();
  rows_put(scr, x, lane);
  wave_fence();
  return scr_to_ct(scr, c, g);
}


}  // namespace adt
