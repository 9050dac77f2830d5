// Shared device-side toolkit for the ADT hot-path kernels (gfx950 / CDNA4 only).
//
// Conventions
//   * wave = 64 lanes; lane l -> c = l & 15 (tile row/col on the lane), g = l >> 4 (k group 0..3).
//   * All matrix products go through ONE primitive, mma16<PREC>(acc, a, b): a 16x16 output tile that
//     contracts 32 k-slots.  Lane (c, g) supplies 8 floats a[j] = A[row c][slot(g, j)],
//     b[j] = B[slot(g, j)][col c].  PREC_F32 issues 8 x v_mfma_f32_16x16x4_f32 (exact fp32, used to pin
//     parity against the oracle), PREC_BF16 converts to bf16 and issues 1 x v_mfma_f32_16x16x32_bf16
//     (fp32 accumulate; the benchmark precision).  Any slot order is legal as long as A and B agree.
//   * C/D layout of a 16x16 tile: col = c, row = 4*g + r for accumulator register r = 0..3.
//   * LDS row tiles are fp32, row stride RS = K + 4 dwords: contiguous 8-float fragments are two
//     ds_read_b128; "strided" fragments (8 rows at one column, rows 4g+j / 16+4g+(j-4)) are conflict-free
//     because 4*RS == 16 (mod 32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { PREC_F32 = 0, PREC_BF16 = 1 };

struct Frag8 {
  float v[8];
};

template <int PREC>
__device__ __forceinline__ f32x4 mma16(f32x4 acc, const Frag8& a, const Frag8& b) {
  if constexpr (PREC == PREC_BF16) {
    bf16x8 pa, pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pa[j] = (__bf16)a.v[j];
      pb[j] = (__bf16)b.v[j];
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
    return acc;
  }
}

// 8 contiguous floats at p (16-byte aligned).
__device__ __forceinline__ Frag8 frag_contig(const float* p) {
  Frag8 f;
  const float4 x = *reinterpret_cast<const float4*>(p);
  const float4 y = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = x.x; f.v[1] = x.y; f.v[2] = x.z; f.v[3] = x.w;
  f.v[4] = y.x; f.v[5] = y.y; f.v[6] = y.z; f.v[7] = y.w;
  return f;
}

// slot(g, j) of the "strided" order: rows 4g + (j&3) + 16*(j>>2) of a 32-row block.
__device__ __forceinline__ int slot_row(int g, int j) { return 4 * g + (j & 3) + 16 * (j >> 2); }

// 8 floats of one row in slot order: columns 4g..4g+3 and 16+4g..16+4g+3 of a 32-column block at p.
__device__ __forceinline__ Frag8 frag_slotc(const float* p, int g) {
  Frag8 f;
  const float4 x = *reinterpret_cast<const float4*>(p + 4 * g);
  const float4 y = *reinterpret_cast<const float4*>(p + 16 + 4 * g);
  f.v[0] = x.x; f.v[1] = x.y; f.v[2] = x.z; f.v[3] = x.w;
  f.v[4] = y.x; f.v[5] = y.y; f.v[6] = y.z; f.v[7] = y.w;
  return f;
}

// 8 floats down one column: base points at [row0][col]; rs = row stride in floats.
__device__ __forceinline__ Frag8 frag_strided(const float* base, int rs, int g) {
  Frag8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = base[slot_row(g, j) * rs];
  return f;
}

// ---------------------------------------------------------------------------------------------
// dropout RNG: lowbias32 hash of (element index ^ key(seed, site)); identical to oracle/rng.py.
__host__ __device__ __forceinline__ uint32_t adt_hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t adt_site_key(uint32_t seed, uint32_t site) {
  return adt_hash32(seed ^ (site * 0x9E3779B9u));
}
// One 32-bit hash serves the four elements 4k .. 4k+3, one byte each: element idx reads byte (idx & 3) of hash32((idx >> 2) ^ key)
// and is dropped iff that byte < thr (an 8-bit threshold, round(p * 256)).  A lane that holds four consecutive elements -- an MFMA
// accumulator register quad -- hashes once per quad (adt_keep4) instead of four times; identical to oracle/rng.py.
__host__ __device__ __forceinline__ bool adt_keep(uint32_t key, uint32_t idx, uint32_t thr) {
  return ((adt_hash32((idx >> 2) ^ key) >> (8u * (idx & 3u))) & 0xFFu) >= thr;
}
// the four keep decisions of elements idx4 .. idx4 + 3 (idx4 % 4 == 0) from one hash: bit r of the result = keep(idx4 + r)
__host__ __device__ __forceinline__ uint32_t adt_keep4(uint32_t key, uint32_t idx4, uint32_t thr) {
  const uint32_t h = adt_hash32((idx4 >> 2) ^ key);
  return ((h & 0xFFu) >= thr ? 1u : 0u) | (((h >> 8) & 0xFFu) >= thr ? 2u : 0u) | (((h >> 16) & 0xFFu) >= thr ? 4u : 0u) |
         ((h >> 24) >= thr ? 8u : 0u);
}

// the keep bits of idx0 .. idx0 + 3 for ANY idx0: one hash when the quad is aligned (`aligned`: wave-uniform, e.g. L % 4 == 0 for the
// L x L probability rows of an attention), four otherwise.  Same decisions as adt_keep per element.
__host__ __device__ __forceinline__ uint32_t adt_keep4_any(uint32_t key, uint32_t idx0, uint32_t thr, bool aligned) {
  if (aligned) return adt_keep4(key, idx0, thr);
  return (adt_keep(key, idx0, thr) ? 1u : 0u) | (adt_keep(key, idx0 + 1u, thr) ? 2u : 0u) | (adt_keep(key, idx0 + 2u, thr) ? 4u : 0u) |
         (adt_keep(key, idx0 + 3u, thr) ? 8u : 0u);
}

// The same four decisions WITHOUT compares: bit 8r + 7 of the result is set iff byte r of h >= thr (1 <= thr <= 255; the other bits
// are scratch).  byte >= thr <=> the 8-bit sum byte + (256 - thr) carries out = majority(byte's bit 7, the constant's bit 7, the carry
// out of the low seven bits); three bit operations for the four bytes (the last is one v_bitop3_b32), no VCC round trips.
// clo = ((256 - thr) & 0x7F) * 0x01010101, chi = ((256 - thr) & 0x80) * 0x01010101 (adt_keep7_consts).
__host__ __device__ __forceinline__ uint32_t adt_keep7(uint32_t h, uint32_t clo, uint32_t chi) {
  const uint32_t lo = (h & 0x7F7F7F7Fu) + clo;
  return (h & lo) | (chi & (h | lo));
}
__host__ __device__ __forceinline__ void adt_keep7_consts(uint32_t thr, uint32_t& clo, uint32_t& chi) {
  const uint32_t c = 256u - thr;
  clo = (c & 0x7Fu) * 0x01010101u;
  chi = (c & 0x80u) * 0x01010101u;
}
// bits 7, 15, 23, 31 of a keep7 word gathered into a nibble (bit r = keep of byte r): the four products land on distinct bits 28 .. 31
__host__ __device__ __forceinline__ uint32_t adt_keep7_nibble(uint32_t k7) { return ((k7 & 0x80808080u) * 0x00204081u) >> 28; }

struct DropCfg {
  const uint32_t* seed;  // device scalar (changes every step; lives in device memory so a captured graph replays)
  uint32_t site;
  uint32_t thr;    // 8-bit threshold round(p * 256): drop iff the element's random byte < thr ; 0 = dropout off
  float scale;     // 1 / (1 - thr / 256): unbiased at the quantised rate
};

__device__ __forceinline__ uint32_t drop_key(const DropCfg& d) { return d.thr ? adt_site_key(*d.seed, d.site) : 0u; }

// LDS-DMA: 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4), no VGPR in between.  lds_dst: WAVE-UNIFORM LDS
// byte address of the 1 KiB piece; the hardware adds lane * 16.  Issued from inline asm (hipcc would put vmcnt(0) in front of every ds_read
// that follows a visible global_load_lds).  The request counts in vmcnt like any load (in issue order), and the compiler does not know
// about it: wait with adt_wait_vm0() before the barrier that publishes the LDS bytes.
__device__ __forceinline__ void adt_glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void adt_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// one contiguous block of `bytes` (a multiple of 1,024) global -> LDS, the 1 KiB pieces dealt round-robin to the NW waves of the workgroup
template <int NW>
__device__ __forceinline__ void adt_glds_block(const void* gsrc, const void* lds_dst, int bytes) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t dst = (uint32_t)(uintptr_t)lds_dst;       // generic pointer to LDS: the low 32 bits are the LDS byte address
  for (int p = w; p * 1024 < bytes; p += NW)
    adt_glds16(reinterpret_cast<const unsigned char*>(gsrc) + p * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(dst + p * 1024));
}

// ---------------------------------------------------------------------------------------------
// reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

#define ADT_DEVICE_INLINE __device__ __forceinline__

// wave sum that stays in the vector ALU: four DPP steps inside each row of 16 lanes, then v_permlane16_swap / v_permlane32_swap across the
// rows (gfx950).  wave_sum above goes through ds_bpermute_b32 six times: a dependent chain of LDS-crossbar round trips.
__device__ __forceinline__ float wave_sum_valu(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const u2v a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v += __builtin_bit_cast(float, (threadIdx.x & 16) ? a[0] : a[1]);
  const unsigned w = __builtin_bit_cast(unsigned, v);
  const u2v b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  v += __builtin_bit_cast(float, (threadIdx.x & 32) ? b[0] : b[1]);
  return v;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop that is unrolled in the front end.  `#pragma unroll` runs late in
// the optimizer: a register array indexed by the loop counter is still dynamically indexed when scalar replacement looks at it, stays an
// alloca and ends up in scratch (the weight-image staging registers of the per-sequence kernels: 96-208 B of scratch per lane).
#include <utility>
template <typename F, int... I>
ADT_DEVICE_INLINE void adt_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
ADT_DEVICE_INLINE void adt_static_for(F&& f) { adt_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Touch every 64-byte line of the kernel-argument segment at kernel entry.  The per-sequence kernels take 300-450 bytes of arguments; hipcc
// loads the fields where they are first needed, so a prologue met four or five scalar-cache misses ONE AFTER THE OTHER (each behind the branch
// or wait that precedes it) before its first vector load was issued.  Requested back to back here, the lines arrive in one round trip and the
// compiler's own s_loads hit the scalar cache.  One asm statement, ending in the wait: the destination SGPR is written asynchronously, so it
// must not be handed back to the compiler before the loads have returned.
template <int BYTES>
ADT_DEVICE_INLINE void adt_prefetch_kernargs() {
  static_assert(BYTES > 0 && BYTES <= 512, "kernel-argument struct of at most 512 bytes");
  const auto p = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned t;
  if constexpr (BYTES <= 128)
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
  else if constexpr (BYTES <= 256)
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t) : "s"(p) : "memory");
  else if constexpr (BYTES <= 384)
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0xc0\n\t"
                 "s_load_dword %0, %1, 0x100\n\ts_load_dword %0, %1, 0x140\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
  else
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0xc0\n\t"
                 "s_load_dword %0, %1, 0x100\n\ts_load_dword %0, %1, 0x140\n\ts_load_dword %0, %1, 0x180\n\ts_load_dword %0, %1, 0x1c0\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p) : "memory");
}

// Zero-fill as an ordinary kernel.  The library never uses hipMemsetAsync: its entry points are captured into HIP graphs by the
// trainers, and on this stack a captured memset NODE was observed to start writing a stale non-zero pattern after a few hundred
// replays (the gradient-norm accumulators then read ~4e30 or NaN for the rest of the process; see DESIGN.md "graph memset").
// A kernel node has no such state.
namespace adt {
static __global__ __launch_bounds__(256) void k_zero_f32(float* p, size_t n) {
  if (((uintptr_t)p & 15) == 0) {      // 16-byte stores over the aligned body, scalar tail
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
static __global__ __launch_bounds__(256) void k_zero_rows_f32(float* p, size_t ld, int cols, size_t rows) {
  const size_t n = rows * (size_t)cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[(i / cols) * ld + (i % cols)] = 0.f;
}
static inline int zero_rows_f32_async(float* p, size_t ld, int cols, size_t rows, hipStream_t s) {
  const size_t n = rows * (size_t)cols;
  if (n == 0) return 0;
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  if (ld == (size_t)cols) hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
  else hipLaunchKernelGGL(k_zero_rows_f32, dim3((unsigned)blocks), dim3(256), 0, s, p, ld, cols, rows);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
static inline int zero_f32_async(float* p, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  size_t blocks = (n + 4095) / 4096;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace adt
