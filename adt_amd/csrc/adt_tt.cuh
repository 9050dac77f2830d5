// Register-resident TRANSPOSED row chains (bf16 MFMA operands, d == 64).
//
// A wave keeps its 16-token tile as X^T: `TT.v[nt][r]` = element (feature 16 nt + 4 g + r, token c) -- the MFMA C layout with the
// FEATURE on the accumulator rows and the token on the lane.  Every product of the layer then is Y^T = W X^T:
//   * the A operand is the weight (row n, 8 contraction slots per lane) -- one ds_read_b128 from a pre-packed, slot-ordered LDS image;
//   * the B operand is X^T itself, taken straight from the accumulator registers of the previous product (accumulator-as-operand:
//     slot (g, j) of a 32-feature block is feature 4g + (j & 3) + 16 (j >> 2), i.e. registers r = j & 3 of tiles 2 kb + (j >> 2));
//   * the result lands in the same layout.
// So a whole chain LayerNorm -> in-projection -> ... -> FFN runs with NO layout change and NO LDS traffic for activations (the
// row-major chains of adt_wave.cuh cross a per-wave LDS scratch twice per product: 57 % of the fused forward's wave cycles were
// s_waitcnt stalls, profiles/r02_*).  Feature reductions (LayerNorm, classifier) are 16 in-lane adds + 2 cross-lane steps; bias,
// gamma and beta are 16 per-lane constants; four consecutive features per register quad mean one dropout hash per quad
// (adt_keep4) and 8- / 16-byte global and LDS accesses.
//
// Attention in this layout: S^T = K Q^T with the key on the accumulator rows (A = a row of the slot-ordered K image, B = the q
// registers), softmax statistics per lane column, P^T as the B operand of O^T = V^T P^T whose A operand (4 consecutive keys of
// one feature) comes from the ROW-major V image through ds_read_b64_tr_b16 -- the output is again a TT tile, ready for out_proj.
#pragma once
#include "adt_attn_bf16.cuh"
#include "adt_wave.cuh"

namespace adt {

struct TT { f32x4 v[4]; };

constexpr int TT_RS = 72;                      // bf16 row stride of every [row][64] LDS image (weights, K, V, ...)
constexpr int TT_WIMG = 64 * TT_RS;

ADT_DEVICE_INLINE bf16x8 tt_pack(const f32x4& lo, const f32x4& hi, float mul = 1.0f) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = (__bf16)(lo[j] * mul); o[4 + j] = (__bf16)(hi[j] * mul); }
  return o;
}

struct TTB { bf16x8 kb[2]; };                  // X^T as the B operands of the two 32-feature contraction blocks

ADT_DEVICE_INLINE TTB tt_bfrags(const TT& x, float mul = 1.0f) {
  TTB b;
  b.kb[0] = tt_pack(x.v[0], x.v[1], mul);
  b.kb[1] = tt_pack(x.v[2], x.v[3], mul);
  return b;
}

// Y^T = W X^T ; img: slot-ordered image of W ([n][32 kb + 8 g + j] = W[n][32 kb + 4 g + (j & 3) + 16 (j >> 2)])
ADT_DEVICE_INLINE TT tt_gemm(const TTB& b, const __bf16* img, int c, int g) {
  int opaque_zero = 0;                         // keep the (loop-invariant) image reads inside the tile loop: see gemm_w
  asm volatile("" : "+v"(opaque_zero));
  img += opaque_zero;
  TT y;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
      acc = mfma_bf16(acc, *reinterpret_cast<const bf16x8*>(img + (16 * nt + c) * TT_RS + 32 * kb + 8 * g), b.kb[kb]);
    y.v[nt] = acc;
  }
  return y;
}

// 16 per-lane constants of a 64-vector (bias, gamma, beta): element (nt, r) = p[16 nt + 4 g + r]
struct TTV { f32x4 v[4]; };
ADT_DEVICE_INLINE TTV tt_vec(const float* p, int g) {
  TTV t;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 x = *reinterpret_cast<const float4*>(p + 16 * nt + 4 * g);
    t.v[nt] = f32x4{x.x, x.y, x.z, x.w};
  }
  return t;
}
ADT_DEVICE_INLINE void tt_add_vec(TT& t, const float* p, int g) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 x = *reinterpret_cast<const float4*>(p + 16 * nt + 4 * g);
    t.v[nt] += f32x4{x.x, x.y, x.z, x.w};
  }
}
ADT_DEVICE_INLINE void tt_add(TT& a, const TT& b) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) a.v[nt] += b.v[nt];
}
ADT_DEVICE_INLINE TT tt_zero() {
  TT t;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) t.v[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  return t;
}

// global fp32 rows <-> TT: `row` points at this lane's token row (64 floats); lanes whose token is absent pass valid = false
// Lanes without a row read a block of zeros instead of skipping the load.  `if (valid) x = *p` made hipcc branch around every load and --
// because the loaded registers are re-packed inside the branch -- wait vmcnt(0) right behind it: the prologues of the per-sequence kernels were
// six to ten SERIAL memory round trips (cdna_hip_programming.md, ".s-level traps" (c)).  Selecting the ADDRESS keeps the loads unconditional
// and back to back; the wait moves to the first use.
static __device__ __attribute__((aligned(256))) const float tt_zero_row[64] = {};
// ids[row] for lanes with a row, 0 for the others -- by address select (see tt_load): `valid ? ids[row] : 0` is a branch with the wait inside
ADT_DEVICE_INLINE int tt_load_id(const int* ids, int row, bool valid) {
  typedef const int __attribute__((address_space(1))) * gi;
  return *(valid ? (gi)(ids + row) : (gi)tt_zero_row);
}
ADT_DEVICE_INLINE TT tt_load(const float* row, bool valid, int g) {
  TT t;
  typedef const f32x4 __attribute__((address_space(1))) * gp4;       // explicitly global: a select of two generic pointers becomes flat_load
  const gp4 p = valid ? (gp4)(row + 4 * g) : (gp4)(tt_zero_row + 4 * g);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) t.v[nt] = p[4 * nt];
  return t;
}
// The branching form (a load only where there is a row).  k_seqtt_attn_pre_bwd keeps it: at its 256-register budget the 26 loads that the
// unconditional form puts in flight at once spilled 208 bytes per lane and cost 5 us per launch (52.5 -> 57.0 us).
ADT_DEVICE_INLINE TT tt_load_if(const float* row, bool valid, int g) {
  TT t;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) x = *reinterpret_cast<const float4*>(row + 16 * nt + 4 * g);
    t.v[nt] = f32x4{x.x, x.y, x.z, x.w};
  }
  return t;
}
ADT_DEVICE_INLINE void tt_store(float* row, const TT& t, bool valid, int g) {
  if (!valid) return;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<float4*>(row + 16 * nt + 4 * g) = make_float4(t.v[nt][0], t.v[nt][1], t.v[nt][2], t.v[nt][3]);
}

// Tensors SAVED for the backward as bf16 rows use the REGISTER ORDER of a transposed tile: feature f = 16 nt + 4 g + r sits at element
// 16 g + 4 nt + r of its 64-element row (adt_saved_pos), so a lane's sixteen values are 32 contiguous bytes -- two 16-byte accesses, a
// token's row 128 contiguous bytes.  In natural order a lane wrote four 8-byte pieces and the write counter showed 1.6x the bytes
// stored (profiles/r02_dec_fwd_pmc.json).  Every reader of these buffers goes through tt_load_bf16 / rows_load_saved / the bf16 staging
// of k_seq_attn_bwd, which undo the permutation.
ADT_DEVICE_INLINE void tt_store_bf16(__bf16* row, const TT& t, bool valid, int g) {
  if (!valid) return;
  *reinterpret_cast<bf16x8*>(row + 16 * g) = tt_pack(t.v[0], t.v[1]);
  *reinterpret_cast<bf16x8*>(row + 16 * g + 8) = tt_pack(t.v[2], t.v[3]);
}
ADT_DEVICE_INLINE TT tt_load_bf16(const __bf16* row, bool valid, int g) {
  TT t;
  typedef const bf16x8 __attribute__((address_space(1))) * gp8;       // see tt_load
  const gp8 p = valid ? (gp8)(row + 16 * g) : (gp8)(tt_zero_row + 8 * g);
  const bf16x8 lo = p[0], hi = p[1];
#pragma unroll
  for (int r = 0; r < 4; ++r) { t.v[0][r] = (float)lo[r]; t.v[1][r] = (float)lo[4 + r]; t.v[2][r] = (float)hi[r]; t.v[3][r] = (float)hi[4 + r]; }
  return t;
}

// a saved (B*L x 64) tensor in either format: fp32 rows, or bf16 rows in the first half of the same buffer
ADT_DEVICE_INLINE void tt_save(float* buf, size_t row, const TT& t, bool valid, int g, int as_bf16) {
  if (!buf) return;
  if (as_bf16) tt_store_bf16(reinterpret_cast<__bf16*>(buf) + row * 64, t, valid, g);
  else tt_store(buf + row * 64, t, valid, g);
}
// A saved tile in two steps: tt_saved_request issues the loads (raw registers, nothing consumes them), tt_saved_value converts where the
// tile is first needed.  tt_load_saved converts inside the `as_bf16` branch right behind its loads, and a use behind a load is a wait behind a
// load: in a prologue that requests ten tensors every such call was one more serial memory round trip.
typedef unsigned tt_u4 __attribute__((ext_vector_type(4)));
struct TTSaved { tt_u4 q[4]; };
ADT_DEVICE_INLINE TTSaved tt_saved_request(const float* buf, size_t row, bool valid, int g, int as_bf16) {
  typedef const tt_u4 __attribute__((address_space(1))) * gq;
  TTSaved r;
  if (as_bf16) {
    const gq p = valid ? (gq)(reinterpret_cast<const __bf16*>(buf) + row * 64 + 16 * g) : (gq)tt_zero_row;
    r.q[0] = p[0]; r.q[1] = p[1];
    r.q[2] = r.q[3] = tt_u4{0u, 0u, 0u, 0u};
  } else {
    const gq p = valid ? (gq)(buf + row * 64 + 4 * g) : (gq)(tt_zero_row + 4 * g);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) r.q[nt] = p[4 * nt];
  }
  return r;
}
ADT_DEVICE_INLINE TT tt_saved_value(TTSaved r, int as_bf16) {
  // opaque: keeps the conversion below from being hoisted back into the branch that loaded the registers
  uint32_t w0 = r.q[0].x, w1 = r.q[0].y, w2 = r.q[0].z, w3 = r.q[0].w, w4 = r.q[1].x, w5 = r.q[1].y, w6 = r.q[1].z, w7 = r.q[1].w;
  asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7));
  TT t;
  if (as_bf16) {
    union { tt_u4 u; bf16x8 b; } lo, hi;
    lo.u = tt_u4{w0, w1, w2, w3}; hi.u = tt_u4{w4, w5, w6, w7};
#pragma unroll
    for (int e = 0; e < 4; ++e) { t.v[0][e] = (float)lo.b[e]; t.v[1][e] = (float)lo.b[4 + e]; t.v[2][e] = (float)hi.b[e]; t.v[3][e] = (float)hi.b[4 + e]; }
  } else {
    t.v[0] = f32x4{__uint_as_float(w0), __uint_as_float(w1), __uint_as_float(w2), __uint_as_float(w3)};
    t.v[1] = f32x4{__uint_as_float(w4), __uint_as_float(w5), __uint_as_float(w6), __uint_as_float(w7)};
#pragma unroll
    for (int nt = 2; nt < 4; ++nt)
      t.v[nt] = f32x4{__uint_as_float(r.q[nt].x), __uint_as_float(r.q[nt].y), __uint_as_float(r.q[nt].z), __uint_as_float(r.q[nt].w)};
  }
  return t;
}
ADT_DEVICE_INLINE TT tt_load_saved_if(const float* buf, size_t row, bool valid, int g, int as_bf16) {      // branching form, see tt_load_if
  if (as_bf16) {
    TT t = tt_zero();
    if (valid) {
      const __bf16* r = reinterpret_cast<const __bf16*>(buf) + row * 64;
      const bf16x8 lo = *reinterpret_cast<const bf16x8*>(r + 16 * g), hi = *reinterpret_cast<const bf16x8*>(r + 16 * g + 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { t.v[0][e] = (float)lo[e]; t.v[1][e] = (float)lo[4 + e]; t.v[2][e] = (float)hi[e]; t.v[3][e] = (float)hi[4 + e]; }
    }
    return t;
  }
  return tt_load_if(buf + row * 64, valid, g);
}
ADT_DEVICE_INLINE TT tt_load_saved(const float* buf, size_t row, bool valid, int g, int as_bf16) {
  if (as_bf16) return tt_load_bf16(reinterpret_cast<const __bf16*>(buf) + row * 64, valid, g);
  return tt_load(buf + row * 64, valid, g);
}

// Exchanges between the four lanes (g = 0..3) that hold one token column.  v_permlane16_swap / v_permlane32_swap (gfx950) stay in
// the vector ALU; __shfl_xor compiles to ds_bpermute_b32, a round trip through the LDS crossbar on the critical path of every
// LayerNorm, softmax and classifier reduction.  With both operands the same register, the swap leaves lane ^ 16 (lane ^ 32) of the
// value in one of the two results: rows of `a` that were swapped in hold the partner's value, the others find it in `b`.
typedef unsigned tt_u2 __attribute__((ext_vector_type(2)));
ADT_DEVICE_INLINE unsigned tt_xor16u(unsigned u) {
  const tt_u2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return (threadIdx.x & 16) ? r[0] : r[1];
}
ADT_DEVICE_INLINE unsigned tt_xor32u(unsigned u) {
  const tt_u2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return (threadIdx.x & 32) ? r[0] : r[1];
}
ADT_DEVICE_INLINE float tt_xor16(float v) { return __uint_as_float(tt_xor16u(__float_as_uint(v))); }
ADT_DEVICE_INLINE float tt_xor32(float v) { return __uint_as_float(tt_xor32u(__float_as_uint(v))); }

// feature reductions: every lane of a token column ends up with the column's total
ADT_DEVICE_INLINE float tt_colsum(float v) {
  v += tt_xor16(v);
  v += tt_xor32(v);
  return v;
}
ADT_DEVICE_INLINE float tt_colmax(float v) {
  v = fmaxf(v, tt_xor16(v));
  v = fmaxf(v, tt_xor32(v));
  return v;
}
ADT_DEVICE_INLINE uint32_t tt_color(uint32_t v) {
  v |= tt_xor16u(v);
  v |= tt_xor32u(v);
  return v;
}

// sum over the 16 lanes of a row (the 16 tokens of a tile, for one g): every lane ends up with the row total.  Four DPP steps in the
// vector ALU (quad butterflies, then the half-row and row mirrors).
ADT_DEVICE_INLINE float tt_rowsum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

// Small per-feature vectors (biases, LayerNorm gamma / beta, classifier weights) are copied to LDS once per workgroup: as global
// loads they sat on the critical path of every tile (one dependent L2 round trip, ~800 cycles, per bias add / LayerNorm).
template <int NTHREADS>
ADT_DEVICE_INLINE void tt_stage_vec(float* dst, const float* src, int n) {
  if (src == nullptr) return;
  for (int i = threadIdx.x; i < n; i += NTHREADS) dst[i] = src[i];
}

// LayerNorm over the 64 features of each token (eps inside the square root, as torch.nn.LayerNorm); returns gamma * xhat + beta
ADT_DEVICE_INLINE TT tt_layernorm(const TT& x, const float* gamma, const float* beta, float eps, int g) {
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) s += (x.v[nt][0] + x.v[nt][1]) + (x.v[nt][2] + x.v[nt][3]);
  const float mu = tt_colsum(s) * (1.0f / 64);
  TT d;
  float q = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float t = x.v[nt][r] - mu; d.v[nt][r] = t; q += t * t; }
  const float rstd = 1.0f / sqrtf(tt_colsum(q) * (1.0f / 64) + eps);
  const TTV gm = tt_vec(gamma, g), bt = tt_vec(beta, g);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) d.v[nt][r] = d.v[nt][r] * rstd * gm.v[nt][r] + bt.v[nt][r];
  return d;
}

// inverted dropout on a (rows x 64) tensor: element (token row, feature f) has index row * 64 + f; one hash per register quad
ADT_DEVICE_INLINE void tt_dropout(TT& t, uint32_t key, const DropCfg& d, uint32_t row_global, int g) {
  if (!d.thr) return;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const uint32_t bits = adt_keep4(key, row_global * 64u + (uint32_t)(16 * nt + 4 * g), d.thr);
#pragma unroll
    for (int r = 0; r < 4; ++r) t.v[nt][r] = ((bits >> r) & 1u) ? t.v[nt][r] * d.scale : 0.f;
  }
}

// ---- LDS images written from TT tiles ---------------------------------------------------------------------------------------
// slot-ordered [token][64] image (the A operand of products that contract over FEATURES, e.g. the keys of S^T = K Q^T):
// feature 16 nt + 4 g + r sits at 32 (nt >> 1) + 8 g + 4 (nt & 1) + r, so one ds_read_b128 at [row][32 kb + 8 g] is a whole fragment
ADT_DEVICE_INLINE void tt_put_slot(__bf16* img, int token, const TT& t, bool valid, int g) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    bf16x4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = (__bf16)(valid ? t.v[nt][r] : 0.f);
    *reinterpret_cast<bf16x4*>(img + token * TT_RS + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1)) = b;
  }
}
// natural-order [token][64] image (read through ds_read_b64_tr_b16 by products that contract over TOKENS)
ADT_DEVICE_INLINE void tt_put_rows(__bf16* img, int token, const TT& t, bool valid, int g) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    bf16x4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = (__bf16)(valid ? t.v[nt][r] : 0.f);
    *reinterpret_cast<bf16x4*>(img + token * TT_RS + 16 * nt + 4 * g) = b;
  }
}

// A-operand fragment with the FEATURE on the lane and 8 TOKENS in slot order (rows row0 + 4g .. +3 and row0 + 16 + 4g .. +3) from a
// natural-order [token][64] image: two hardware-transposed reads.  Lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3
// of its group's 4 x 16 block and receives column (c) of the four rows.  EXEC must be full (no divergence around this call).
typedef short tt_s4 __attribute__((ext_vector_type(4)));
ADT_DEVICE_INLINE bf16x8 tt_trfrag(const __bf16* img, int row0, int col0, int c, int g) {
  const __bf16* p = img + (row0 + 4 * g + (c >> 2)) * TT_RS + col0 + 4 * (c & 3);
  const tt_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tt_s4 __attribute__((address_space(3)))*)(p));
  const tt_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tt_s4 __attribute__((address_space(3)))*)(p + 16 * TT_RS));
  union { struct { tt_s4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// ---- causal attention of one query tile, one head ----------------------------------------------------------------------------
// sK: slot-ordered key image, sV: natural-order value image (both [LP][TT_RS], rows >= L zero).  fq: the head's query operand(s),
// pre-multiplied by log2(e) / sqrt(hd).  One sweep: all score tiles stay in registers (the workgroup owns a CU: 256 VGPRs per wave),
// so the MFMAs of the sweep are independent and issue back to back, and the exponentials form one long independent stream.
// Output: o[nt] = O^T rows (features 16 nt + 4g + r of this head), column = query c; log-sum-exp and dropout keep bits to HBM.
// INLINE ASM AND MFMA RESULTS.  On gfx950 the wait states between a v_mfma and a vector instruction that reads its result are the
// COMPILER's job, and its hazard recognizer does not look inside asm statements: an asm that consumes an accumulator register straight
// from an MFMA reads it before the matrix pipe has written it -- silently, and differently from run to run (found by the data-parallel
// determinism test: a v_max3_f32 asm on raw score registers made the forward differ by 2e-3 between two identical launches).  Every asm
// helper below therefore takes only values that an ordinary (compiler-visible) vector instruction has produced.
//
// max of a score quad into a running maximum: the two pair maxima are plain fmaxf on the MFMA results (compiler-managed), v_max3_f32
// folds them in (hipcc does not form v_max3 from nested fmaxf here): 3 instructions per quad instead of 4
ADT_DEVICE_INLINE float tt_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
ADT_DEVICE_INLINE float tt_max_quad(float m, const f32x4& acc) { return tt_max3(m, fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])); }
// x where bit POS of k is set, else +0.0: v_bfe_i32 (all-ones / zero) + v_and_b32 on the fp32 bits.  One asm statement: as C++ hipcc turns
// `x & sext(bit)` back into v_and + v_cmp + v_cndmask (and a VCC wait state), and between two asm statements it pads the dependent pair with s_nop.
// x must come from an ordinary vector instruction (here: v_exp_f32 -- the v_bfe in front of the v_and is the one instruction a reader of a
// transcendental result must stay behind), never straight from an MFMA: see above
template <int POS>
ADT_DEVICE_INLINE float tt_keep_if_bit(float x, uint32_t k) {
  float r;
  asm("v_bfe_i32 %0, %1, %3, 1\n\tv_and_b32 %0, %0, %2" : "=&v"(r) : "v"(k), "v"(x), "n"(POS));
  return r;
}
template <int HD, int MAXKT>
ADT_DEVICE_INLINE void tt_attn_tile(const __bf16* sK, const __bf16* sV, const bf16x8* fq, int kcol, int vcol, int qt, int L, int bh,
                                    uint32_t bh_rng, const DropCfg& drop, uint32_t key_rng, float* lse, uint32_t* mask, int lane,
                                    int c, int g, f32x4 (&o)[HD / 16]) {
  constexpr int NT = HD / 16, KB = (HD + 31) / 32;
  const int q = qt * 16 + c;
  const int nkt = qt + 1;
  f32x4 s[MAXKT];
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        acc = mfma_bf16(acc, *reinterpret_cast<const bf16x8*>(sK + (kt * 16 + c) * TT_RS + kcol + 32 * kb + 8 * g), fq[kb]);
      if (kt == qt) {              // the diagonal tile is the only one the causal mask cuts (every other key < every query of the tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (4 * g + r <= c) ? acc[r] : -INFINITY;
      }
      s[kt] = acc;
    } else {
      s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) m = tt_max_quad(m, s[kt]);
  m = tt_colmax(m);
  // dropout: element (query q, key j) has index idx_q + j with idx_q a multiple of 4 (L % 4 == 0 on this path), so the register quad of
  // keys 16 kt + 4 g .. + 3 shares the hash word (idx_q >> 2) + 4 kt + g.  Per quad: one hash, three bit operations for the four keep
  // flags (adt_keep7); per element: a sign-extending bit-field extract and an AND on the fp32 bits -- no compare / select pairs and no
  // multiply (the 1 / (1 - p) factor goes into the final normalisation).
  const uint32_t word_q = ((bh_rng * (uint32_t)L + (uint32_t)q) * (uint32_t)L >> 2) + (uint32_t)g;
  uint32_t clo = 0u, chi = 0u;
  if (drop.thr) adt_keep7_consts(drop.thr, clo, chi);
  float sum = 0.f;
  uint32_t mw[MAXKT / 2];
#pragma unroll
  for (int i = 0; i < MAXKT / 2; ++i) mw[i] = 0u;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kp = 0; kp < MAXKT / 2; ++kp) {
    if (2 * kp < nkt) {
      uint32_t k7[2] = {0x80808080u, 0x80808080u};
      if (drop.thr) {            // wave-uniform; the second tile of the last pair may lie beyond the causal range: its scores are -inf
#pragma unroll
        for (int t = 0; t < 2; ++t) k7[t] = adt_keep7(adt_hash32((word_q + (uint32_t)(4 * (2 * kp + t))) ^ key_rng), clo, chi);
        mw[kp] = (adt_keep7_nibble(k7[0]) | (adt_keep7_nibble(k7[1]) << 16)) << (4 * g);
      }
      float pe[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[2 * kp + t][r] - m);      // exp2(-inf) = 0 for masked / absent keys
          sum += e;
          pe[t][r] = e;
        }
#define TT_KEEP(r) pe[t][r] = tt_keep_if_bit<8 * (r) + 7>(pe[t][r], k7[t])
        TT_KEEP(0); TT_KEEP(1); TT_KEEP(2); TT_KEEP(3);
#undef TT_KEEP
      }
      const bf16x8 fp = tt_pack(f32x4{pe[0][0], pe[0][1], pe[0][2], pe[0][3]}, f32x4{pe[1][0], pe[1][1], pe[1][2], pe[1][3]});
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] = mfma_bf16(o[nt], tt_trfrag(sV, kp * 32, vcol + nt * 16, c, g), fp);
    }
  }
  sum = tt_colsum(sum);
  if (g == 0 && q < L) lse[(size_t)bh * L + q] = (m + __builtin_amdgcn_logf(sum)) * 0.6931471805599453f;   // natural log of sum exp(score)
  const float inv = drop.scale / sum;          // drop.scale = 1 without dropout
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] *= inv;
  if (mask && drop.thr) {
    uint32_t ow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ow[i] = i < MAXKT / 2 ? tt_color(mw[i]) : 0u;
    if (g == 0 && q < L) {
      uint4* dst = reinterpret_cast<uint4*>(mask + ((size_t)bh * L + q) * 8);
      dst[0] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      dst[1] = make_uint4(ow[4], ow[5], ow[6], ow[7]);
    }
  }
}

// The same attention as two ROLLED sweeps over the key tiles, ALL heads of the query tile per iteration (the default; -DADT_ATTN_UNROLLED
// restores the form above): sweep 1 takes the row maxima, sweep 2 recomputes the scores of a pair of key tiles (KB more MFMAs per tile and
// head: the matrix pipe is idle most of the time here), exponentiates and multiplies by V.  The unrolled form keeps all 14 score tiles in
// registers and is ~100 KB of straight-line code per kernel, every instruction executed once per wave: with twelve waves at different
// places of it, instruction fetch and each wave's own read -> MFMA -> exp -> MFMA latency chain paced the phase, not the vector ALU
// (profiles/r03_stamps_fwd12.txt: 14k cycles for a wave whose instruction stream issues in 7k).  Here the loop body is ~150 instructions and the
// H heads are independent chains inside it.  Keep-bit words go to HBM per key pair (words of pairs beyond the causal range are not
// written: no reader looks at them).  Returns the TT tile (features x queries) of all heads.
template <int HD>
ADT_DEVICE_INLINE TT tt_attn_heads_loop(const __bf16* sK, const __bf16* sV, const bf16x8* fq, int qt, int L, int b, uint32_t b_offset, const DropCfg& drop,
                                        uint32_t key_rng, float* lse, uint32_t* mask, int c, int g) {
  constexpr int H = 64 / HD, NT = HD / 16, KB = (HD + 31) / 32;
  const int q = qt * 16 + c;
  const __bf16* kbase = sK + c * TT_RS + 8 * g;                 // key row 16 kt + c of head h: + kt * 16 * TT_RS + kcol(h)
  float m[H];
#pragma unroll
  for (int h = 0; h < H; ++h) m[h] = -INFINITY;
#pragma unroll 2
  for (int kt = 0; kt < qt; ++kt) {                              // tiles below the diagonal: every key precedes every query of the tile
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int kcol = HD == 16 ? 32 * (h >> 1) : h * HD;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) acc = mfma_bf16(acc, *reinterpret_cast<const bf16x8*>(kbase + kt * 16 * TT_RS + kcol + 32 * kb), fq[h * KB + kb]);
      m[h] = tt_max_quad(m[h], acc);
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int kcol = HD == 16 ? 32 * (h >> 1) : h * HD;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc = mfma_bf16(acc, *reinterpret_cast<const bf16x8*>(kbase + qt * 16 * TT_RS + kcol + 32 * kb), fq[h * KB + kb]);
#pragma unroll
    for (int r = 0; r < 4; ++r) m[h] = fmaxf(m[h], (4 * g + r <= c) ? acc[r] : -INFINITY);
    m[h] = tt_colmax(m[h]);
  }
  // dropout: see tt_attn_tile.  Head h of sequence b is row (b * H + h) of the (B H, L, L) probability tensor
  const uint32_t bh0 = (uint32_t)(b * H) + b_offset * (uint32_t)H;
  uint32_t word_q[H];
#pragma unroll
  for (int h = 0; h < H; ++h) word_q[h] = (((bh0 + (uint32_t)h) * (uint32_t)L + (uint32_t)q) * (uint32_t)L >> 2) + (uint32_t)g;
  uint32_t clo = 0u, chi = 0u;
  if (drop.thr) adt_keep7_consts(drop.thr, clo, chi);
  uint32_t* mrow = (mask && drop.thr && q < L) ? mask + ((size_t)(b * H) * L + q) * 8 : nullptr;      // head h: + h * L * 8
  float sum[H];
  TT o;
#pragma unroll
  for (int h = 0; h < H; ++h) sum[h] = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) o.v[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int npair = (qt + 2) >> 1;
#pragma unroll 1
  for (int kp = 0; kp < npair; ++kp) {
    const bool edge = 2 * kp + 1 >= qt;      // wave-uniform: the last pair holds the diagonal tile and, for even qt, a tile beyond it
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int kcol = HD == 16 ? 32 * (h >> 1) : h * HD;
      f32x4 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          s[t] = mfma_bf16(s[t], *reinterpret_cast<const bf16x8*>(kbase + (2 * kp + t) * 16 * TT_RS + kcol + 32 * kb), fq[h * KB + kb]);
      }
      if (edge) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * kp + t;
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][r] = (kt < qt || (kt == qt && 4 * g + r <= c)) ? s[t][r] : -INFINITY;
        }
      }
      uint32_t k7[2] = {0x80808080u, 0x80808080u};
      if (drop.thr) {
#pragma unroll
        for (int t = 0; t < 2; ++t) k7[t] = adt_keep7(adt_hash32((word_q[h] + (uint32_t)(4 * (2 * kp + t))) ^ key_rng), clo, chi);
        const uint32_t ow = tt_color((adt_keep7_nibble(k7[0]) | (adt_keep7_nibble(k7[1]) << 16)) << (4 * g));
        if (g == 0 && mrow) mrow[(size_t)h * L * 8 + kp] = ow;
      }
      float pe[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[t][r] - m[h]);      // exp2(-inf) = 0 for masked keys
          sum[h] += e;
          pe[t][r] = e;
        }
#define TT_KEEP(r) pe[t][r] = tt_keep_if_bit<8 * (r) + 7>(pe[t][r], k7[t])
        TT_KEEP(0); TT_KEEP(1); TT_KEEP(2); TT_KEEP(3);
#undef TT_KEEP
      }
      const bf16x8 fp = tt_pack(f32x4{pe[0][0], pe[0][1], pe[0][2], pe[0][3]}, f32x4{pe[1][0], pe[1][1], pe[1][2], pe[1][3]});
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o.v[h * NT + nt] = mfma_bf16(o.v[h * NT + nt], tt_trfrag(sV, kp * 32, h * HD + nt * 16, c, g), fp);
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float st = tt_colsum(sum[h]);
    if (g == 0 && q < L) lse[(size_t)(b * H + h) * L + q] = (m[h] + __builtin_amdgcn_logf(st)) * 0.6931471805599453f;   // natural log of sum exp(score)
    const float inv = drop.scale / st;          // drop.scale = 1 without dropout
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o.v[h * NT + nt] *= inv;
  }
  return o;
}

// query operands of every head from a TT q tile (scaled): fq[h * KB + kb]
template <int HD>
ADT_DEVICE_INLINE void tt_qfrags(const TT& q, float mul, bf16x8 (&fq)[(64 / HD) * ((HD + 31) / 32)]) {
  if constexpr (HD == 64) {
    fq[0] = tt_pack(q.v[0], q.v[1], mul);
    fq[1] = tt_pack(q.v[2], q.v[3], mul);
  } else if constexpr (HD == 32) {
    fq[0] = tt_pack(q.v[0], q.v[1], mul);
    fq[1] = tt_pack(q.v[2], q.v[3], mul);
  } else {          // HD == 16: head h is tile h; it occupies the low (even h) or high (odd h) four slots of its 32-feature block
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 4; ++h) fq[h] = (h & 1) ? tt_pack(z, q.v[h], mul) : tt_pack(q.v[h], z, mul);
  }
}

// attention of every head for one query tile -> a TT tile (features x queries)
template <int HD, int MAXKT>
ADT_DEVICE_INLINE TT tt_attn_heads(const __bf16* sK, const __bf16* sV, const bf16x8* fq, int tile, int L, int b, uint32_t b_offset,
                                   DropCfg drop, uint32_t site, uint32_t seedv, float* lse, uint32_t* mask, int lane, int c, int g) {
  constexpr int H = 64 / HD, NT = HD / 16, KB = (HD + 31) / 32;
  const uint32_t key_rng = drop.thr ? adt_site_key(seedv, site) : 0u;
#ifndef ADT_ATTN_UNROLLED
  return tt_attn_heads_loop<HD>(sK, sV, fq, tile, L, b, b_offset, drop, key_rng, lse, mask, c, g);
#else
  TT o;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    f32x4 oh[NT];
    const int bh = b * H + h;
    const int kcol = HD == 16 ? 32 * (h >> 1) : h * HD;      // start of the 32-feature slot block(s) that hold this head's keys
    tt_attn_tile<HD, MAXKT>(sK, sV, fq + h * KB, kcol, h * HD, tile, L, bh, (uint32_t)bh + b_offset * (uint32_t)H, drop, key_rng, lse, mask,
                            lane, c, g, oh);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o.v[h * NT + nt] = oh[nt];
  }
  return o;
#endif
}

}  // namespace adt
