// C ABI (include/adt_hip.h, "wide" section): launch wrappers for the general dense layers, the masked attention and
// the row kernels used by the BERT4Rec-ADT and STOSA-ADT paths.  Host code only enqueues work on the caller's stream.
#include <cstdlib>
#include "adt_host.h"

#include "adt_attn_gen.cuh"
#include "adt_gemm.cuh"
#include "adt_dense_rows.cuh"
#include "adt_stosa.cuh"
#include "adt_wattn_mfma.cuh"
#include "adt_wide.cuh"

using namespace adt;

static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return adt_set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

static int grid_for(size_t work_items, int per_block, int cap) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return (int)g;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- masked attention dispatch ---------------------------------------------------------------------------------
template <int PREC, int HD, int MAXKT, bool CSK = false>
static int launch_attn_gen(bool bwd, const AttnGenArgs& a, hipStream_t s) {
  // bf16 operands: the forward (<= 106 VGPRs at every head size) and the backward at head size <= 64 (<= 124) fit the 128-register budget of
  // 16 waves = 4 per SIMD: one query / key tile per wave at L = 200 instead of two, and twice the waves to cover each other's LDS and MFMA latency
  constexpr int NWF = PREC == PREC_BF16 ? 16 : 8, NWB = (PREC == PREC_BF16 && HD <= 64) ? 16 : 8;
  const size_t smem = bwd ? AttnGenLds<PREC, HD, MAXKT>::bwd_bytes : AttnGenLds<PREC, HD, MAXKT>::fwd_bytes;
  if (smem > 160 * 1024) return adt_set_error("masked attention: L=%d hd=%d prec=%d needs %zu B of LDS (> 160 KB)", a.a.L, HD, PREC, smem);
  const void* fn = bwd ? (const void*)k_attn_gen_bwd<PREC, HD, MAXKT, NWB> : (const void*)k_attn_gen_fwd<PREC, HD, MAXKT, NWF, CSK>;
  static bool done[2] = {false, false};
  if (!done[bwd ? 1 : 0]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return adt_set_error("masked attention: hipFuncSetAttribute(%zu)", smem);
    done[bwd ? 1 : 0] = true;
  }
  AttnGenArgs args = a;
  void* kargs[] = {&args};
  if (hipLaunchKernel(fn, dim3(a.a.B * a.a.H), dim3((bwd ? NWB : NWF) * 64), kargs, smem, s) != hipSuccess) return adt_set_error("masked attention: launch failed");
  return check_launch(bwd ? "attn_masked_bwd" : "attn_masked_fwd");
}

// backward staged in NCH chunks of the sequence (k_attn_gen_bwd_chunked): hd = 128 always, and hd = 64 in the exact-fp32 mode at
// L > 128, whose whole-(b, h) images (fp32: 248 KB at L = 200) do not fit the 160 KB of LDS
template <int PREC, int HD, int MAXKT, int NCH>
static int launch_attn_gen_bwd_chunked(const AttnGenArgs& a, hipStream_t s) {
  constexpr int NW = 8;
  const size_t smem = AttnChunkLds<PREC, HD, MAXKT, NCH>::bwd_bytes;
  if (smem > 160 * 1024) return adt_set_error("masked attention bwd: L=%d hd=%d prec=%d needs %zu B of LDS (> 160 KB)", a.a.L, HD, PREC, smem);
  const void* fn = (const void*)k_attn_gen_bwd_chunked<PREC, HD, MAXKT, NCH, NW>;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return adt_set_error("masked attention: hipFuncSetAttribute(%zu)", smem);
    done = true;
  }
  AttnGenArgs args = a;
  void* kargs[] = {&args};
  if (hipLaunchKernel(fn, dim3(a.a.B * a.a.H), dim3(NW * 64), kargs, smem, s) != hipSuccess) return adt_set_error("masked attention: launch failed");
  return check_launch("attn_masked_bwd(chunked)");
}

template <int PREC, int HD>
static int dispatch_attn_gen_l(bool bwd, const AttnGenArgs& a, hipStream_t s) {
  if (a.a.L <= 64) return launch_attn_gen<PREC, HD, 4>(bwd, a, s);
  if (a.a.L <= 128) return launch_attn_gen<PREC, HD, 8>(bwd, a, s);
  if (a.a.L <= 224) {
    if constexpr (PREC == PREC_F32 && HD == 64) {
      if (bwd) return launch_attn_gen_bwd_chunked<PREC, HD, 16, 2>(a, s);
    }
    return launch_attn_gen<PREC, HD, 14>(bwd, a, s);
  }
  return adt_set_error("masked attention: L=%d > 224 unsupported", a.a.L);
}

// hd = 128 (sasrec d = 256, H = 2): forward with the whole (b, h) resident, backward staged in NCH chunks
template <int PREC, int MAXKT, int NCH>
static int launch_attn_gen_128(bool bwd, const AttnGenArgs& a, hipStream_t s) {
  constexpr int HD = 128;
  if (!bwd) {     // causal without key padding (the d = 256 SASRec template): skip the key tiles above the diagonal
    const bool csk = a.a.causal && a.kid == nullptr && a.fill <= -1e9f;
    return csk ? launch_attn_gen<PREC, HD, MAXKT, true>(false, a, s) : launch_attn_gen<PREC, HD, MAXKT, false>(false, a, s);
  }
  return launch_attn_gen_bwd_chunked<PREC, HD, MAXKT, NCH>(a, s);
}

template <int PREC>
static int dispatch_attn_gen(bool bwd, int hd, const AttnGenArgs& a, hipStream_t s) {
  if (hd == 16) return dispatch_attn_gen_l<PREC, 16>(bwd, a, s);
  if (hd == 32) return dispatch_attn_gen_l<PREC, 32>(bwd, a, s);
  if (hd == 64) return dispatch_attn_gen_l<PREC, 64>(bwd, a, s);
  if (hd == 128) {
    if (a.a.L <= 64) return launch_attn_gen_128<PREC, 4, 1>(bwd, a, s);
    if (a.a.L <= 256) return launch_attn_gen_128<PREC, 16, 2>(bwd, a, s);
    return adt_set_error("masked attention: L=%d > 256 unsupported at head_dim 128", a.a.L);
  }
  return adt_set_error("masked attention: head_dim=%d unsupported (16/32/64/128)", hd);
}

template <class K>
static int gemm_launch(K kernel, size_t smem, int grid, hipStream_t s, const void* args_ptr, bool& attr_done) {
  if (!attr_done) {
    if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return adt_set_error("dense: hipFuncSetAttribute(%zu)", smem);
    attr_done = true;
  }
  void* kargs[] = {const_cast<void*>(args_ptr)};
  if (hipLaunchKernel((const void*)kernel, dim3(grid), dim3(GTH), kargs, smem, s) != hipSuccess) return adt_set_error("dense: launch failed");
  return 0;
}


// ---- row-streaming kernels (adt_dense_rows.cuh): bf16 operands, contraction 64 / 128 / 256 ---------------------------------------
static float* g_dense_ws = nullptr;      // scratch registered by the host (adt_dense_workspace): private partials of the 256 x 256 weight gradients
static int64_t g_dense_ws_bytes = 0;
static int g_rows_enabled = 1;
// ADT_STAGE256=0 in the environment keeps the 256-wide forward / input-gradient stage kernels (k_dense_fwd256, k_dense_dx256) off: A/B runs
static bool stage_kernels_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_STAGE256"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on != 0;
}      // adt_dense_rows_enable(0) routes everything to the tiled kernels (A/B measurements, tests)

template <class KFn, class Args>
static int rows_launch(KFn kernel, const Args& a, int n_panels, int pc, int contraction, int T, hipStream_t s, bool& attr_done) {
  const size_t smem = rows_lds_bytes(contraction, pc);
  if (!attr_done) {
    if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return adt_set_error("dense rows: hipFuncSetAttribute(%zu)", smem);
    attr_done = true;
  }
  // one workgroup per CU and panel group; never more row groups than 16-row tiles / waves
  const int ntiles = (T + 15) / 16;
  int nrg = 256 / n_panels;
  if (smem <= 64 * 1024) nrg *= 2;
  const int need = (ntiles + ROWS_NW - 1) / ROWS_NW;
  if (nrg > need) nrg = need;
  if (nrg < 1) nrg = 1;
  Args args = a;
  int np = n_panels, pcv = pc;
  void* kargs[] = {&args, &np, &pcv};
  if (hipLaunchKernel((const void*)kernel, dim3(nrg * n_panels), dim3(ROWS_NW * 64), kargs, smem, s) != hipSuccess) return adt_set_error("dense rows: launch failed");
  return 0;
}

static bool rows_fwd_ok(const DenseFwdArgs& a) {
  if (!g_rows_enabled) return false;
  if (!(a.K == 64 || a.K == 128 || a.K == 256) || (a.N % 4) || (a.ldy % 4) || !aligned16(a.Y)) return false;
  if (a.b && !aligned16(a.b)) return false;
  if (a.U && ((a.ldu % 4) || !aligned16(a.U))) return false;
  if (a.R && ((a.ldr % 4) || !aligned16(a.R))) return false;
  if (a.R2 && ((a.ldr2 % 4) || !aligned16(a.R2))) return false;
  return true;
}

static int launch_dense_fwd_rows(const DenseFwdArgs& a, hipStream_t s) {
  const int pc = a.N >= ROWS_PC ? ROWS_PC : (a.N + 15) / 16 * 16;
  const int n_panels = (a.N + pc - 1) / pc;
  static bool done[3] = {false, false, false};
  int rc;
  // residual loads run one chunk of CH column tiles ahead (double buffered in registers): 8 tiles, or 4 when there are two residuals
  if (a.R2) {
    static bool done4[3] = {false, false, false};
    if (a.K == 64) rc = rows_launch(k_dense_fwd_rows<2, 4>, a, n_panels, pc, 64, a.T, s, done4[0]);
    else if (a.K == 128) rc = rows_launch(k_dense_fwd_rows<4, 4>, a, n_panels, pc, 128, a.T, s, done4[1]);
    else rc = rows_launch(k_dense_fwd_rows<8, 4>, a, n_panels, pc, 256, a.T, s, done4[2]);
  } else if (a.K == 64) rc = rows_launch(k_dense_fwd_rows<2, 8>, a, n_panels, pc, 64, a.T, s, done[0]);
  else if (a.K == 128) rc = rows_launch(k_dense_fwd_rows<4, 8>, a, n_panels, pc, 128, a.T, s, done[1]);
  else rc = rows_launch(k_dense_fwd_rows<8, 8>, a, n_panels, pc, 256, a.T, s, done[2]);
  return rc ? rc : check_launch("dense_fwd(rows)");
}

static bool rows_dx_ok(const DenseBwdArgs& a) {
  if (!g_rows_enabled || !a.dX) return false;
  const int N = a.G.N;
  if ((N % 64) || N > 1024 || (a.K % 4) || (a.lddx % 4) || !aligned16(a.dX) || (a.G.lddy % 4) || !aligned16(a.G.dY)) return false;
  if (a.G.act != ACT_NONE && ((a.G.ldu % 4) || !aligned16(a.G.U))) return false;
  return true;
}

// contraction chunks of <= 256 columns of G / rows of W; every chunk after the first accumulates
static int launch_dense_dx_rows(const DenseBwdArgs& a0, hipStream_t s) {
  const int N = a0.G.N;
  static bool done[6] = {false, false, false, false, false, false};
  const bool has_u = a0.G.act != ACT_NONE;
  for (int n0 = 0; n0 < N;) {
    // with an activation the saved pre-activation rides along in registers: contraction chunks of 128 instead of 256
    int chunk = (N - n0 >= 256 && !has_u) ? 256 : (N - n0 >= 128 ? 128 : 64);
    DenseBwdArgs a = a0;
    a.G.dY = a0.G.dY + n0; a.G.U = a0.G.U ? a0.G.U + n0 : nullptr; a.G.N = chunk; a.G.idx_off = a0.G.idx_off + n0;
    a.W = a0.W + (size_t)n0 * a0.ldw;
    a.beta = (n0 > 0) ? 1 : a0.beta;
    a.dW = nullptr; a.db = nullptr;
    const int pc = a.K >= ROWS_PC ? ROWS_PC : (a.K + 15) / 16 * 16;
    const int n_panels = (a.K + pc - 1) / pc;
    int rc;
    if (has_u) {
      if (chunk == 64) rc = rows_launch(k_dense_dx_rows<2, true>, a, n_panels, pc, 64, a.G.T, s, done[3]);
      else rc = rows_launch(k_dense_dx_rows<4, true>, a, n_panels, pc, 128, a.G.T, s, done[4]);
    } else if (chunk == 64) rc = rows_launch(k_dense_dx_rows<2, false>, a, n_panels, pc, 64, a.G.T, s, done[0]);
    else if (chunk == 128) rc = rows_launch(k_dense_dx_rows<4, false>, a, n_panels, pc, 128, a.G.T, s, done[1]);
    else rc = rows_launch(k_dense_dx_rows<8, false>, a, n_panels, pc, 256, a.G.T, s, done[2]);
    if (rc) return rc;
    n0 += chunk;
  }
  return check_launch("dense_bwd_dx(rows)");
}

template <int PREC>
static int launch_dense_fwd(const DenseFwdArgs& a0, hipStream_t s) {
  if (PREC == PREC_BF16 && g_rows_enabled && a0.K == 256 && (a0.N % 256) == 0 && a0.N <= 1024 &&
      (a0.ldx % 4) == 0 && aligned16(a0.X) && (a0.ldw % 4) == 0 && aligned16(a0.W) && (a0.ldy % 4) == 0 && aligned16(a0.Y) &&
      (!a0.U || ((a0.ldu % 4) == 0 && aligned16(a0.U))) && (!a0.R || ((a0.ldr % 4) == 0 && aligned16(a0.R))) && (!a0.R2 || ((a0.ldr2 % 4) == 0 && aligned16(a0.R2))) &&
      stage_kernels_on()) {
    // K = 256, N = 256 .. 1024: weight rows in registers, activations through LDS, compile-time epilogue on rows (adt_gemm.cuh: k_dense_fwd256)
    const int T = a0.T;
    int nwg = (T + DWP_TS - 1) / DWP_TS;
    const int cap = a0.N > 256 ? 512 / (a0.N / 256) : 256;
    if (nwg > cap) nwg = cap;
    const int chunk = ((T + nwg - 1) / nwg + DWP_TS - 1) / DWP_TS * DWP_TS;
    const dim3 grid((T + chunk - 1) / chunk, a0.N / 256);
    const int epi = ((a0.R || a0.R2 || a0.ids) ? 1 : 0) | (a0.drop.thr ? 2 : 0) | ((a0.act != ACT_NONE || a0.U) ? 4 : 0);
    switch (epi) {
      case 0: hipLaunchKernelGGL(k_dense_fwd256<0>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
      case 1: hipLaunchKernelGGL(k_dense_fwd256<1>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
      case 2: hipLaunchKernelGGL(k_dense_fwd256<2>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
      case 3: hipLaunchKernelGGL(k_dense_fwd256<3>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
      case 4: hipLaunchKernelGGL(k_dense_fwd256<4>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
      case 5: hipLaunchKernelGGL(k_dense_fwd256<5>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
      case 6: hipLaunchKernelGGL(k_dense_fwd256<6>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
      default: hipLaunchKernelGGL(k_dense_fwd256<7>, grid, dim3(DWP_NTH), 0, s, a0, chunk); break;
    }
    return 0;
  }
  if (PREC == PREC_BF16 && rows_fwd_ok(a0)) return launch_dense_fwd_rows(a0, s);
  DenseFwdArgs a = a0;
  a.nt_n = a.N > 64 ? (a.N + 127) / 128 : 1;
  a.nt_m = (a.T + GBM - 1) / GBM;
  const int grid = xcd_grid(a.nt_m, a.nt_n);      // XCD-aware 1-D launch (xcd_tile)
  static bool done[2] = {false, false};
  if (a.N > 64) { if (gemm_launch(k_dense_fwd<PREC, 128>, gemm_lds_bytes<PREC, 128>(), grid, s, &a, done[0])) return -1; }
  else if (gemm_launch(k_dense_fwd<PREC, 64>, gemm_lds_bytes<PREC, 64>(), grid, s, &a, done[1])) return -1;
  return check_launch("dense_fwd");
}

template <int PREC>
static int launch_dense_bwd(const DenseBwdArgs& a0, hipStream_t s) {
  DenseBwdArgs a = a0;
  const int T = a.G.T, N = a.G.N, K = a.K;
  if (PREC == PREC_BF16 && g_rows_enabled && a.dX && (N % 256) == 0 && N <= 768 && K == 256 && (a.ldw % 4) == 0 && aligned16(a.W) &&
      (a.G.lddy % 4) == 0 && aligned16(a.G.dY) && (a.G.act == ACT_NONE || ((a.G.ldu % 4) == 0 && aligned16(a.G.U))) && stage_kernels_on()) {
    // contraction 256 / 512 / 768 into 256 columns: weight in registers, gradient tiles through LDS, transposed output (adt_gemm.cuh:
    // k_dense_dx256; wider outputs -- K = 1024 as four column blocks -- measured neutral against the row-streaming kernel and stay there)
    const int NB = N / 256, kblocks = K / 256;
    int nwg = (T + DWP_TS - 1) / DWP_TS;
    const int cap = 256 / (kblocks > 2 ? 2 : 1);
    if (nwg > cap) nwg = cap;
    const int chunk = ((T + nwg - 1) / nwg + DWP_TS - 1) / DWP_TS * DWP_TS;
    DenseBwdArgs d = a;
    d.t_chunk = chunk;
    const dim3 grid((T + chunk - 1) / chunk, kblocks);
    const size_t smem = (size_t)DWP_IMG * NB;
    if (NB == 1) hipLaunchKernelGGL(k_dense_dx256<1>, grid, dim3(DWP_NTH), smem, s, d);
    else if (NB == 2) hipLaunchKernelGGL(k_dense_dx256<2>, grid, dim3(DWP_NTH), smem, s, d);
    else hipLaunchKernelGGL(k_dense_dx256<3>, grid, dim3(DWP_NTH), smem, s, d);
    a.dX = nullptr;
  }
  if (PREC == PREC_BF16 && rows_dx_ok(a)) {
    if (launch_dense_dx_rows(a, s)) return -1;
    a.dX = nullptr;
  }
  if (a.dX) {
    const int gx = K > 64 ? (K + 127) / 128 : 1, gy = (T + GBM - 1) / GBM;
    // long contractions over few output tiles (the all-item logits: N = V + 100) are split over blockIdx.z
    int splits = 1;
    if (N >= 4096 && gx * gy < 1024) {
      splits = (2048 + gx * gy - 1) / (gx * gy);
      if (splits > 32) splits = 32;
      a.n_chunk = ((N + splits - 1) / splits + GBK - 1) / GBK * GBK;
      splits = (N + a.n_chunk - 1) / a.n_chunk;
      if (splits > 1 && !a.beta && adt::zero_rows_f32_async(a.dX, (size_t)a.lddx, K, (size_t)T, s)) return adt_set_error("dense_bwd: zero");
    }
    a.nt_a = gx; a.nt_b = gy; a.nt_z = splits;
    const int grid = xcd_grid(gy, gx * splits);
    static bool done[2] = {false, false};
    if (K > 64) { if (gemm_launch(k_dense_bwd_dx<PREC, 128>, gemm_lds_bytes<PREC, 128>(), grid, s, &a, done[0])) return -1; }
    else if (gemm_launch(k_dense_bwd_dx<PREC, 64>, gemm_lds_bytes<PREC, 64>(), grid, s, &a, done[1])) return -1;
  }
  if (a.dW && PREC == PREC_BF16 && g_rows_enabled && (N % 256) == 0 && (K % 256) == 0 && (N / 256) * (K / 256) <= 4 && g_dense_ws && (a.ldx % 4) == 0 &&
      aligned16(a.X) && (a.G.lddy % 4) == 0 && aligned16(a.G.dY)) {
    // the whole 256 x 256 product (of each 256 x 256 block) per workgroup, private partials in the registered workspace + a reduce
    // (adt_gemm.cuh: k_dense_dw256); 256 workgroups in all
    const int blocks = (N / 256) * (K / 256), kblocks = K / 256;
    int nwg = (T + DWP_TS - 1) / DWP_TS;
    if (nwg > 256 / blocks) nwg = 256 / blocks;
    if ((int64_t)nwg * blocks * 262144 <= g_dense_ws_bytes) {
      const int chunk = ((T + nwg - 1) / nwg + DWP_TS - 1) / DWP_TS * DWP_TS;
      nwg = (T + chunk - 1) / chunk;
      a.t_chunk = chunk;
      hipLaunchKernelGGL(k_dense_dw256, dim3(nwg, blocks), dim3(DWP_NTH), 0, s, a, g_dense_ws);
      const int per = nwg >= 128 ? 32 : 16;
      hipLaunchKernelGGL(k_dense_dw256_reduce, dim3(64, (nwg + per - 1) / per, blocks), dim3(256), 0, s, (const float*)g_dense_ws, nwg, per, a.dW, a.lddw, kblocks);
      a.dW = nullptr;
    }
  }
  // the 256 x 128-block kernel pays off from four output blocks on (N = 768: 127 us vs 225 us tiled); at two blocks (256 x 256)
  // its 32 K atomics per workgroup cost what the deeper stages save (84 us vs 77 us)
  if (a.dW && PREC == PREC_BF16 && g_rows_enabled && ((N + DW_BN - 1) / DW_BN) * ((K + DW_BK - 1) / DW_BK) >= 4 && (N % 4) == 0 && (K % 4) == 0 && (a.ldx % 4) == 0 &&
      aligned16(a.X) && (a.G.lddy % 4) == 0 && aligned16(a.G.dY) && (a.G.act == ACT_NONE || ((a.G.ldu % 4) == 0 && aligned16(a.G.U)))) {
    const int n_blocks = (N + DW_BN - 1) / DW_BN, k_blocks = (K + DW_BK - 1) / DW_BK, tiles = n_blocks * k_blocks;
    // one workgroup per CU (96 KB of LDS): 256 workgroups when the T chunks stay >= 4 stages, chunks are multiples of 64 rows
    int splits = tiles >= 256 ? 1 : 256 / tiles;
    int chunk = ((T + splits - 1) / splits + DW_TS - 1) / DW_TS * DW_TS;
    if (chunk < DW_TS) chunk = DW_TS;
    splits = (T + chunk - 1) / chunk;
    a.t_chunk = chunk;
    static bool done = false;
    if (!done) {
      if (hipFuncSetAttribute((const void*)k_dense_dw_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DW_LDS_BYTES) != hipSuccess)
        return adt_set_error("dense dw rows: hipFuncSetAttribute");
      done = true;
    }
    DenseBwdArgs args = a;
    args.nt_z = splits;
    int nb = n_blocks, kb = k_blocks;
    void* kargs[] = {&args, &nb, &kb};
    if (hipLaunchKernel((const void*)k_dense_dw_rows, dim3(xcd_grid(splits, tiles)), dim3(DW_NTH), kargs, DW_LDS_BYTES, s) != hipSuccess) return adt_set_error("dense dw rows: launch failed");
    a.dW = nullptr;
  }
  if (a.dW && PREC == PREC_BF16 && g_rows_enabled && (N % 64) == 0 && (K % 64) == 0 && (N / 64) * (K / 64) <= 4 && (a.ldx % 4) == 0 && aligned16(a.X) && (a.G.lddy % 4) == 0 && aligned16(a.G.dY)) {
    // 64 x 64 layers: 128-row stages, ~200 workgroups (adt_gemm.cuh: k_dense_dw64)
    int chunk = ((T + 255) / 256 + DW64_ROWS - 1) / DW64_ROWS * DW64_ROWS;
    if (chunk < DW64_ROWS) chunk = DW64_ROWS;
    a.t_chunk = chunk;
    const int nwg = (T + chunk - 1) / chunk, blocks = (N / 64) * (K / 64);
    // private partials in the registered workspace + an ordered sum (no float atomics) when there is one
    float* const part = (g_dense_ws && (int64_t)nwg * blocks * DW64_PART * 4 <= g_dense_ws_bytes) ? g_dense_ws : nullptr;
    hipLaunchKernelGGL(k_dense_dw64, dim3(nwg, blocks), dim3(DW64_NTH), 0, s, a, part);
    if (part)
      hipLaunchKernelGGL(k_dense_dw64_reduce, dim3(17, blocks), dim3(1024), 0, s, (const float*)part, nwg, a.t_dev, T, chunk, a.dW, a.lddw, K / 64, a.db);
    a.dW = nullptr;
  }
  if (a.dW) {
    const int bn = K > 64 ? 128 : 64;
    const int gx = (K + bn - 1) / bn, gy = (N + GBM - 1) / GBM;
    // split T so that enough workgroups are in flight while each still amortises the 16 K atomics of its flush over >= 10
    // k-steps; the targets are measured (tools/bench_dense.py at T = 51,200: 4 tiles 78 us @512 vs 98 @1024; 12 tiles 223 @2048
    // vs 247 @1024; 16 tiles 565 @1024 vs 602 @2048)
    const int tiles = gx * gy;
    const int target_wgs = tiles <= 4 ? 512 : (tiles <= 12 ? 2048 : 1024);
    int splits = (target_wgs + tiles - 1) / tiles;
    int chunk = ((T + splits - 1) / splits + GBK - 1) / GBK * GBK;
    if (chunk < GBK) chunk = GBK;
    splits = (T + chunk - 1) / chunk;
    a.t_chunk = chunk;
    a.nt_a = gx; a.nt_b = gy; a.nt_z = splits;
    const int grid = xcd_grid(splits, gx * gy);
    static bool done[2] = {false, false};
    if (bn == 128) { if (gemm_launch(k_dense_bwd_dw<PREC, 128>, gemm_lds_bytes<PREC, 128>(), grid, s, &a, done[0])) return -1; }
    else if (gemm_launch(k_dense_bwd_dw<PREC, 64>, gemm_lds_bytes<PREC, 64>(), grid, s, &a, done[1])) return -1;
  }
  return check_launch("dense_bwd");
}

// The same attention on the matrix cores (adt_wattn_mfma.cuh): hd 16 or 32, L <= 128; prec picks bf16 operands or the exact fp32 MFMA.
// Returns 1 when the shape is not covered (the caller uses adt_wattn_fwd / adt_wattn_bwd).
static bool wattn_mfma_ok(int L, int hd, const int* lds, int n) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_WATTN_MFMA"); on = (e && atoi(e) == 0) ? 0 : 1; }
  if (!on || (hd != 16 && hd != 32) || L > 128) return false;
  for (int i = 0; i < n; ++i)
    if (lds[i] % 4) return false;
  return true;
}

template <int PREC, int HD>
static int wattn_mfma_launch(bool bwd, const WAttnArgs& a, hipStream_t s) {
  const void* fn = bwd ? (const void*)k_wattn_mfma_bwd<PREC, HD> : (const void*)k_wattn_mfma_fwd<PREC, HD, 8>;
  const size_t smem = wattn_mfma_lds_bytes(a.L, HD, bwd);
  if (smem > 160 * 1024) return 1;
  static bool done[2] = {false, false};
  if (!done[bwd]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return adt_set_error("wattn_mfma: hipFuncSetAttribute");
    done[bwd] = true;
  }
  WAttnArgs args = a;
  void* kargs[] = {&args};
  if (hipLaunchKernel(fn, dim3(a.B * a.H), dim3(WM_NW * 64), kargs, smem, s) != hipSuccess) return adt_set_error("wattn_mfma: launch failed");
  return check_launch(bwd ? "wattn_mfma_bwd" : "wattn_mfma_fwd");
}

static int wattn_mfma_dispatch(int prec, bool bwd, const WAttnArgs& a, hipStream_t s) {
  if (prec == ADT_PREC_BF16) return a.hd == 16 ? wattn_mfma_launch<PREC_BF16, 16>(bwd, a, s) : wattn_mfma_launch<PREC_BF16, 32>(bwd, a, s);
  return a.hd == 16 ? wattn_mfma_launch<PREC_F32, 16>(bwd, a, s) : wattn_mfma_launch<PREC_F32, 32>(bwd, a, s);
}


extern "C" {

int adt_dense_workspace(void* ws, int64_t bytes) {
  g_dense_ws = static_cast<float*>(ws);
  g_dense_ws_bytes = ws ? bytes : 0;
  return 0;
}

int adt_dense_rows_enable(int on) {
  const int was = g_rows_enabled;
  g_rows_enabled = on ? 1 : 0;
  return was;
}

int adt_dense_fwd(int prec, const float* X, int ldx, const float* W, int ldw, const float* b, int T, int K, int N, int act, float* U,
                  int ldu, float p, const uint32_t* seed, uint32_t site, uint32_t row_offset, const float* R, int ldr,
                  const float* R2, int ldr2, const int32_t* mask_ids, float* Y, int ldy, const int32_t* t_dev, void* stream) {
  if (T <= 0 || K <= 0 || N <= 0) return adt_set_error("dense_fwd: empty shape");
  if ((ldx % 4) || (ldw % 4) || !aligned16(X) || !aligned16(W)) return adt_set_error("dense_fwd: operands must be 16-byte aligned with ld %% 4 == 0");
  if (act < 0 || act > ACT_ELU1) return adt_set_error("dense_fwd: act=%d", act);
  DenseFwdArgs a{};
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.b = b; a.T = T; a.K = K; a.N = N; a.Y = Y; a.ldy = ldy; a.U = U; a.ldu = ldu; a.act = act;
  a.drop = adt_make_drop(p, seed, site); a.row_offset = row_offset; a.R = R; a.ldr = ldr; a.R2 = R2; a.ldr2 = ldr2; a.ids = mask_ids; a.t_dev = t_dev;
  return prec == ADT_PREC_F32 ? launch_dense_fwd<PREC_F32>(a, (hipStream_t)stream) : launch_dense_fwd<PREC_BF16>(a, (hipStream_t)stream);
}

int adt_dense_bwd(int prec, const float* dY, int lddy, int T, int K, int N, const int32_t* mask_ids, float p, const uint32_t* seed,
                  uint32_t site, uint32_t row_offset, int act, const float* U, int ldu, const float* X, int ldx, const float* W, int ldw,
                  float* dX, int lddx, int beta, float* dW, int lddw, float* db, const int32_t* t_dev, void* stream) {
  if (T <= 0 || K <= 0 || N <= 0) return adt_set_error("dense_bwd: empty shape");
  if ((lddy % 4) || !aligned16(dY) || (dW && ((ldx % 4) || !aligned16(X))) || (dX && ((ldw % 4) || !aligned16(W))))
    return adt_set_error("dense_bwd: operands must be 16-byte aligned with ld %% 4 == 0");
  if (act != ACT_NONE && !U) return adt_set_error("dense_bwd: activation gradient needs the saved pre-activation U");
  DenseBwdArgs a{};
  a.G.dY = dY; a.G.lddy = lddy; a.G.T = T; a.G.N = N; a.G.U = U; a.G.ldu = ldu; a.G.act = act;
  a.G.drop = adt_make_drop(p, seed, site); a.G.row_offset = row_offset; a.G.ids = mask_ids; a.G.idx_ld = N; a.G.idx_off = 0;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.K = K; a.dX = dX; a.lddx = lddx; a.beta = beta; a.dW = dW; a.lddw = lddw; a.db = db; a.t_dev = t_dev;
  return prec == ADT_PREC_F32 ? launch_dense_bwd<PREC_F32>(a, (hipStream_t)stream) : launch_dense_bwd<PREC_BF16>(a, (hipStream_t)stream);
}

static int fill_attn(AttnGenArgs& g, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int B, int H, int L, int hd,
                     int causal, const int32_t* kid, float fill, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset) {
  if ((ldq % 4) || (ldk % 4) || (ldv % 4) || (hd % 8)) return adt_set_error("masked attention: ld %% 4, hd %% 8");
  g.a.Q = Q; g.a.ldq = ldq; g.a.K = K; g.a.ldk = ldk; g.a.V = V; g.a.ldv = ldv; g.a.B = B; g.a.H = H; g.a.L = L; g.a.causal = causal;
  g.a.scale = 1.0f / sqrtf((float)hd); g.a.drop = adt_make_drop(p, seed, site); g.a.bh_offset = b_offset * (uint32_t)H;
  g.kid = kid; g.fill = fill;
  return 0;
}

int adt_attn_masked_fwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int B, int H, int L, int hd,
                        int causal, const int32_t* key_ids, float fill, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset,
                        float* O, int ldo, float* LSE, void* stream) {
  AttnGenArgs g{};
  if (fill_attn(g, Q, ldq, K, ldk, V, ldv, B, H, L, hd, causal, key_ids, fill, p, seed, site, b_offset)) return -1;
  g.a.O = O; g.a.ldo = ldo; g.a.LSE = LSE;
  return prec == ADT_PREC_F32 ? dispatch_attn_gen<PREC_F32>(false, hd, g, (hipStream_t)stream) : dispatch_attn_gen<PREC_BF16>(false, hd, g, (hipStream_t)stream);
}

int adt_attn_masked_bwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                        const float* LSE, const float* dO, int lddo, int B, int H, int L, int hd, int causal, const int32_t* key_ids,
                        float fill, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset, float* dQ, int lddq, float* dK,
                        int lddk, float* dV, int lddv, void* stream) {
  AttnGenArgs g{};
  if (fill_attn(g, Q, ldq, K, ldk, V, ldv, B, H, L, hd, causal, key_ids, fill, p, seed, site, b_offset)) return -1;
  if ((ldo % 4) || (lddo % 4) || (lddq % 4) || (lddk % 4) || (lddv % 4)) return adt_set_error("masked attention bwd: ld %% 4");
  g.a.O = const_cast<float*>(O); g.a.ldo = ldo; g.a.LSE = const_cast<float*>(LSE); g.a.dO = dO; g.a.lddo = lddo;
  g.a.dQ = dQ; g.a.lddq = lddq; g.a.dK = dK; g.a.lddk = lddk; g.a.dV = dV; g.a.lddv = lddv;
  return prec == ADT_PREC_F32 ? dispatch_attn_gen<PREC_F32>(true, hd, g, (hipStream_t)stream) : dispatch_attn_gen<PREC_BF16>(true, hd, g, (hipStream_t)stream);
}

int adt_embed_sum_fwd(const int32_t* ids, const float* E, const float* P, const float* S0, float scale, int T, int L, int d, float* X,
                      void* stream) {
  if (d % 4) return adt_set_error("embed_sum_fwd: d %% 4");
  EmbedSumArgs a{ids, E, P, S0, scale, T, L, d, X};
  hipLaunchKernelGGL(k_embed_sum_fwd, dim3(grid_for((size_t)T * d / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("embed_sum_fwd");
}

int adt_dropact_fwd(const float* X, int64_t n, float p, const uint32_t* seed, uint32_t site, uint32_t idx_offset, int act, float* Y,
                    void* stream) {
  if (n % 4) return adt_set_error("dropact_fwd: n %% 4");
  DropActArgs a{};
  a.X = X; a.Y = Y; a.n = (size_t)n; a.drop = adt_make_drop(p, seed, site); a.idx_offset = idx_offset; a.act = act;
  hipLaunchKernelGGL(k_dropact<false>, dim3(grid_for((size_t)n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("dropact_fwd");
}

int adt_dropact_bwd(const float* dY, const float* X, int64_t n, float p, const uint32_t* seed, uint32_t site, uint32_t idx_offset, int act,
                    float* dX, int accumulate, void* stream) {
  if (n % 4) return adt_set_error("dropact_bwd: n %% 4");
  DropActArgs a{};
  a.X = X; a.dY = dY; a.dX = dX; a.n = (size_t)n; a.drop = adt_make_drop(p, seed, site); a.idx_offset = idx_offset; a.act = act;
  a.accumulate = accumulate;
  hipLaunchKernelGGL(k_dropact<true>, dim3(grid_for((size_t)n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("dropact_bwd");
}

int adt_gather_rows(const float* F, int ldf, const int32_t* rows, int M, const int32_t* m_dev, int d, float* out, int ldo, void* stream) {
  if (M <= 0) return 0;
  if ((d % 4) || (ldf % 4) || (ldo % 4)) return adt_set_error("gather_rows: d, ld %% 4");
  RowsArgs a{F, ldf, out, ldo, rows, M, d, 0, 0, m_dev};
  hipLaunchKernelGGL(k_rows, dim3(grid_for((size_t)M * d / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("gather_rows");
}

int adt_scatter_rows(const float* G, int ldg, const int32_t* rows, int M, const int32_t* m_dev, int d, float* dF, int lddf, int accumulate,
                     void* stream) {
  if (M <= 0) return 0;
  if ((d % 4) || (ldg % 4) || (lddf % 4)) return adt_set_error("scatter_rows: d, ld %% 4");
  RowsArgs a{G, ldg, dF, lddf, rows, M, d, 1, accumulate, m_dev};
  hipLaunchKernelGGL(k_rows, dim3(grid_for((size_t)M * d / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("scatter_rows");
}

int adt_ce_rows(float* logits, int ld, const int32_t* labels, int M, const int32_t* m_dev, int V, const float* inv_count, float* loss64,
                void* stream) {
  if (M <= 0) return 0;
  CeArgs a{logits, ld, labels, M, V, inv_count, loss64, m_dev};
  const size_t smem = ((size_t)(V + 3) / 4 * 4 + 16) * sizeof(float);
  if (smem <= 150 * 1024 && (ld % 4) == 0 && aligned16(logits)) {      // row resident in LDS: one read + one write per element
    static bool done = false;
    if (!done) {
      if (hipFuncSetAttribute((const void*)k_ce_rows_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
        return adt_set_error("ce_rows: hipFuncSetAttribute");
      done = true;
    }
    hipLaunchKernelGGL(k_ce_rows_lds, dim3(M < 1024 ? M : 1024), dim3(CE_NTH), smem, (hipStream_t)stream, a);
    return check_launch("ce_rows(lds)");
  }
  hipLaunchKernelGGL(k_ce_rows, dim3(M < 4096 ? M : 4096), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("ce_rows");
}

int adt_axpy(float* dst, const float* src, float alpha, int accumulate, int64_t n, const int32_t* mask_ids, int d, void* stream) {
  if (n % 4 || (mask_ids && (d <= 0 || d % 4))) return adt_set_error("axpy: n, d %% 4");
  AxpyArgs a{dst, src, alpha, accumulate, (size_t)n, mask_ids, d};
  hipLaunchKernelGGL(k_axpy, dim3(grid_for((size_t)n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("axpy");
}

int adt_log_softmax_fwd(const float* X, int64_t rows, int H, float* Y, void* stream) {
  if (H < 1 || H > 8) return adt_set_error("log_softmax: H=%d (1..8)", H);
  LsmArgs a{X, Y, nullptr, nullptr, (size_t)rows, H, 0};
  hipLaunchKernelGGL(k_logsoftmax_rows<false>, dim3(grid_for((size_t)rows, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("log_softmax_fwd");
}

int adt_log_softmax_bwd(const float* Y, const float* dY, int64_t rows, int H, float* dX, int accumulate, void* stream) {
  if (H < 1 || H > 8) return adt_set_error("log_softmax: H=%d (1..8)", H);
  LsmArgs a{nullptr, const_cast<float*>(Y), dY, dX, (size_t)rows, H, accumulate};
  hipLaunchKernelGGL(k_logsoftmax_rows<true>, dim3(grid_for((size_t)rows, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("log_softmax_bwd");
}

int adt_grad_sumsq(const float* G, int64_t n, float* out64, void* stream) {
  RangeOptArgs a{};
  a.G = const_cast<float*>(G); a.n = (size_t)n; a.out64 = out64;
  if (adt::zero_f32_async(out64, 64, (hipStream_t)stream)) return adt_set_error("grad_sumsq: zero");
  hipLaunchKernelGGL(k_sumsq64, dim3(grid_for((size_t)n, 256, 512)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("grad_sumsq");
}

int adt_adam_range(float* P, float* G, float* M, float* V, int64_t n, float l2, float clip, float lr, float b1, float b2, float eps, float step,
                   const float* gn2_slots, void* stream) {
  if (n <= 0) return 0;
  RangeOptArgs a{P, G, M, V, (size_t)n, l2, clip, lr, b1, b2, eps, step, gn2_slots, nullptr, 0.f};
  hipLaunchKernelGGL(k_adam_range, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("adam_range");
}

int adt_adamw_range(float* P, float* G, float* M, float* V, int64_t n, float wd, float clip, float lr, float b1, float b2, float eps, float step,
                    const float* gn2_slots, void* stream) {
  if (n <= 0) return 0;
  RangeOptArgs a{P, G, M, V, (size_t)n, 0.f, clip, lr, b1, b2, eps, step, gn2_slots, nullptr, wd};
  hipLaunchKernelGGL(k_adam_range, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("adamw_range");
}

// ---- STOSA-ADT (adt_stosa.cuh) ----------------------------------------------------------------------------------
int adt_wattn_fwd(const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                  const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, int B, int H, int L, int hd, float p,
                  const uint32_t* seed, uint32_t site, uint32_t b_offset, float* Om, int ldom, float* Oc, int ldoc, float* LSE,
                  void* stream) {
  if (hd != 16 && hd != 32 && hd != 64) return adt_set_error("wattn: head_dim=%d unsupported (16/32/64)", hd);
  if (L > 256) return adt_set_error("wattn: L=%d > 256 unsupported", L);
  WAttnArgs a{};
  a.Qm = Qm; a.ldqm = ldqm; a.Qc = Qc; a.ldqc = ldqc; a.Km = Km; a.ldkm = ldkm; a.Kc = Kc; a.ldkc = ldkc; a.Vm = Vm; a.ldvm = ldvm;
  a.Vc = Vc; a.ldvc = ldvc; a.kid = key_ids; a.B = B; a.H = H; a.L = L; a.hd = hd; a.scale = 1.0f / sqrtf((float)hd);
  a.drop = adt_make_drop(p, seed, site); a.bh_offset = b_offset * (uint32_t)H; a.Om = Om; a.ldom = ldom; a.Oc = Oc; a.ldoc = ldoc; a.LSE = LSE;
  const size_t smem = wattn_lds_bytes(L, hd, false);
  if (smem > 160 * 1024) return adt_set_error("wattn_fwd: %zu B of LDS", smem);
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)k_wattn_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return adt_set_error("wattn_fwd: hipFuncSetAttribute");
    done = true;
  }
  hipLaunchKernelGGL(k_wattn_fwd, dim3(B * H), dim3(256), smem, (hipStream_t)stream, a);
  return check_launch("wattn_fwd");
}

int adt_wattn_bwd(const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                  const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, const float* Om, int ldom, const float* Oc,
                  int ldoc, const float* LSE, const float* dOm, int lddom, const float* dOc, int lddoc, int B, int H, int L, int hd, float p,
                  const uint32_t* seed, uint32_t site, uint32_t b_offset, float* dQm, float* dQc, float* dKm, float* dKc, float* dVm,
                  float* dVc, int ldd, void* stream) {
  if (hd != 16 && hd != 32 && hd != 64) return adt_set_error("wattn: head_dim=%d unsupported (16/32/64)", hd);
  if (L > 256) return adt_set_error("wattn: L=%d > 256 unsupported", L);
  WAttnArgs a{};
  a.Qm = Qm; a.ldqm = ldqm; a.Qc = Qc; a.ldqc = ldqc; a.Km = Km; a.ldkm = ldkm; a.Kc = Kc; a.ldkc = ldkc; a.Vm = Vm; a.ldvm = ldvm;
  a.Vc = Vc; a.ldvc = ldvc; a.kid = key_ids; a.B = B; a.H = H; a.L = L; a.hd = hd; a.scale = 1.0f / sqrtf((float)hd);
  a.drop = adt_make_drop(p, seed, site); a.bh_offset = b_offset * (uint32_t)H;
  a.Om = const_cast<float*>(Om); a.ldom = ldom; a.Oc = const_cast<float*>(Oc); a.ldoc = ldoc; a.LSE = const_cast<float*>(LSE);
  a.dOm = dOm; a.lddom = lddom; a.dOc = dOc; a.lddoc = lddoc;
  a.dQm = dQm; a.dQc = dQc; a.dKm = dKm; a.dKc = dKc; a.dVm = dVm; a.dVc = dVc; a.ldd = ldd;
  const size_t smem = wattn_lds_bytes(L, hd, true);
  if (smem > 160 * 1024) return adt_set_error("wattn_bwd: %zu B of LDS", smem);
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)k_wattn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return adt_set_error("wattn_bwd: hipFuncSetAttribute");
    done = true;
  }
  hipLaunchKernelGGL(k_wattn_bwd, dim3(B * H), dim3(256), smem, (hipStream_t)stream, a);
  return check_launch("wattn_bwd");
}

// The same attention on the matrix cores; 1 = shape not covered (see wattn_mfma_ok above)
int adt_wattn_mfma_fwd(int prec, const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                       const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, int B, int H, int L, int hd, float p,
                       const uint32_t* seed, uint32_t site, uint32_t b_offset, float* Om, int ldom, float* Oc, int ldoc, float* LSE,
                       void* stream) {
  const int lds[8] = {ldqm, ldqc, ldkm, ldkc, ldvm, ldvc, ldom, ldoc};
  if (!wattn_mfma_ok(L, hd, lds, 8)) return 1;
  WAttnArgs a{};
  a.Qm = Qm; a.ldqm = ldqm; a.Qc = Qc; a.ldqc = ldqc; a.Km = Km; a.ldkm = ldkm; a.Kc = Kc; a.ldkc = ldkc; a.Vm = Vm; a.ldvm = ldvm;
  a.Vc = Vc; a.ldvc = ldvc; a.kid = key_ids; a.B = B; a.H = H; a.L = L; a.hd = hd; a.scale = 1.0f / sqrtf((float)hd);
  a.drop = adt_make_drop(p, seed, site); a.bh_offset = b_offset * (uint32_t)H; a.Om = Om; a.ldom = ldom; a.Oc = Oc; a.ldoc = ldoc; a.LSE = LSE;
  return wattn_mfma_dispatch(prec, false, a, (hipStream_t)stream);
}

int adt_wattn_mfma_bwd(int prec, const float* Qm, int ldqm, const float* Qc, int ldqc, const float* Km, int ldkm, const float* Kc, int ldkc,
                       const float* Vm, int ldvm, const float* Vc, int ldvc, const int32_t* key_ids, const float* Om, int ldom, const float* Oc,
                       int ldoc, const float* LSE, const float* dOm, int lddom, const float* dOc, int lddoc, int B, int H, int L, int hd, float p,
                       const uint32_t* seed, uint32_t site, uint32_t b_offset, float* dQm, float* dQc, float* dKm, float* dKc, float* dVm,
                       float* dVc, int ldd, void* stream) {
  const int lds[11] = {ldqm, ldqc, ldkm, ldkc, ldvm, ldvc, lddom, lddoc, ldd, ldom, ldoc};
  if (!wattn_mfma_ok(L, hd, lds, 11)) return 1;
  WAttnArgs a{};
  a.Qm = Qm; a.ldqm = ldqm; a.Qc = Qc; a.ldqc = ldqc; a.Km = Km; a.ldkm = ldkm; a.Kc = Kc; a.ldkc = ldkc; a.Vm = Vm; a.ldvm = ldvm;
  a.Vc = Vc; a.ldvc = ldvc; a.kid = key_ids; a.B = B; a.H = H; a.L = L; a.hd = hd; a.scale = 1.0f / sqrtf((float)hd);
  a.drop = adt_make_drop(p, seed, site); a.bh_offset = b_offset * (uint32_t)H; a.LSE = const_cast<float*>(LSE);
  a.Om = const_cast<float*>(Om); a.ldom = ldom; a.Oc = const_cast<float*>(Oc); a.ldoc = ldoc;
  a.dOm = dOm; a.lddom = lddom; a.dOc = dOc; a.lddoc = lddoc;
  a.dQm = dQm; a.dQc = dQc; a.dKm = dKm; a.dKc = dKc; a.dVm = dVm; a.dVc = dVc; a.ldd = ldd;
  return wattn_mfma_dispatch(prec, true, a, (hipStream_t)stream);
}

int adt_wdist_bpr(const float* Sm, const float* Sc, int lds, const float* Em, const float* Ec, const int32_t* pos, const int32_t* neg, int T,
                  int d, float pvn_weight, const float* inv_count, float* dSm, float* dSc, int ldds, float* dEm, float* dEc, float* loss3,
                  void* stream) {
  if (d % 64) return adt_set_error("wdist_bpr: d=%d must be a multiple of 64", d);
  WBprArgs a{Sm, Sc, lds, Em, Ec, pos, neg, T, d, pvn_weight, inv_count, dSm, dSc, ldds, dEm, dEc, loss3};
  hipLaunchKernelGGL(k_wdist_bpr, dim3(grid_for(T, 4, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("wdist_bpr");
}

int adt_wdist_full(const float* Sm, const float* Sc, int lds, const float* Em, const float* Ec, int B, int V, int d, float* dist, int ldo,
                   void* stream) {
  if (d % 4) return adt_set_error("wdist_full: d %% 4");
  WFullArgs a{Sm, Sc, lds, Em, Ec, B, V, d, dist, ldo};
  hipLaunchKernelGGL(k_wdist_full, dim3(grid_for((size_t)B * V, 16, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("wdist_full");
}

int adt_topk_masked(float* dist, int ld, int B, int N, const int32_t* indptr, const int32_t* indices, int k, int32_t* out_idx,
                    float* out_val, void* stream) {
  if (B <= 0) return 0;
  if (k <= 0 || k > N) return adt_set_error("topk_masked: need 0 < k <= N");
  if (ld < N) return adt_set_error("topk_masked: ld < N");
  if ((indptr == nullptr) != (indices == nullptr)) return adt_set_error("topk_masked: indptr and indices go together");
  TopkArgs a{dist, ld, B, N, indptr, indices, k, out_idx, out_val};
  hipLaunchKernelGGL(k_topk_masked, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("topk_masked");
}

int adt_dense_gradsrc(const float* dY, int lddy, int T, int N, const int32_t* mask_ids, float p, const uint32_t* seed, uint32_t site,
                      uint32_t row_offset, int act, const float* U, int ldu, float* G, int ldg, const int32_t* t_dev, void* stream) {
  if (T <= 0 || N <= 0) return 0;
  if ((N % 4) || (lddy % 4) || (ldg % 4) || !aligned16(dY) || !aligned16(G)) return adt_set_error("dense_gradsrc: N, ld %% 4 and 16-byte alignment required");
  if (act != ACT_NONE && (!U || (ldu % 4) || !aligned16(U))) return adt_set_error("dense_gradsrc: activation needs the saved pre-activation");
  GradSrcOutArgs a{};
  a.G.dY = dY; a.G.lddy = lddy; a.G.T = T; a.G.N = N; a.G.U = U; a.G.ldu = ldu; a.G.act = act;
  a.G.drop = adt_make_drop(p, seed, site); a.G.row_offset = row_offset; a.G.ids = mask_ids; a.G.idx_ld = N; a.G.idx_off = 0;
  a.out = G; a.ldo = ldg; a.t_dev = t_dev;
  hipLaunchKernelGGL(k_gradsrc, dim3(grid_for((size_t)T * (N / 4), 1024, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("dense_gradsrc");
}

}  // extern "C"
