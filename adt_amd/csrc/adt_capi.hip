// C ABI (include/adt_hip.h): launch wrappers for the per-stage kernels.  Host code only enqueues work on
// the caller's stream: no allocation, no synchronisation, graph-capturable.
#include "adt_host.h"
#include <stdlib.h>

#include "adt_attn.cuh"
#include "adt_attn_bf16.cuh"
#include "adt_bwdchain.cuh"
#include "adt_fwdchain.cuh"
#include "adt_misc.cuh"
#include "adt_itemgrad.cuh"
#include "adt_rowops.cuh"

using namespace adt;

static thread_local char g_err[512] = "";

int adt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return adt_set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

DropCfg adt_make_drop(float p, const uint32_t* seed, uint32_t site) {
  DropCfg d;
  d.seed = seed;
  d.site = site;
  const int t8 = p > 0.f ? (int)((double)p * 256.0 + 0.5) : 0;      // oracle/rng.py: threshold(p)
  if (t8 > 0 && seed != nullptr) {
    d.thr = (uint32_t)(t8 > 255 ? 255 : t8);
    d.scale = (float)(1.0 / (1.0 - (double)d.thr / 256.0));
  } else {
    d.thr = 0;
    d.scale = 1.0f;
  }
  return d;
}

static int grid_for(size_t work_items, int per_block, int cap) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return (int)g;
}

template <int PREC>
static void launch_linear_bwd(const LinBwdArgs& a, int nch, int grid, hipStream_t s) {
  if (nch == 1) hipLaunchKernelGGL((k_linear_bwd<PREC, 64, 1>), dim3(grid), dim3(256), 0, s, a);
  else if (nch == 2) hipLaunchKernelGGL((k_linear_bwd<PREC, 64, 2>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_linear_bwd<PREC, 64, 3>), dim3(grid), dim3(256), 0, s, a);
}

// ---- attention dispatch --------------------------------------------------------------------------
static int attn_waves(bool bwd) {
  // waves per workgroup: tunable for experiments (ADT_ATTN_FWD_NW / ADT_ATTN_BWD_NW = 4 or 8)
  static int nw[2] = {0, 0};
  if (!nw[0]) {
    const char* f = getenv("ADT_ATTN_FWD_NW");
    const char* b = getenv("ADT_ATTN_BWD_NW");
    nw[0] = (f && atoi(f) == 4) ? 4 : 8;
    nw[1] = (b && atoi(b) == 4) ? 4 : 8;
  }
  return nw[bwd ? 1 : 0];
}

// bf16-image kernels (adt_attn_bf16.cuh) for the bf16-operand precision
template <int HD, int MAXKT, int NW>
static int launch_attn_bf16(bool bwd, const AttnArgs& a, hipStream_t s) {
  const size_t smem = bwd ? AttnBf16Lds<HD, MAXKT>::bwd_bytes : AttnBf16Lds<HD, MAXKT>::fwd_bytes;
  if (smem > 160 * 1024) return adt_set_error("attention(bf16): L/hd too large for LDS-resident form (%zu B)", smem);
  static bool attr_done[2] = {false, false};
  if (!attr_done[bwd ? 1 : 0]) {
    hipError_t e = bwd ? hipFuncSetAttribute((const void*)k_attn_bwd_bf16<HD, MAXKT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                       : hipFuncSetAttribute((const void*)k_attn_fwd_bf16<HD, MAXKT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return adt_set_error("attention(bf16): hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[bwd ? 1 : 0] = true;
  }
  const int grid = a.B * a.H;
  if (bwd) hipLaunchKernelGGL((k_attn_bwd_bf16<HD, MAXKT, NW>), dim3(grid), dim3(NW * 64), smem, s, a);
  else hipLaunchKernelGGL((k_attn_fwd_bf16<HD, MAXKT, NW>), dim3(grid), dim3(NW * 64), smem, s, a);
  return check_launch(bwd ? "attn_bwd_bf16" : "attn_fwd_bf16");
}

template <int PREC, int HD, int MAXKT, int NW>
static int launch_attn_nw(bool bwd, const AttnArgs& a, hipStream_t s) {
  if constexpr (PREC == PREC_BF16) {
    static int use_img = -1;
    if (use_img < 0) { const char* e = getenv("ADT_ATTN_BF16_IMG"); use_img = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_img) return launch_attn_bf16<HD, MAXKT, NW>(bwd, a, s);
  }
  constexpr int RS = HD + 4, LP = MAXKT * 16;
  const size_t smem = bwd ? (size_t)(4 * LP * RS + 2 * LP) * sizeof(float) : (size_t)(2 * LP * RS) * sizeof(float);
  if (smem > 160 * 1024) return adt_set_error("attention: L/hd too large for LDS-resident form (%zu B)", smem);
  static bool attr_done[2] = {false, false};
  if (!attr_done[bwd ? 1 : 0]) {
    hipError_t e = bwd ? hipFuncSetAttribute((const void*)k_attn_bwd<PREC, HD, MAXKT, NW>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                       : hipFuncSetAttribute((const void*)k_attn_fwd<PREC, HD, MAXKT, NW>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return adt_set_error("attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[bwd ? 1 : 0] = true;
  }
  const int grid = a.B * a.H;
  if (bwd) hipLaunchKernelGGL((k_attn_bwd<PREC, HD, MAXKT, NW>), dim3(grid), dim3(NW * 64), smem, s, a);
  else hipLaunchKernelGGL((k_attn_fwd<PREC, HD, MAXKT, NW>), dim3(grid), dim3(NW * 64), smem, s, a);
  return check_launch(bwd ? "attn_bwd" : "attn_fwd");
}

template <int PREC, int HD, int MAXKT>
static int launch_attn(bool bwd, const AttnArgs& a, hipStream_t s) {
  if (attn_waves(bwd) == 4) return launch_attn_nw<PREC, HD, MAXKT, 4>(bwd, a, s);
  return launch_attn_nw<PREC, HD, MAXKT, 8>(bwd, a, s);
}

template <int PREC, int HD>
static int dispatch_attn_l(bool bwd, const AttnArgs& a, hipStream_t s) {
  if (a.L <= 64) return launch_attn<PREC, HD, 4>(bwd, a, s);
  if (a.L <= 128) return launch_attn<PREC, HD, 8>(bwd, a, s);
  if (a.L <= 224) return launch_attn<PREC, HD, 14>(bwd, a, s);
  return adt_set_error("attention: L=%d > 224 unsupported", a.L);
}

template <int PREC>
static int dispatch_attn_hd(bool bwd, int hd, const AttnArgs& a, hipStream_t s) {
  if (hd == 16) return dispatch_attn_l<PREC, 16>(bwd, a, s);
  if (hd == 32) return dispatch_attn_l<PREC, 32>(bwd, a, s);
  if (hd == 64 && (!bwd || a.L <= 128)) return dispatch_attn_l<PREC, 64>(bwd, a, s);
  return adt_set_error("attention: head_dim=%d (L=%d) unsupported", hd, a.L);
}

static int dispatch_attn(int prec, bool bwd, int hd, const AttnArgs& a, hipStream_t s) {
  if ((a.ldq % 4) || (a.ldk % 4) || (a.ldv % 4) || (a.ldo % 4)) return adt_set_error("attention: ld %% 4");
  if (prec == ADT_PREC_F32) return dispatch_attn_hd<PREC_F32>(bwd, hd, a, s);
  return dispatch_attn_hd<PREC_BF16>(bwd, hd, a, s);
}

template <int PREC, int NW>
static int launch_bwdchain_t(int which, const BwdChainArgs& a, hipStream_t s) {
  const int ntiles = (a.T + 15) / 16;
  int grid = (ntiles + NW - 1) / NW;
  if (grid > 256) grid = 256;
  // whole rounds: every workgroup runs the same number of cooperative rounds; fewer, fuller rounds waste less
  {
    const int per = grid * NW;
    const int rounds = (ntiles + per - 1) / per;
    grid = (ntiles + rounds * NW - 1) / (rounds * NW);
  }
  const void* fn = nullptr;
  size_t smem = 0;
  switch (which) {
    case 0: fn = (const void*)k_enc_post_bwd<PREC, NW, 0>; smem = BwdLds<PREC, NW, 3>::bytes; break;
    case 6: fn = (const void*)k_enc_post_bwd<PREC, NW, 2>; smem = BwdLds<PREC, NW, 3>::bytes; break;
    case 7: fn = (const void*)k_enc_post_bwd<PREC, NW, 4>; smem = BwdLds<PREC, NW, 3>::bytes; break;
    case 1: fn = (const void*)k_dec_post_bwd<PREC, NW>; smem = BwdLds<PREC, NW, 3>::bytes; break;
    case 2: fn = (const void*)k_pre_bwd<PREC, NW, true>; smem = BwdLds<PREC, NW, 3>::bytes; break;
    case 3: fn = (const void*)k_pre_bwd<PREC, NW, false>; smem = BwdLds<PREC, NW, 3>::bytes; break;
    case 4: fn = (const void*)k_dec_mid_bwd<PREC, NW>; smem = BwdLds<PREC, NW, 2>::bytes; break;
    case 5: fn = (const void*)k_kv_bwd<PREC, NW>; smem = BwdLds<PREC, NW, 2>::bytes; break;
    default: return adt_set_error("bwdchain: bad kernel id %d", which);
  }
  if (smem > 160 * 1024) return adt_set_error("bwdchain %d: %zu B of LDS", which, smem);
  static bool done[8] = {false, false, false, false, false, false, false, false};
  if (!done[which]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return adt_set_error("bwdchain: hipFuncSetAttribute");
    done[which] = true;
  }
  BwdChainArgs args = a;
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("ADT_BWD_ABLATE"); ablate = e ? atoi(e) : 0; }
  args.ablate = ablate;
  void* kargs[] = {&args};
  if (hipLaunchKernel(fn, dim3(grid), dim3(NW * 64), kargs, smem, s) != hipSuccess) return adt_set_error("bwdchain %d: launch failed", which);
  return check_launch("bwdchain");
}

template <int PREC, int NW>
static int launch_fwdchain_t(int which, const FwdChainArgs& a, hipStream_t s) {
  const int ntiles = (a.T + 15) / 16;
  int grid = (ntiles + NW - 1) / NW;
  static int cap = 0;
  if (!cap) { const char* e = getenv("ADT_FWD_GRID"); cap = e ? atoi(e) : 512; if (cap < 1) cap = 512; }
  if (grid > cap) grid = cap;
  const void* fn = nullptr;
  size_t smem = 0;
  switch (which) {
    case 0: fn = (const void*)k_pre_fwd<PREC, NW, true>; smem = FwdLds<PREC, NW, 3>::bytes; break;
    case 1: fn = (const void*)k_pre_fwd<PREC, NW, false>; smem = FwdLds<PREC, NW, 3>::bytes; break;
    case 2: fn = (const void*)k_enc_post_fwd<PREC, NW, 0>; smem = FwdLds<PREC, NW, 3>::bytes; break;
    case 6: fn = (const void*)k_enc_post_fwd<PREC, NW, 2>; smem = FwdLds<PREC, NW, 3>::bytes; break;
    case 7: fn = (const void*)k_enc_post_fwd<PREC, NW, 4>; smem = FwdLds<PREC, NW, 3>::bytes; break;
    case 8: fn = (const void*)k_enc_post_fwd<PREC, NW, 8>; smem = FwdLds<PREC, NW, 3>::bytes; break;
    case 3: fn = (const void*)k_dec_mid_fwd<PREC, NW>; smem = FwdLds<PREC, NW, 2>::bytes; break;
    case 4: fn = (const void*)k_dec_post_fwd<PREC, NW>; smem = FwdLds<PREC, NW, 3>::bytes; break;
    case 5: fn = (const void*)k_final_fwd<PREC, NW>; smem = FwdLds<PREC, NW, 4>::bytes; break;
    default: return adt_set_error("fwdchain: bad kernel id %d", which);
  }
  static bool done[9] = {false, false, false, false, false, false, false, false, false};
  if (!done[which]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return adt_set_error("fwdchain: hipFuncSetAttribute");
    done[which] = true;
  }
  FwdChainArgs args = a;
  void* kargs[] = {&args};
  if (hipLaunchKernel(fn, dim3(grid), dim3(NW * 64), kargs, smem, s) != hipSuccess) return adt_set_error("fwdchain %d: launch failed", which);
  return check_launch("fwdchain");
}

int adt_launch_fwdchain(int prec, int which, const FwdChainArgs& a, void* stream) {
  if (prec == ADT_PREC_F32) return launch_fwdchain_t<PREC_F32, 8>(which, a, (hipStream_t)stream);
  return launch_fwdchain_t<PREC_BF16, 8>(which, a, (hipStream_t)stream);
}

int adt_launch_bwdchain(int prec, int which, const BwdChainArgs& a, void* stream) {
  // cooperative weight gradients (adt_wave.cuh: Coop): 8 waves per workgroup in bf16 mode; the fp32-exact mode
  // publishes fp32 register images (twice the LDS) and runs 4 waves per workgroup
  if (prec == ADT_PREC_F32) return launch_bwdchain_t<PREC_F32, 4>(which, a, (hipStream_t)stream);
  return launch_bwdchain_t<PREC_BF16, 8>(which, a, (hipStream_t)stream);
}

extern "C" {

int adt_version(void) { return 1; }
const char* adt_last_error(void) { return g_err; }

int adt_rng_keep(uint32_t seed, uint32_t site, uint32_t idx, float p) {
  DropCfg d = adt_make_drop(p, &seed, site);
  if (!d.thr) return 1;
  return adt_keep(adt_site_key(seed, site), idx, d.thr) ? 1 : 0;
}

int adt_embed_fwd(const int32_t* ids, const float* E, const float* P, int T, int L, int d, float p,
                  const uint32_t* seed, uint32_t site, uint32_t row_offset, float* X, void* stream) {
  if (d % 4) return adt_set_error("embed_fwd: d %% 4 != 0");
  EmbedArgs a{};
  a.ids = ids; a.E = E; a.P = P; a.T = T; a.L = L; a.d = d; a.scale = sqrtf((float)d);
  a.drop = adt_make_drop(p, seed, site); a.row_offset = row_offset; a.X = X;
  hipLaunchKernelGGL(k_embed_fwd, dim3(grid_for((size_t)T * d / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("embed_fwd");
}

int adt_embed_bwd(const int32_t* ids, const float* dX, int T, int L, int d, float p, const uint32_t* seed,
                  uint32_t site, uint32_t row_offset, float* dE, float* dP, void* stream) {
  if (d % 4 || T % L) return adt_set_error("embed_bwd: bad shape");
  EmbedArgs a{};
  a.ids = ids; a.T = T; a.L = L; a.d = d; a.scale = sqrtf((float)d);
  a.drop = adt_make_drop(p, seed, site); a.row_offset = row_offset; a.dX = dX; a.dE = dE; a.dP = dP;
  const int B = T / L;
  const int gx = (L * d / 4 + 255) / 256;
  int gy = B < 32 ? B : 32;
  hipLaunchKernelGGL(k_posemb_bwd, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, a);
  ScatterArgs sc{};
  sc.ids = ids; sc.G = dX; sc.ldg = d; sc.rowscale = nullptr; sc.T = T; sc.d = d; sc.scale = a.scale; sc.drop = a.drop;
  sc.row_offset = row_offset; sc.dE = dE;
  hipLaunchKernelGGL(k_item_scatter, dim3(grid_for(T, 4, 2048)), dim3(256), 0, (hipStream_t)stream, sc);
  return check_launch("embed_bwd");
}

int adt_item_scatter(const int32_t* ids, const float* G, int ldg, const float* rowscale, int T, int d, float scale, float p,
                     const uint32_t* seed, uint32_t site, uint32_t row_offset, float* rep, int nrep, int64_t rep_stride,
                     void* stream) {
  ScatterArgs sc{};
  sc.ids = ids; sc.G = G; sc.ldg = ldg; sc.rowscale = rowscale; sc.T = T; sc.d = d; sc.scale = scale;
  sc.drop = adt_make_drop(p, seed, site); sc.row_offset = row_offset; sc.dE = rep; sc.nrep = nrep; sc.rep_stride = (size_t)rep_stride;
  hipLaunchKernelGGL(k_item_scatter, dim3(grid_for(T, 4, 2048)), dim3(256), 0, (hipStream_t)stream, sc);
  return check_launch("item_scatter");
}

// fused forms used by the executor (adt_host.h)
int adt_logits_bwd_scatter(const float* F, int ldf, const float* E, const int32_t* pos, const int32_t* neg, const float* dpos, const float* dneg,
                           int T, int d, float* dF, int lddf, float* rep, int nrep, int64_t rep_stride, void* stream) {
  LogitsScatterArgs a{F, ldf, E, pos, neg, dpos, dneg, T, d, dF, lddf, rep, nrep, (size_t)rep_stride};
  hipLaunchKernelGGL(k_logits_bwd_scatter, dim3(grid_for(T, 4, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("logits_bwd_scatter");
}

int adt_logits_bce_scatter(const float* F, const float* E, const int32_t* pos, const int32_t* neg, const float* norms, int T, float* pos_logits,
                           float* neg_logits, float* dpos, float* dneg, float* loss_bce, float* dF, float* rep, int nrep, int64_t rep_stride, void* stream) {
  return adt_logits_bce_scatter_ex(F, E, pos, neg, norms, T, pos_logits, neg_logits, dpos, dneg, loss_bce, dF, rep, nrep, rep_stride, 0, stream);
}
// neg_only: the item rows of the positive ids are left to adt_embed_bwd3 of the same step
int adt_logits_bce_scatter_ex(const float* F, const float* E, const int32_t* pos, const int32_t* neg, const float* norms, int T, float* pos_logits,
                              float* neg_logits, float* dpos, float* dneg, float* loss_bce, float* dF, float* rep, int nrep, int64_t rep_stride,
                              int neg_only, void* stream) {
  LogitsBceArgs a{F, E, pos, neg, norms, T, pos_logits, neg_logits, dpos, dneg, loss_bce, dF, rep, nrep, (size_t)rep_stride, neg_only};
  hipLaunchKernelGGL(k_logits_bce_scatter, dim3(grid_for(T, 4 * 16, 1024)), dim3(256), 0, (hipStream_t)stream, a);      // ~16 rows per wave
  return check_launch("logits_bce_scatter");
}

// adt_loss_seeds_split_prefetch(): the NEXT adt_loss_seeds_prefetch of this host thread copies only the first half of the ring slot and leaves the
// second half (+ the staged mark) to the next adt_embed_bwd3 launch of this thread (same step, later in the stream).
static thread_local RingPrefetchArgs g_pf_job;
static thread_local bool g_pf_set = false, g_pf_split = false;
void adt_loss_seeds_split_prefetch() { g_pf_split = true; g_pf_set = false; }

/* The encoder embedding gradient, the decoder embedding gradient and the positive-logit rows into the item-table replicas in one pass (d = 64,
 * T a multiple of L): one atomic row-add per token where the three ids are the shifts of one item list that the reference's sampler produces
 * (seq[b, l] == dec[b, l + 1] == pos[b, l - 1]); any other ids are added on their own.  dP += the positional sums of both embeddings. */
int adt_embed_bwd3(const int32_t* seq, const int32_t* dec, const int32_t* pos, const float* dXs, const float* dXd, const float* F, const float* dpos,
                   int T, int L, float p, const uint32_t* seed, uint32_t site_seq, uint32_t site_dec, uint32_t row_offset, float* dP, float* rep,
                   int nrep, int64_t rep_stride, void* stream) {
  if (L < 1 || T % L) return adt_set_error("embed_bwd3: T %d is not a multiple of L %d", T, L);
  const int B = T / L, ns = B < 32 ? B : 32;
  EmbedBwd3Args a{seq, dec, pos, dXs, dXd, F, dpos, T, L, 8.0f, adt_make_drop(p, seed, site_seq), adt_make_drop(p, seed, site_dec), row_offset, dP, rep, nrep,
                  (size_t)rep_stride, ns};
  if (g_pf_set) {      // the second half of this step's ring prefetch (adt_loss_seeds_split_prefetch)
    g_pf_set = false;
    hipLaunchKernelGGL(k_embed_bwd3_prefetch, dim3((L * ns + 3) / 4 + 16), dim3(256), 0, (hipStream_t)stream, a, g_pf_job, 16);
  } else {
    hipLaunchKernelGGL(k_embed_bwd3, dim3((L * ns + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("embed_bwd3");
}

int adt_embed_bwd_rep(const int32_t* ids, const float* dX, int T, int L, int d, float p, const uint32_t* seed, uint32_t site, uint32_t row_offset,
                      float* dP, float* rep, int nrep, int64_t rep_stride, void* stream) {
  if (d != 64 || T % L) {   // general widths: the two separate passes
    if (int rc = adt_posemb_bwd(ids, dX, T, L, d, p, seed, site, row_offset, dP, stream)) return rc;
    return adt_item_scatter(ids, dX, d, nullptr, T, d, sqrtf((float)d), p, seed, site, row_offset, rep, nrep, rep_stride, stream);
  }
  const int B = T / L, ns = B < 32 ? B : 32;
  EmbedBwdArgs a{ids, dX, T, L, sqrtf((float)d), adt_make_drop(p, seed, site), row_offset, dP, rep, nrep, (size_t)rep_stride, ns};
  hipLaunchKernelGGL(k_embed_bwd64, dim3((L * ns + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("embed_bwd_rep");
}

int adt_replica_reduce2(float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1, const float* r1, int64_t n1, int nrep1, int64_t s1,
                        void* stream) {
  if (n0 % 4 || s0 % 4 || n1 % 4 || s1 % 4) return adt_set_error("replica_reduce2: n, stride %% 4");
  if (n0 <= 0) return adt_replica_reduce(d1, r1, n1, nrep1, s1, stream);
  if (n1 <= 0) return adt_replica_reduce(d0, r0, n0, nrep0, s0, stream);
  RepReduce2Args a{{d0, d1}, {r0, r1}, {(size_t)n0, (size_t)n1}, {nrep0, nrep1}, {(size_t)s0, (size_t)s1}, 0};
  const int g0 = grid_for((size_t)n0 / 4, 256, 1024), g1 = grid_for((size_t)n1 / 4, 256, 1024);
  a.g0 = g0;
  hipLaunchKernelGGL(k_replica_reduce2, dim3(g0 + g1), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("replica_reduce2");
}

int adt_replica_reduce(float* dE, const float* rep, int64_t n, int nrep, int64_t rep_stride, void* stream) {
  if (n % 4 || rep_stride % 4) return adt_set_error("replica_reduce: n, stride %% 4");
  hipLaunchKernelGGL(k_replica_reduce, dim3(grid_for((size_t)n / 4, 256, 1024)), dim3(256), 0, (hipStream_t)stream, dE, rep, (size_t)n, nrep,
                     (size_t)rep_stride);
  return check_launch("replica_reduce");
}

int adt_posemb_bwd(const int32_t* ids, const float* dX, int T, int L, int d, float p, const uint32_t* seed,
                   uint32_t site, uint32_t row_offset, float* dP, void* stream) {
  if (d % 4 || T % L) return adt_set_error("posemb_bwd: bad shape");
  EmbedArgs a{};
  a.ids = ids; a.T = T; a.L = L; a.d = d; a.scale = sqrtf((float)d);
  a.drop = adt_make_drop(p, seed, site); a.row_offset = row_offset; a.dX = dX; a.dP = dP;
  const int B = T / L;
  hipLaunchKernelGGL(k_posemb_bwd, dim3((L * d / 4 + 255) / 256, B < 32 ? B : 32), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("posemb_bwd");
}

int adt_logits_bwd_df(const float* E, const int32_t* pos, const int32_t* neg, const float* dpos, const float* dneg, int T,
                      int d, float* dF, int lddf, void* stream) {
  LogitsArgs a{};
  a.E = E; a.pos = pos; a.neg = neg; a.T = T; a.d = d; a.dpos = dpos; a.dneg = dneg; a.dF = dF; a.lddf = lddf;
  hipLaunchKernelGGL(k_logits_bwd, dim3(grid_for(T, 16, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("logits_bwd_df");
}

int adt_layernorm_fwd(const float* X, int ldx, const float* gamma, const float* beta, float eps, int T, int d,
                      float* Y, int ldy, void* stream) {
  LnArgs a{};
  a.X = X; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.eps = eps; a.Y = Y; a.ldy = ldy; a.T = T;
  const int grid = grid_for(T, 16, 2048);
  if (d == 64) hipLaunchKernelGGL(k_ln_fwd<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (d == 128) hipLaunchKernelGGL(k_ln_fwd<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (d == 256) hipLaunchKernelGGL(k_ln_fwd<256>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else return adt_set_error("layernorm_fwd: d=%d unsupported (64/128/256)", d);
  return check_launch("layernorm_fwd");
}

int adt_layernorm_bwd(const float* dY, int lddy, const float* X, int ldx, const float* gamma, float eps, int T,
                      int d, float* dX, int lddx, int accumulate, float* dgamma, float* dbeta, void* stream) {
  return adt_layernorm_bwd_rep(dY, lddy, X, ldx, gamma, eps, T, d, dX, lddx, accumulate, dgamma, dbeta, 1, 0, stream);
}

// dgamma / dbeta into nrep replicas (block b -> replica b % nrep): with one copy the grid is capped at 256 blocks (every block ends with one
// atomic per column on the same 2 d addresses: a 256-deep chain, ~6 us) and one wave per SIMD cannot hide the row latency; with replicas
// the grid is 1,024 blocks.
int adt_layernorm_bwd_rep(const float* dY, int lddy, const float* X, int ldx, const float* gamma, float eps, int T, int d, float* dX, int lddx,
                          int accumulate, float* dgamma, float* dbeta, int nrep, int64_t rep_stride, void* stream) {
  LnArgs a{};
  a.X = X; a.ldx = ldx; a.gamma = gamma; a.eps = eps; a.T = T; a.dY = dY; a.lddy = lddy; a.dX = dX; a.lddx = lddx;
  a.acc = accumulate; a.dgamma = dgamma; a.dbeta = dbeta; a.nrep = nrep; a.rep_stride = (size_t)rep_stride;
  const int grid = grid_for(T, 16, nrep > 1 ? 1024 : 256);
  if (d == 64) hipLaunchKernelGGL(k_ln_bwd<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (d == 128) hipLaunchKernelGGL(k_ln_bwd<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (d == 256) hipLaunchKernelGGL(k_ln_bwd<256>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else return adt_set_error("layernorm_bwd: d=%d unsupported (64/128/256)", d);
  return check_launch("layernorm_bwd");
}

/* adt_layernorm_bwd_rep with PRIVATE per-block sums: block b stores dgamma / dbeta sums at part + b * 2 * d (dgamma | dbeta); returns the number of
 * blocks (<= max_blocks) or < 0.  The caller adds them up in block order (k_fold_parts_gradnorm, job 3). */
int adt_layernorm_bwd_parts(const float* dY, int lddy, const float* X, int ldx, const float* gamma, float eps, int T, int d, float* dX, int lddx,
                            int accumulate, float* part, int max_blocks, void* stream) {
  LnArgs a{};
  a.X = X; a.ldx = ldx; a.gamma = gamma; a.eps = eps; a.T = T; a.dY = dY; a.lddy = lddy; a.dX = dX; a.lddx = lddx;
  a.acc = accumulate; a.dgamma = part; a.dbeta = part + d; a.nrep = 1; a.rep_stride = (size_t)(2 * d); a.plain = 1;
  const int grid = grid_for(T, 16, max_blocks);
  if (d == 64) hipLaunchKernelGGL(k_ln_bwd<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (d == 128) hipLaunchKernelGGL(k_ln_bwd<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (d == 256) hipLaunchKernelGGL(k_ln_bwd<256>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else return adt_set_error("layernorm_bwd_parts: d=%d unsupported (64/128/256)", d);
  const int rc = check_launch("layernorm_bwd_parts");
  return rc < 0 ? rc : grid;
}

int adt_linear_fwd(int prec, const float* X, int ldx, const float* W, const float* b, int T, int K, int N,
                   float* Y, int ldy, float p, const uint32_t* seed, uint32_t site, uint32_t row_offset,
                   int relu, const float* R1, int ldr1, const float* R2, int ldr2, const int32_t* mask_ids,
                   void* stream) {
  if (K != 64) return adt_set_error("linear_fwd: K=%d unsupported (64)", K);
  if (N % 16 || N <= 0) return adt_set_error("linear_fwd: N=%d must be a positive multiple of 16", N);
  if ((ldx % 4) || (ldy % 4) || (R1 && ldr1 % 4) || (R2 && ldr2 % 4)) return adt_set_error("linear_fwd: ld %% 4");
  LinFwdArgs a{};
  a.X = X; a.ldx = ldx; a.W = W; a.b = b; a.N = N; a.Y = Y; a.ldy = ldy; a.T = T;
  a.drop = adt_make_drop(p, seed, site); a.row_offset = row_offset; a.relu = relu;
  a.R1 = R1; a.ldr1 = ldr1; a.R2 = R2; a.ldr2 = ldr2; a.ids = mask_ids;
  const int grid = grid_for(T, BM, 768);
  if (prec == ADT_PREC_F32) hipLaunchKernelGGL((k_linear_fwd<PREC_F32, 64>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((k_linear_fwd<PREC_BF16, 64>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("linear_fwd");
}

int adt_linear_bwd(int prec, const float* dY, int lddy, const float* X, int ldx, const float* W, int T, int K,
                   int N, const int32_t* mask_ids, float p, const uint32_t* seed, uint32_t site,
                   uint32_t row_offset, const float* U, int ldu, float* dX, int lddx, int beta,
                   const float* Radd, int ldradd, const int32_t* radd_ids, float* dW, float* db, void* stream) {
  if (K != 64) return adt_set_error("linear_bwd: K=%d unsupported (64)", K);
  if (N % 16 || N <= 0 || N > 192) return adt_set_error("linear_bwd: N=%d must be a multiple of 16 in (0,192]", N);
  if ((lddy % 4) || (ldx % 4) || (dX && lddx % 4) || (U && ldu % 4) || (Radd && ldradd % 4))
    return adt_set_error("linear_bwd: ld %% 4");
  LinBwdArgs a{};
  a.dY = dY; a.lddy = lddy; a.X = X; a.ldx = ldx; a.W = W; a.N = N; a.T = T; a.ids = mask_ids;
  a.drop = adt_make_drop(p, seed, site); a.row_offset = row_offset; a.U = U; a.ldu = ldu;
  a.dX = dX; a.lddx = lddx; a.beta = beta; a.Radd = Radd; a.ldradd = ldradd; a.radd_ids = radd_ids;
  a.dW = dW; a.db = db;
  const int nch = (N + 63) / 64;
  static int cap = 0;
  if (!cap) { const char* e = getenv("ADT_LINBWD_GRID"); cap = e ? atoi(e) : 256; if (cap < 1) cap = 256; }
  const int grid = grid_for(T, BM, cap);
  if (prec == ADT_PREC_F32) launch_linear_bwd<PREC_F32>(a, nch, grid, (hipStream_t)stream);
  else launch_linear_bwd<PREC_BF16>(a, nch, grid, (hipStream_t)stream);
  return check_launch("linear_bwd");
}

int adt_attn_fwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int B,
                 int H, int L, int hd, int causal, float p, const uint32_t* seed, uint32_t site,
                 uint32_t b_offset, float* O, int ldo, float* LSE, uint32_t* mask, void* stream) {
  AttnArgs a{};
  a.mask = mask;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.LSE = LSE;
  a.B = B; a.H = H; a.L = L; a.causal = causal; a.scale = 1.0f / sqrtf((float)hd);
  a.drop = adt_make_drop(p, seed, site); a.bh_offset = b_offset * (uint32_t)H;
  return dispatch_attn(prec, false, hd, a, (hipStream_t)stream);
}

int adt_attn_bwd(int prec, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                 const float* O, int ldo, const float* LSE, const float* dO, int lddo, int B, int H, int L,
                 int hd, int causal, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset,
                 float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, const uint32_t* mask, void* stream) {
  AttnArgs a{};
  a.mask = const_cast<uint32_t*>(mask);
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = const_cast<float*>(O); a.ldo = ldo;
  a.LSE = const_cast<float*>(LSE); a.B = B; a.H = H; a.L = L; a.causal = causal; a.scale = 1.0f / sqrtf((float)hd);
  a.drop = adt_make_drop(p, seed, site); a.bh_offset = b_offset * (uint32_t)H;
  a.dO = dO; a.lddo = lddo; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
  if ((lddo % 4) || (lddq % 4) || (lddk % 4) || (lddv % 4)) return adt_set_error("attn_bwd: ld %% 4");
  if (prec == ADT_PREC_BF16) {      // d = 64: one workgroup per sequence, one natural-order image per tensor (adt_seqattn.cuh)
    const int rc = adt_launch_seq_attn_bwd(hd, a, stream);
    if (rc <= 0) return rc;
  }
  return dispatch_attn(prec, true, hd, a, (hipStream_t)stream);
}

}  // extern "C"

int adt_attn_bwd_saved_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo, const float* LSE,
                            const float* dO, int lddo, int B, int H, int L, int hd, float p, const uint32_t* seed, uint32_t site, uint32_t b_offset,
                            float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, const uint32_t* mask, int out_bf16, void* stream) {
  AttnArgs a{};
  a.mask = const_cast<uint32_t*>(mask);
  a.out_bf16 = out_bf16;
  a.Q = reinterpret_cast<const float*>(Q); a.ldq = ldq; a.K = reinterpret_cast<const float*>(K); a.ldk = ldk;
  a.V = reinterpret_cast<const float*>(V); a.ldv = ldv; a.O = reinterpret_cast<float*>(const_cast<void*>(O)); a.ldo = ldo; a.in_bf16 = 1;
  a.LSE = const_cast<float*>(LSE); a.B = B; a.H = H; a.L = L; a.causal = 1; a.scale = 1.0f / sqrtf((float)hd);
  a.drop = adt_make_drop(p, seed, site); a.bh_offset = b_offset * (uint32_t)H;
  a.dO = dO; a.lddo = lddo; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
  if ((lddo % 4) || (lddq % 4) || (lddk % 4) || (lddv % 4) || (ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8)) return adt_set_error("attn_bwd (bf16 operands): ld");
  const int rc = adt_launch_seq_attn_bwd(hd, a, stream);
  if (rc == 1) return adt_set_error("attn_bwd (bf16 operands): shape L=%d H=%d hd=%d is not covered by the per-sequence kernel", L, H, hd);
  return rc;
}

extern "C" {

int adt_headcls_fwd(const float* O, int ldo, const float* Ws, const float* bs, int B, int L, int H, int hd,
                    float* rec, void* stream) {
  if (H > MAXH || hd % 4) return adt_set_error("headcls: H=%d (max %d) hd=%d", H, MAXH, hd);
  HeadClsArgs a{};
  a.O = O; a.ldo = ldo; a.Ws = Ws; a.bs = bs; a.B = B; a.L = L; a.H = H; a.hd = hd; a.rec = rec;
  hipLaunchKernelGGL(k_headcls_fwd, dim3(grid_for((size_t)B * L * H, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("headcls_fwd");
}

int adt_headcls_bwd(const float* O, int ldo, const float* Ws, const float* rec, const float* drec, int B, int L,
                    int H, int hd, float* dO, int lddo, float* dWs, float* dbs, void* stream) {
  if (H > MAXH || hd % 4) return adt_set_error("headcls: H=%d (max %d) hd=%d", H, MAXH, hd);
  HeadClsArgs a{};
  a.O = O; a.ldo = ldo; a.Ws = Ws; a.B = B; a.L = L; a.H = H; a.hd = hd; a.rec = const_cast<float*>(rec);
  a.drec = drec; a.dO = dO; a.lddo = lddo; a.dWs = dWs; a.dbs = dbs;
  const size_t smem = (size_t)(H * hd + H) * sizeof(float);
  hipLaunchKernelGGL(k_headcls_bwd, dim3(grid_for((size_t)B * L, 16, 1024)), dim3(256), smem, (hipStream_t)stream, a);
  return check_launch("headcls_bwd");
}

int adt_logits_fwd(const float* F, int ldf, const float* E, const int32_t* pos, const int32_t* neg, int T, int d,
                   float* pos_logits, float* neg_logits, void* stream) {
  LogitsArgs a{};
  a.F = F; a.ldf = ldf; a.E = E; a.pos = pos; a.neg = neg; a.T = T; a.d = d; a.pos_logits = pos_logits;
  a.neg_logits = neg_logits;
  hipLaunchKernelGGL(k_logits_fwd, dim3(grid_for(T, 16, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("logits_fwd");
}

int adt_logits_bwd(const float* F, int ldf, const float* E, const int32_t* pos, const int32_t* neg,
                   const float* dpos, const float* dneg, int T, int d, float* dF, int lddf, float* dE,
                   void* stream) {
  LogitsArgs a{};
  a.F = F; a.ldf = ldf; a.E = E; a.pos = pos; a.neg = neg; a.T = T; a.d = d; a.dpos = dpos; a.dneg = dneg;
  a.dF = dF; a.lddf = lddf; a.dE = dE;
  hipLaunchKernelGGL(k_logits_bwd, dim3(grid_for(T, 16, 2048)), dim3(256), 0, (hipStream_t)stream, a);
  ScatterArgs sc{};
  sc.G = F; sc.ldg = ldf; sc.T = T; sc.d = d; sc.scale = 1.0f; sc.drop = adt_make_drop(0.f, nullptr, 0); sc.dE = dE;
  sc.ids = pos; sc.rowscale = dpos;
  hipLaunchKernelGGL(k_item_scatter, dim3(grid_for(T, 4, 2048)), dim3(256), 0, (hipStream_t)stream, sc);
  sc.ids = neg; sc.rowscale = dneg;
  hipLaunchKernelGGL(k_item_scatter, dim3(grid_for(T, 4, 2048)), dim3(256), 0, (hipStream_t)stream, sc);
  return check_launch("logits_bwd");
}

int adt_bce_seed(const float* pos_logits, const float* neg_logits, const int32_t* pos, int T, const float* norms,
                 float* dpos, float* dneg, float* loss2, void* stream) {
  BceArgs a{pos_logits, neg_logits, pos, T, norms, dpos, dneg, loss2};
  hipLaunchKernelGGL(k_bce, dim3(grid_for(T, 256, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("bce_seed");
}

int adt_mse_seed(const float* A, const float* Bm, int64_t n, float lambda, const float* norms, float* GA,
                 int accumulate_a, float* GB, float* loss1, void* stream) {
  if (n % 4) return adt_set_error("mse_seed: n %% 4");
  MseArgs a{A, Bm, (size_t)n, lambda, norms, GA, accumulate_a, GB, loss1};
  hipLaunchKernelGGL(k_mse_seed, dim3(grid_for((size_t)n / 4, 256, 512)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("mse_seed");
}

int adt_nll_seed(const float* rec, int n_rows, int H, float lambda2, const float* norms, float* drec,
                 float* loss1, void* stream) {
  NllArgs a{rec, n_rows, H, lambda2, norms, drec, loss1};
  hipLaunchKernelGGL(k_nll_seed, dim3(grid_for((size_t)n_rows * H * H, 256, 512)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("nll_seed");
}

// the executor's loss assembly in one launch (adt_host.h); nmse, nnll <= 4
int adt_loss_seeds(const float* pos_logits, const float* neg_logits, const int32_t* pos, int T, const float* norms, float* dpos, float* dneg,
                   float* loss_bce, int nmse, const float* const* A, const float* const* Bm, int64_t n, const float* lambdas, float* const* GA,
                   int accumulate_a, float* const* GB, float* const* loss_mse, int nnll, const float* const* rec, int n_rows, int H, float lambda2,
                   float* const* drec, float* const* loss_nll, void* stream) {
  return adt_loss_seeds_prefetch(pos_logits, neg_logits, pos, T, norms, dpos, dneg, loss_bce, nmse, A, Bm, n, lambdas, GA, accumulate_a, GB, loss_mse, nnll,
                                 rec, n_rows, H, lambda2, drec, loss_nll, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, stream);
}
// The logits / BCE / item-row pass of the deferred path (adt_logits_bce_scatter) as the first workgroups of the NEXT adt_loss_seeds_prefetch launch of
// this host thread (one launch, no second stream: adt_sasrec.hip, forward_loss_lean).
static thread_local LogitsBceArgs g_lb_job;
static thread_local bool g_lb_set = false;
void adt_loss_seeds_attach_logits(const float* F, const float* E, const int32_t* pos, const int32_t* neg, const float* norms, int T, float* pos_logits,
                                  float* neg_logits, float* dpos, float* dneg, float* loss_bce, float* dF, float* rep, int nrep, int64_t rep_stride,
                                  int neg_only) {
  g_lb_job = LogitsBceArgs{F, E, pos, neg, norms, T, pos_logits, neg_logits, dpos, dneg, loss_bce, dF, rep, nrep, (size_t)rep_stride, neg_only};
  g_lb_set = true;
}

int adt_loss_seeds_prefetch(const float* pos_logits, const float* neg_logits, const int32_t* pos, int T, const float* norms, float* dpos, float* dneg,
                            float* loss_bce, int nmse, const float* const* A, const float* const* Bm, int64_t n, const float* lambdas, float* const* GA,
                            int accumulate_a, float* const* GB, float* const* loss_mse, int nnll, const float* const* rec, int n_rows, int H, float lambda2,
                            float* const* drec, float* const* loss_nll, const int32_t* ring, int64_t slot_ints, int nslots, int64_t n_ints,
                            uint32_t* state, uint32_t* consumed, int32_t* staging, void* stream) {
  if (nmse > 4 || nnll > 4 || n % 4) return adt_set_error("loss_seeds: at most 4 + 4 terms, n %% 4");
  LossSeedsArgs a{};
  if (ring && staging && state) {
    a.pf = RingPrefetchArgs{ring, (size_t)slot_ints, nslots, (size_t)n_ints, state, consumed, staging, 0, g_pf_split ? 2 : 0};
    a.gp = 32;      // one round of 8 x 16-byte loads per thread covers the flagship batch (819 KB): a PCIe read wants everything in flight at once
    if (g_pf_split) {      // ... its second half rides on the next adt_embed_bwd3 launch of this host thread
      g_pf_job = a.pf; g_pf_job.part = 1; g_pf_set = true; g_pf_split = false;
    }
  }
  a.bce = BceArgs{pos_logits, neg_logits, pos, T, norms, dpos, dneg, loss_bce};
  // (a seed that is not materialised -- GA[i] or GB[i] == nullptr -- leaves its coefficient at norms[8 + i] for its consumer)
  for (int i = 0; i < nmse; ++i)
    a.mse[i] = MseArgs{A[i], Bm[i], (size_t)n, lambdas[i], norms, GA[i], accumulate_a, GB[i], loss_mse[i], (!GA[i] || !GB[i]) ? const_cast<float*>(norms) + 8 + i : nullptr};
  for (int i = 0; i < nnll; ++i) a.nll[i] = NllArgs{rec[i], n_rows, H, lambda2, norms, drec[i], loss_nll[i]};
  a.nmse = nmse; a.nnll = nnll;
  if (g_lb_set) { a.lb = g_lb_job; a.gl = grid_for(a.lb.T, 4 * 16, 1024); g_lb_set = false; }
  a.gb = pos_logits ? grid_for(T, 256, 256) : 0;      // no logits: the BCE seed is formed elsewhere (adt_logits_bce_scatter)
  a.gm = nmse ? grid_for((size_t)n / 4, 256, 512) : 1;
  a.gn = nnll ? grid_for((size_t)n_rows * H * H, 256, 512) : 1;
  hipLaunchKernelGGL(k_loss_seeds, dim3(a.gl + a.gb + nmse * a.gm + nnll * a.gn + a.gp), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("loss_seeds");
}

// ---- deterministic item-table / positional-table gradient (adt_itemgrad.cuh) -----------------------------------------------------------
namespace {
struct ItemWork { int64_t hist, base, perm, pitem, pmeta, prow, pcoef, cflag, carry, total; int nblk; };
ItemWork item_work(int nsrc, int T, int V1) {
  ItemWork w;
  const int64_t N = (int64_t)nsrc * T;
  auto up = [](int64_t x) { return (x + 63) / 64 * 64; };
  w.nblk = (int)((N + IG_PER_BLOCK - 1) / IG_PER_BLOCK);
  int64_t o = 0;
  w.hist = o; o += up((int64_t)IS_NCH * V1);
  w.base = o; o += up(V1 + 1 + 256);      // + the totals of the 64-item groups
  w.perm = o; o += up(N);
  w.pitem = o; o += up(N);
  w.pmeta = o; o += up(N);
  w.prow = o; o += up(2 * N);          // 64-bit words
  w.pcoef = o; o += up(2 * N);
  w.cflag = o; o += up((int64_t)w.nblk * 4);
  w.carry = o; o += up((int64_t)w.nblk * 2 * 64);
  w.total = o;
  return w;
}
}  // namespace

int adt_item_sort_supported(int V1) { return V1 >= 2 && V1 <= IS_MAXV1 ? 1 : 0; }      // (and at most 262,144 entries: adt_item_sort checks)
int64_t adt_item_sort_work_ints(int nsrc, int T, int V1) { return item_work(nsrc, T, V1).total; }

/* kind[s] 0: rows[s] = gradient of an embedding layer's output (T x 64), summed as rows * emb_scale * keep / (1 - p) with the forward's dropout
 * decisions (element index (t + row_offset) * 64 + f) ; 1: rows[s] * coef[s][t].  The pointers are only recorded here (the gather plan). */
int adt_item_sort(const int32_t* const* ids, int nsrc, int T, int V1, const float* const* rows, const float* const* coef, const int* kind,
                  uint32_t row_offset, int32_t* work, void* stream) {
  if (nsrc < 1 || nsrc > 4 || T < 1 || !adt_item_sort_supported(V1)) return adt_set_error("item_sort: nsrc %d, T %d, %d items + 1", nsrc, T, V1);
  if (((uintptr_t)work & 7) != 0) return adt_set_error("item_sort: work must be 8-byte aligned");
  const ItemWork w = item_work(nsrc, T, V1);
  ItemSortArgs a{};
  for (int s = 0; s < 4; ++s) {
    const int k = s < nsrc ? s : 0;
    a.ids[s] = ids[k]; a.rows[s] = rows[k]; a.coef[s] = coef ? coef[k] : nullptr; a.kind[s] = kind[k];
    if (s < nsrc && (!rows[s] || (kind[s] == 1 && (!coef || !coef[s])))) return adt_set_error("item_sort: source %d has no rows", s);
  }
  a.nsrc = nsrc; a.T = T; a.V1 = V1; a.hist = work + w.hist; a.base = work + w.base; a.bsum = a.base + V1 + 1; a.perm = work + w.perm; a.pitem = work + w.pitem;
  a.row_offset = row_offset; a.pmeta = reinterpret_cast<uint32_t*>(work + w.pmeta);
  a.prow = reinterpret_cast<uint64_t*>(work + w.prow); a.pcoef = reinterpret_cast<uint64_t*>(work + w.pcoef);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)V1 * sizeof(int);
  if ((int64_t)nsrc * T > (int64_t)IS_NCH * 64 * IS_MAXR) return adt_set_error("item_sort: %lld entries > %d", (long long)nsrc * T, IS_NCH * 64 * IS_MAXR);
  hipLaunchKernelGGL(k_isort_hist, dim3(IS_NCH), dim3(64), lds, s, a);
  hipLaunchKernelGGL(k_isort_scan_chunks, dim3((V1 + 63) / 64), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_isort_place, dim3(IS_NCH), dim3(64), lds + 256 * sizeof(int), s, a);
  return check_launch("item_sort");
}

/* dE[item] (accumulate ? += : =) sum over the sorted entries of `item` that belong to the sources in src_mask (rows of items without entries are not touched). */
static int item_seg_args(ItemSegArgs& a, const int32_t* work, int nsrc, int T, int V1, uint32_t src_mask, const uint32_t* site, float p,
                         const uint32_t* seed, float emb_scale, float* dE, int accumulate) {
  if (nsrc < 1 || nsrc > 4 || !adt_item_sort_supported(V1)) return adt_set_error("item_segsum: nsrc %d, %d items + 1", nsrc, V1);
  const ItemWork w = item_work(nsrc, T, V1);
  a.pitem = work + w.pitem; a.total = work + w.base + V1; a.nblk = w.nblk;
  a.pmeta = reinterpret_cast<const uint32_t*>(work + w.pmeta);
  a.prow = reinterpret_cast<const uint64_t*>(work + w.prow); a.pcoef = reinterpret_cast<const uint64_t*>(work + w.pcoef);
  for (int s = 0; s < 4; ++s) a.site[s] = site ? site[s < nsrc ? s : 0] : 0u;
  a.src_mask = src_mask & ((1u << nsrc) - 1u);
  const DropCfg dc = adt_make_drop(p, seed, 0);
  a.seed = seed; a.thr = dc.thr; a.dscale = dc.scale; a.emb_scale = emb_scale;
  a.dE = dE; a.rmw = accumulate ? 1 : 0;
  a.carry = reinterpret_cast<float*>(const_cast<int32_t*>(work) + w.carry); a.cflag = const_cast<int32_t*>(work) + w.cflag;
  return 0;
}
static int pos_sum_args(PosSumArgs& a, const int32_t* const* ids, const float* const* dX, const uint32_t* site, int nsrc, int B, int L, float p,
                        const uint32_t* seed, uint32_t row_offset, float* dP) {
  if (nsrc < 1 || nsrc > 2) return adt_set_error("posemb_sum: nsrc %d", nsrc);
  for (int s = 0; s < 2; ++s) { const int k = s < nsrc ? s : 0; a.ids[s] = ids[k]; a.dX[s] = dX[k]; a.site[s] = site[k]; }
  a.nsrc = nsrc; a.B = B; a.L = L;
  const DropCfg dc = adt_make_drop(p, seed, 0);
  a.seed = seed; a.thr = dc.thr; a.dscale = dc.scale; a.row_offset = row_offset; a.dP = dP;
  return 0;
}

int adt_item_segsum(const int32_t* work, int nsrc, int T, int V1, uint32_t src_mask, const uint32_t* site, float p, const uint32_t* seed,
                    float emb_scale, float* dE, int accumulate, void* stream) {
  ItemSegArgs a{};
  if (item_seg_args(a, work, nsrc, T, V1, src_mask, site, p, seed, emb_scale, dE, accumulate)) return 1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_item_segsum, dim3(a.nblk), dim3(IG_WAVES * 64), 0, s, a);
  hipLaunchKernelGGL(k_item_carry, dim3(a.nblk), dim3(64), 0, s, a);
  return check_launch("item_segsum");
}

/* dP[l] += sum_b [ids != 0] keep / (1 - p) dX[b, l] for nsrc (1 or 2) embedding layers, b ascending */
int adt_posemb_sum(const int32_t* const* ids, const float* const* dX, const uint32_t* site, int nsrc, int B, int L, float p, const uint32_t* seed,
                   uint32_t row_offset, float* dP, void* stream) {
  PosSumArgs a{};
  if (pos_sum_args(a, ids, dX, site, nsrc, B, L, p, seed, row_offset, dP)) return 1;
  hipLaunchKernelGGL(k_posemb_sum, dim3(L), dim3(PS_WAVES * 64), 0, (hipStream_t)stream, a);
  return check_launch("posemb_sum");
}

/* adt_item_segsum + adt_posemb_sum (same p / seed) in one launch + the carry launch */
int adt_item_segsum_posemb(const int32_t* work, int nsrc, int T, int V1, uint32_t src_mask, const uint32_t* site, float p, const uint32_t* seed,
                           float emb_scale, float* dE, int accumulate, const int32_t* const* pos_ids, const float* const* pos_dX,
                           const uint32_t* pos_site, int pos_nsrc, int B, int L, uint32_t row_offset, float* dP, void* stream) {
  ItemSegArgs a{};
  PosSumArgs ps{};
  if (item_seg_args(a, work, nsrc, T, V1, src_mask, site, p, seed, emb_scale, dE, accumulate)) return 1;
  if (pos_sum_args(ps, pos_ids, pos_dX, pos_site, pos_nsrc, B, L, p, seed, row_offset, dP)) return 1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_item_segsum_posemb, dim3(a.nblk + L), dim3(IG_WAVES * 64), 0, s, a, ps);
  hipLaunchKernelGGL(k_item_carry, dim3(a.nblk), dim3(64), 0, s, a);
  return check_launch("item_segsum_posemb");
}

int adt_clip_adam(float* P, float* G, float* M, float* V, int64_t n, int64_t nE, float wd, float clip, float lr,
                  float b1, float b2, float eps, float grad_scale, float* scal, void* stream) {
  OptArgs a{};
  a.P = P; a.G = G; a.M = M; a.Vv = V; a.n = (size_t)n; a.nE = (size_t)nE; a.wd = wd; a.clip = clip; a.lr = lr;
  a.b1 = b1; a.b2 = b2; a.eps = eps; a.scal = scal; a.grad_scale = grad_scale;
  hipStream_t s = (hipStream_t)stream;
  if (adt::zero_f32_async(scal + 64, 128, s)) return adt_set_error("clip_adam: zero");
  if (wd != 0.f && nE > 0) hipLaunchKernelGGL(k_sumsq, dim3(grid_for((size_t)nE, 256, 256)), dim3(256), 0, s, (const float*)P, (size_t)nE, scal + 64);
  hipLaunchKernelGGL(k_wd_gradnorm, dim3(grid_for((size_t)n, 256, 512)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_adam, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, s, a);
  return check_launch("clip_adam");
}

/* adt_clip_adam for a step opened by adt_sasrec_step_begin: scal[64..128) already holds the partial sums of ||E||^2 and scal[128..192) is zero */
int adt_clip_adam_pre(float* P, float* G, float* M, float* V, int64_t n, int64_t nE, float wd, float clip, float lr,
                      float b1, float b2, float eps, float grad_scale, float* scal, void* stream) {
  OptArgs a{};
  a.P = P; a.G = G; a.M = M; a.Vv = V; a.n = (size_t)n; a.nE = (size_t)nE; a.wd = wd; a.clip = clip; a.lr = lr;
  a.b1 = b1; a.b2 = b2; a.eps = eps; a.scal = scal; a.grad_scale = grad_scale;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_wd_gradnorm, dim3(grid_for((size_t)n, 256, 512)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_adam, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, s, a);
  return check_launch("clip_adam_pre");
}

/* fold (d0 += replicas r0, d1 += replicas r1) + weight decay on job 0 + ||g||^2 partials, then Adam: adt_replica_reduce2 + adt_clip_adam_pre in two launches.
 * d0 must be the item table's gradient (G + 0, n0 = nE floats); the padding between the two ranges holds zeros and is left alone. */
int adt_fold_clip_adam(float* P, float* G, float* M, float* V, int64_t n, float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1,
                       const float* r1, int64_t n1, int nrep1, int64_t s1, float wd, float clip, float lr, float b1, float b2, float eps, float* scal,
                       void* stream) {
  if (n0 <= 0 || n1 <= 0 || (n0 % 4) || (s0 % 4) || (n1 % 4) || (s1 % 4) || d0 != G) return adt_set_error("fold_clip_adam: ranges");
  OptArgs a{};
  a.P = P; a.G = G; a.M = M; a.Vv = V; a.n = (size_t)n; a.nE = (size_t)n0; a.wd = wd; a.clip = clip; a.lr = lr;
  a.b1 = b1; a.b2 = b2; a.eps = eps; a.scal = scal; a.grad_scale = 1.0f;
  RepReduce2Args r{{d0, d1}, {r0, r1}, {(size_t)n0, (size_t)n1}, {nrep0, nrep1}, {(size_t)s0, (size_t)s1}, 0};
  const int g0 = grid_for((size_t)n0 / 4, 256, 1024), g1 = grid_for((size_t)n1 / 4, 256, 1024);
  r.g0 = g0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fold_wd_gradnorm, dim3(g0 + g1), dim3(256), 0, s, r, a);
  hipLaunchKernelGGL(k_adam, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, s, a);
  return check_launch("fold_clip_adam");
}

/* adt_fold_clip_adam that also sums the per-sequence partials of the 64 x 64 weight gradients (nslots blocks: slot index inside a workgroup's
 * partial area, float offset in G) in workgroup order; mask_base = offset in G of the range the row mask describes (the positional table). */
static int fold_parts_impl(bool adam, float* P, float* G, float* M, float* V, int64_t n, float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1,
                           const float* r1, int64_t n1, int nrep1, int64_t s1, const float* part, int64_t part_stride, const int* nwg_slot, const int* slots,
                           const int* offs, int nslots, const float* vpart, const int* vsrc, const int* vnwg, const int* vstride, const int* voff, int nvec,
                           float* gn_part, float wd, float clip, float lr, float b1, float b2, float eps, float* scal, void* stream) {
  if (n0 <= 0 || n1 <= 0 || (n0 % 4) || (s0 % 4) || (n1 % 4) || (s1 % 4) || d0 != G) return adt_set_error("fold_parts_clip_adam: ranges");
  if (nslots < 1 || nslots > FP_MAXSLOTS || (d1 - G) % 64 || n1 > (int64_t)FP_MASKWORDS * 32 * 64) return adt_set_error("fold_parts_clip_adam: %d blocks, %lld floats", nslots, (long long)n1);
  OptArgs a{};
  a.P = P; a.G = G; a.M = M; a.Vv = V; a.n = (size_t)n; a.nE = (size_t)n0; a.wd = wd; a.clip = clip; a.lr = lr;
  a.b1 = b1; a.b2 = b2; a.eps = eps; a.scal = scal; a.grad_scale = 1.0f; a.fold_only = adam ? 0 : 1;
  RepReduce2Args r{{d0, d1}, {r0, r1}, {(size_t)n0, (size_t)n1}, {nrep0, nrep1}, {(size_t)s0, (size_t)s1}, 0};
  const int g0 = grid_for((size_t)n0 / 4, 256, 1024), g1 = grid_for((size_t)n1 / 4, 256, 1024);
  r.g0 = g0;
  PartFoldArgs pf{};
  pf.part = part; pf.stride = (size_t)part_stride; pf.nslots = nslots; pf.mask_base = d1 - G;
  for (int i = 0; i < nslots; ++i) {
    pf.slot[i] = slots[i]; pf.off[i] = offs[i]; pf.nwg[i] = nwg_slot[i];
    const int64_t r0w = (offs[i] - pf.mask_base) / 64;
    if (offs[i] < pf.mask_base || (offs[i] - pf.mask_base) % 64 || r0w + 64 > (int64_t)FP_MASKWORDS * 32) return adt_set_error("fold_parts_clip_adam: block %d outside the masked range", i);
    for (int64_t rr = r0w; rr < r0w + 64; ++rr) pf.rowmask[rr >> 5] |= 1u << (rr & 31);
  }
  if (nvec < 0 || nvec > FV_MAXCHUNKS) return adt_set_error("fold_parts_clip_adam: %d vector chunks", nvec);
  VecFoldArgs vf{};
  vf.vpart = vpart; vf.n = vpart ? nvec : 0;
  for (int i = 0; i < vf.n; ++i) {      // the vector chunks' rows belong to job 3: job 1 leaves them alone
    vf.src[i] = vsrc[i]; vf.nwg[i] = vnwg[i]; vf.stride[i] = vstride[i]; vf.off[i] = voff[i];
    const int64_t rr = (voff[i] - pf.mask_base) / 64;
    if (voff[i] < pf.mask_base || (voff[i] - pf.mask_base) % 64 || rr >= (int64_t)FP_MASKWORDS * 32) return adt_set_error("fold_parts_clip_adam: vector chunk %d outside the masked range", i);
    pf.rowmask[rr >> 5] |= 1u << (rr & 31);
  }
  const int g2 = 8 * nslots, g3 = vf.n;
  if (gn_part) { a.gn_part = gn_part; a.gn_n = g0 + g1 + g2 + g3; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fold_parts_gradnorm, dim3(g0 + g1 + g2 + g3), dim3(256), 0, s, r, a, pf, vf, g1, g2);
  if (adam) hipLaunchKernelGGL(k_adam, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, s, a);
  return check_launch("fold_parts_clip_adam");
}

int adt_fold_parts_clip_adam(float* P, float* G, float* M, float* V, int64_t n, float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1,
                             const float* r1, int64_t n1, int nrep1, int64_t s1, const float* part, int64_t part_stride, const int* nwg_slot, const int* slots,
                             const int* offs, int nslots, const float* vpart, const int* vsrc, const int* vnwg, const int* vstride, const int* voff, int nvec,
                             float* gn_part, float wd, float clip, float lr, float b1, float b2, float eps, float* scal, void* stream) {
  return fold_parts_impl(true, P, G, M, V, n, d0, r0, n0, nrep0, s0, d1, r1, n1, nrep1, s1, part, part_stride, nwg_slot, slots, offs, nslots, vpart, vsrc, vnwg,
                         vstride, voff, nvec, gn_part, wd, clip, lr, b1, b2, eps, scal, stream);
}

/* the fold half alone: the gradient sums into G (no weight-decay term, no optimizer step) -- in front of a gradient all-reduce */
int adt_fold_parts(float* P, float* G, int64_t n, float* d0, const float* r0, int64_t n0, int nrep0, int64_t s0, float* d1, const float* r1, int64_t n1,
                   int nrep1, int64_t s1, const float* part, int64_t part_stride, const int* nwg_slot, const int* slots, const int* offs, int nslots,
                   const float* vpart, const int* vsrc, const int* vnwg, const int* vstride, const int* voff, int nvec, float* gn_part, float* scal,
                   void* stream) {
  return fold_parts_impl(false, P, G, nullptr, nullptr, n, d0, r0, n0, nrep0, s0, d1, r1, n1, nrep1, s1, part, part_stride, nwg_slot, slots, offs, nslots, vpart,
                         vsrc, vnwg, vstride, voff, nvec, gn_part, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, scal, stream);
}

static void step_begin_extras(StepBeginArgs& a, float* Z, int64_t nz, const float* pack_base, void* pack_img, const int* pack_offs, int npack) {
  a.Z = Z; a.nz = Z ? (size_t)nz : 0;
  a.pk.base = pack_base; a.pk.img = reinterpret_cast<__bf16*>(pack_img); a.pk.n = (pack_base && pack_img && pack_offs) ? npack : 0;
  for (int i = 0; i < a.pk.n; ++i) a.pk.off[i] = pack_offs[i];
}

int adt_step_begin_launch(uint32_t* seed, uint32_t inc, float* norms_dst, const float* norms_src, float* loss, int nloss, float* scal, float* G,
                          int64_t n, const float* E, int64_t nE, float* Z, int64_t nz, const float* pack_base, void* pack_img, const int* pack_offs,
                          int npack, void* stream) {
  if (npack < 0 || npack > 256 || (nz & 3)) return adt_set_error("step_begin: %d weight blocks, %lld floats to zero", npack, (long long)nz);
  StepBeginArgs a{seed, inc, norms_dst, norms_src, loss, nloss, scal, G, (size_t)n, E, (size_t)nE, nullptr, 0, 0, nullptr, 0, nullptr, nullptr};
  step_begin_extras(a, Z, nz, pack_base, pack_img, pack_offs, npack);
  hipLaunchKernelGGL(k_step_begin, dim3(256 + a.pk.n), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("step_begin");
}

int adt_step_begin_ring_launch(uint32_t* seed, uint32_t inc, float* norms_dst, float* loss, int nloss, float* scal, float* G, int64_t n, const float* E,
                               int64_t nE, const int32_t* ring, int64_t slot_ints, int nslots, int32_t* ids_dst, int64_t n_ints, uint32_t* state,
                               uint32_t* consumed, const int32_t* staging, const uint32_t* produced, float* Z, int64_t nz, const float* pack_base,
                               void* pack_img, const int* pack_offs, int npack, void* stream) {
  if (!ring || !ids_dst || !state || nslots < 1 || n_ints < 4 || (n_ints & 3) || slot_ints < n_ints || (slot_ints & 3))
    return adt_set_error("step_begin_ring: ring %p, %d slots of %lld ints, %lld ints per step (multiples of 4)", (const void*)ring, nslots, (long long)slot_ints, (long long)n_ints);
  if (npack < 0 || npack > 256 || (nz & 3)) return adt_set_error("step_begin_ring: %d weight blocks, %lld floats to zero", npack, (long long)nz);
  StepBeginArgs a{seed, inc, norms_dst, nullptr, loss, nloss, scal, G, (size_t)n, E, (size_t)nE, ring, (size_t)slot_ints, nslots, ids_dst, (size_t)n_ints, state, consumed,
                  staging, produced};
  step_begin_extras(a, Z, nz, pack_base, pack_img, pack_offs, npack);
  hipLaunchKernelGGL(k_step_begin, dim3(256 + a.pk.n), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("step_begin_ring");
}

int adt_clip_adam_l2(float* P, float* G, float* M, float* V, int64_t n, float l2, float clip, float lr, float b1, float b2,
                     float eps, float grad_scale, float* scal, void* stream) {
  OptArgs a{};
  a.P = P; a.G = G; a.M = M; a.Vv = V; a.n = (size_t)n; a.nE = 0; a.wd = 0.f; a.clip = clip; a.lr = lr;
  a.b1 = b1; a.b2 = b2; a.eps = eps; a.scal = scal; a.grad_scale = grad_scale; a.l2 = l2;
  hipStream_t s = (hipStream_t)stream;
  if (adt::zero_f32_async(scal + 64, 128, s)) return adt_set_error("clip_adam_l2: zero");
  hipLaunchKernelGGL(k_wd_gradnorm, dim3(grid_for((size_t)n, 256, 512)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_adam, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, s, a);
  return check_launch("clip_adam_l2");
}

int adt_score_rank_bias(const float* F, int ldf, const float* E, const float* bias, const int32_t* cand, int B, int C, int d,
                        float* logits, int32_t* rank, void* stream) {
  ScoreArgs a{F, ldf, E, cand, B, C, d, logits, rank, bias};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_score, dim3(grid_for((size_t)B * C, 16, 4096)), dim3(256), 0, s, a);
  if (rank) hipLaunchKernelGGL(k_rank, dim3(grid_for(B, 4, 1024)), dim3(256), 0, s, a);
  return check_launch("score_rank_bias");
}

int adt_score_rank(const float* F, int ldf, const float* E, const int32_t* cand, int B, int C, int d,
                   float* logits, int32_t* rank, void* stream) {
  ScoreArgs a{F, ldf, E, cand, B, C, d, logits, rank, nullptr};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_score, dim3(grid_for((size_t)B * C, 16, 4096)), dim3(256), 0, s, a);
  if (rank) hipLaunchKernelGGL(k_rank, dim3(grid_for(B, 4, 1024)), dim3(256), 0, s, a);
  return check_launch("score_rank");
}

}  // extern "C"
