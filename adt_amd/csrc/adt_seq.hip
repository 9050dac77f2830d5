// Launchers of the per-sequence fused layer kernels (adt_seqfwd.cuh, adt_seqbwd.cuh): one workgroup per user sequence.
#include "adt_host.h"
#include <stdlib.h>
#include "adt_seqfwd.cuh"
#include "adt_seqfwd_tt.cuh"
#include "adt_seqattn.cuh"
#include "adt_seqbwd_tt.cuh"
#include "adt_seqpost_tt.cuh"

using namespace adt;

static int seq_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return adt_set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

static int seq_launch(const void* fn, size_t smem, bool& attr_done, int grid, const void* args_ptr, hipStream_t s, const char* what, int nwaves = SQ_NW) {
  if (!attr_done) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return adt_set_error("%s: hipFuncSetAttribute(%zu)", what, smem);
    attr_done = true;
  }
  void* kargs[] = {const_cast<void*>(args_ptr)};
  if (hipLaunchKernel(fn, dim3(grid), dim3(nwaves * 64), kargs, smem, s) != hipSuccess) return adt_set_error("%s: launch failed", what);
  return seq_check(what);
}

// ---- pre-packed weight images (adt_wave.cuh: WPack) -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_wimg(PackArgs a) { pack_wimg_block(a, blockIdx.x); }      // adt_wave.cuh

extern "C" int adt_pack_wimg(const float* base, void* img, const int* offs, int n, void* stream) {
  if (n < 1 || n > 256) return adt_set_error("pack_wimg: %d blocks", n);
  PackArgs a;
  a.base = base; a.img = reinterpret_cast<__bf16*>(img); a.n = n;
  for (int i = 0; i < n; ++i) a.off[i] = offs[i];
  hipLaunchKernelGGL(k_pack_wimg, dim3(n), dim3(256), 0, (hipStream_t)stream, a);
  return seq_check("pack_wimg");
}

int adt_seq_supported(int prec, int L, int d, int hd) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_SEQ"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on && prec == ADT_PREC_BF16 && d == 64 && L <= SQ_LP && (hd == 16 || hd == 32 || hd == 64);
}

static int env_on(const char* name) { const char* e = getenv(name); return (e && atoi(e) == 0) ? 0 : 1; }

// Lean saved tensors: the transposed-chain forward writes o, h, u, a1, q2, kv2, o2 as bf16 rows and does not write LN(x) / qkv at all (the fused
// block backward recomputes them from x).  Only when every kernel of the backward that reads them is the per-sequence / flag-aware one:
// the caller must then treat a "not covered" (1) from adt_launch_seq_attn_pre_bwd / adt_launch_seq_attn_bwd as an error.  ADT_SEQ_LEAN=0: off.
int adt_seq_lean(int prec, int L, int d, int hd) {
  static int on = -1;
  if (on < 0) on = env_on("ADT_SEQ_LEAN") && env_on("ADT_SEQ_TT") && env_on("ADT_SEQ_BWD") && env_on("ADT_SEQ_ATTN_BWD") && !getenv("ADT_SEQ_ABLATE");
  return on && adt_seq_supported(prec, L, d, hd) && (L & 3) == 0 && L <= SB_R && L <= 224 && sab_lds_bytes(L, 64 / hd) <= 160 * 1024;
}

static unsigned long long* g_stamps = nullptr;

extern "C" int adt_seq_stamps_read(unsigned long long* out, int n) {      // debugging aid of tools/seq_stamps.py, not part of include/adt_hip.h
  if (!g_stamps) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpy(out, g_stamps, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

static unsigned long long* seq_stamp_buffer(bool for_attention) {      // ADT_SEQ_STAMPS=1: forward kernels; =2: the attention backward
  static int st = -1;
  if (st < 0) {
    const char* e = getenv("ADT_SEQ_STAMPS");
    st = e ? atoi(e) : 0;
    if (st && hipMalloc(&g_stamps, 16 * 16 * sizeof(unsigned long long)) != hipSuccess) st = 0;
  }
  return (st == (for_attention ? 2 : 1)) ? g_stamps : nullptr;
}
static unsigned long long* seq_stamp_buffer_post() { seq_stamp_buffer(false); const char* e = getenv("ADT_SEQ_STAMPS"); return (e && atoi(e) == 3) ? g_stamps : nullptr; }

static void seq_ablate(SeqFwdArgs& a) {
  a.stamps = seq_stamp_buffer(false);
  static int ab = -1;
  if (ab < 0) { const char* e = getenv("ADT_SEQ_ABLATE"); ab = e ? atoi(e) : 0; }
  a.ablate = ab;
  if (ab & 1) { a.qkv = nullptr; a.o = nullptr; a.h = nullptr; a.u = nullptr; a.a1 = nullptr; a.q2 = nullptr; a.kv2 = nullptr; a.o2 = nullptr; a.mask = nullptr; a.mask2 = nullptr; }
}

static bool seq_use_tt(const SeqFwdArgs& a) {     // register-resident transposed chains (adt_seqfwd_tt.cuh); ADT_SEQ_TT=0: the row-major fused form
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_SEQ_TT"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on && a.wp_img != nullptr && (a.L & 3) == 0;      // L % 4 == 0: a quad of attention keys shares one dropout hash word
}

int adt_launch_seq_enc_fwd(int hd, const SeqFwdArgs& a, void* stream) {
  static bool done[3] = {false, false, false};
  static bool done_tt[3] = {false, false, false};
  SeqFwdArgs args = a;
  seq_ablate(args);
  hipStream_t s = (hipStream_t)stream;
  if (seq_use_tt(args)) {
    const int nsp = args.nsplit > 1 ? args.nsplit : 1;
    const size_t smem_tt = SeqTtLds<6>::bytes;
    if (hd == 64) return seq_launch((const void*)k_seqtt_enc_fwd<64>, smem_tt, done_tt[0], a.B * nsp, &args, s, "seqtt_enc_fwd<64>", TQ_FWD_NW);
    if (hd == 32) return seq_launch((const void*)k_seqtt_enc_fwd<32>, smem_tt, done_tt[1], a.B * nsp, &args, s, "seqtt_enc_fwd<32>", TQ_FWD_NW);
    if (hd == 16) return seq_launch((const void*)k_seqtt_enc_fwd<16>, smem_tt, done_tt[2], a.B * nsp, &args, s, "seqtt_enc_fwd<16>", TQ_FWD_NW);
  }
  args.nsplit = 0;
  const size_t smem = SeqFwdLds<6>::bytes;
  if (hd == 64) return seq_launch((const void*)k_seq_enc_fwd<64, 2>, smem, done[0], a.B, &args, s, "seq_enc_fwd<64>");
  if (hd == 32) return seq_launch((const void*)k_seq_enc_fwd<32, 2>, smem, done[1], a.B, &args, s, "seq_enc_fwd<32>");
  if (hd == 16) return seq_launch((const void*)k_seq_enc_fwd<16, 4>, smem, done[2], a.B, &args, s, "seq_enc_fwd<16>");
  return adt_set_error("seq_enc_fwd: head size %d", hd);
}

int adt_launch_seq_dec_fwd(int hd, const SeqFwdArgs& a, void* stream) {
  static bool done[3] = {false, false, false};
  static bool done_tt[3] = {false, false, false};
  SeqFwdArgs args = a;
  seq_ablate(args);
  hipStream_t s = (hipStream_t)stream;
  if (seq_use_tt(args)) {
    const int nsp = args.nsplit > 1 ? args.nsplit : 1;
    const size_t smem_tt = SeqTtLds<5>::bytes;
    if (hd == 64) return seq_launch((const void*)k_seqtt_dec_fwd<64>, smem_tt, done_tt[0], a.B * nsp, &args, s, "seqtt_dec_fwd<64>", TQ_FWD_NW);
    if (hd == 32) return seq_launch((const void*)k_seqtt_dec_fwd<32>, smem_tt, done_tt[1], a.B * nsp, &args, s, "seqtt_dec_fwd<32>", TQ_FWD_NW);
    if (hd == 16) return seq_launch((const void*)k_seqtt_dec_fwd<16>, smem_tt, done_tt[2], a.B * nsp, &args, s, "seqtt_dec_fwd<16>", TQ_FWD_NW);
  }
  args.nsplit = 0;
  const size_t smem = SeqFwdLds<5>::bytes;
  if (hd == 64) return seq_launch((const void*)k_seq_dec_fwd<64>, smem, done[0], a.B, &args, s, "seq_dec_fwd<64>");
  if (hd == 32) return seq_launch((const void*)k_seq_dec_fwd<32>, smem, done[1], a.B, &args, s, "seq_dec_fwd<32>");
  if (hd == 16) return seq_launch((const void*)k_seq_dec_fwd<16>, smem, done[2], a.B, &args, s, "seq_dec_fwd<16>");
  return adt_set_error("seq_dec_fwd: head size %d", hd);
}

// causal attention backward, one workgroup per sequence (adt_seqattn.cuh); returns 1 when the shape is not covered (caller falls back)
int adt_launch_seq_attn_bwd(int hd, const AttnArgs& a, void* stream) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_SEQ_ATTN_BWD"); on = (e && atoi(e) == 0) ? 0 : 1; }
  if (!on || !a.causal || a.H * hd != 64 || a.L > 224) return 1;
  if ((a.ldq % 4) || (a.ldk % 4) || (a.ldv % 4) || (a.ldo % 4)) return 1;
  if (a.in_bf16 && ((a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) || (a.ldo % 8))) return 1;
  const size_t smem = sab_lds_bytes(a.L, a.H);
  if (smem > 160 * 1024) return 1;
  const int mode = a.drop.thr == 0 ? 0 : (a.mask != nullptr ? 1 : 2);
  static bool done[9] = {false, false, false, false, false, false, false, false, false};
  const void* fns[9] = {(const void*)k_seq_attn_bwd<64, 0>, (const void*)k_seq_attn_bwd<64, 1>, (const void*)k_seq_attn_bwd<64, 2>,
                        (const void*)k_seq_attn_bwd<32, 0>, (const void*)k_seq_attn_bwd<32, 1>, (const void*)k_seq_attn_bwd<32, 2>,
                        (const void*)k_seq_attn_bwd<16, 0>, (const void*)k_seq_attn_bwd<16, 1>, (const void*)k_seq_attn_bwd<16, 2>};
  if (hd != 64 && hd != 32 && hd != 16) return 1;
  const int slot = (hd == 64 ? 0 : hd == 32 ? 3 : 6) + mode;
  const void* fn = fns[slot];
  if (!done[slot]) {      // the attribute is the maximum this kernel may ask for, not this launch's size
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return adt_set_error("seq_attn_bwd: hipFuncSetAttribute");
    done[slot] = true;
  }
  AttnArgs args = a;
  args.stamps = seq_stamp_buffer(true);
  void* kargs[] = {&args};
  if (hipLaunchKernel(fn, dim3(a.B), dim3(SAB_NW * 64), kargs, smem, (hipStream_t)stream) != hipSuccess) return adt_set_error("seq_attn_bwd: launch failed");
  return seq_check("seq_attn_bwd");
}

// fused backward of one attention block (adt_seqbwd_tt.cuh); dec = 0: encoder block, 1: decoder self-attention block.
// Returns 1 when the shape is not covered (the caller runs the staged kernels).
template <int HD, bool DEC>
static int seq_attn_pre_bwd_t(int mode, const SeqBwdArgs& a, hipStream_t s) {
  constexpr int H = 64 / HD;
  const size_t smem = SeqBwdLds<H>::bytes;
  static bool done[3] = {false, false, false};
  const void* fns[3] = {(const void*)k_seqtt_attn_pre_bwd<HD, 0, DEC>, (const void*)k_seqtt_attn_pre_bwd<HD, 1, DEC>,
                        (const void*)k_seqtt_attn_pre_bwd<HD, 2, DEC>};
  SeqBwdArgs args = a;
  args.stamps = seq_stamp_buffer(true);
  if (a.nsplit == 2 && !a.part) return adt_set_error("seqtt_attn_pre_bwd: two workgroups per sequence need the weight-gradient partials");
  if (a.nsplit != 2) args.nsplit = 1;
  return seq_launch(fns[mode], smem, done[mode], a.B * args.nsplit, &args, s, "seqtt_attn_pre_bwd");
}

int adt_launch_seq_attn_pre_bwd(int hd, int dec, const SeqBwdArgs& a, void* stream) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ADT_SEQ_BWD"); on = (e && atoi(e) == 0) ? 0 : 1; }
  if (!on || a.wp_img == nullptr || a.L > SB_R || (a.L & 3) || a.H * hd != 64) return 1;
  const int mode = a.drop.thr == 0 ? 0 : (a.mask != nullptr ? 1 : 2);
  hipStream_t s = (hipStream_t)stream;
  if (hd == 32) return dec ? seq_attn_pre_bwd_t<32, true>(mode, a, s) : seq_attn_pre_bwd_t<32, false>(mode, a, s);
  if (hd == 64) return dec ? seq_attn_pre_bwd_t<64, true>(mode, a, s) : seq_attn_pre_bwd_t<64, false>(mode, a, s);
  if (hd == 16) return dec ? seq_attn_pre_bwd_t<16, true>(mode, a, s) : seq_attn_pre_bwd_t<16, false>(mode, a, s);
  return 1;
}

// per-sequence backward of the token-wise chains (adt_seqpost_tt.cuh); 0 launched, 1 not covered (the caller runs the row-major chain kernels)
static bool seq_post_ok(const BwdChainArgs& a, int hd) {
  static int on = -1;
  if (on < 0) on = env_on("ADT_SEQ_POST_BWD");
  return on && a.wp_img != nullptr && a.L <= SB_R && a.T == a.B * a.L && (hd == 16 || hd == 32 || hd == 64);
}

int adt_launch_seq_post_bwd(int hd, int enc, const BwdChainArgs& a, void* stream) {
  if (!seq_post_ok(a, hd)) return 1;
  if (enc && a.drec != nullptr && a.H * hd != 64) return 1;
  static bool done[6] = {false, false, false, false, false, false};
  const void* fns[6] = {(const void*)k_seqtt_post_bwd<64, false>, (const void*)k_seqtt_post_bwd<64, true>, (const void*)k_seqtt_post_bwd<32, false>,
                        (const void*)k_seqtt_post_bwd<32, true>, (const void*)k_seqtt_post_bwd<16, false>, (const void*)k_seqtt_post_bwd<16, true>};
  const int slot = (hd == 64 ? 0 : hd == 32 ? 2 : 4) + (enc ? 1 : 0);
  BwdChainArgs args = a;
  args.stamps = enc ? seq_stamp_buffer_post() : nullptr;
  return seq_launch(fns[slot], SeqPostLds<3>::bytes, done[slot], a.B * (a.nsplit > 1 ? a.nsplit : 1), &args, (hipStream_t)stream, "seqtt_post_bwd", SP_NW);
}

int adt_launch_seq_mid_bwd(int hd, const BwdChainArgs& a, void* stream) {
  if (!seq_post_ok(a, hd)) return 1;
  static bool done = false;
  return seq_launch((const void*)k_seqtt_mid_bwd, SeqPostLds<4>::bytes, done, a.B * (a.nsplit > 1 ? a.nsplit : 1), &a, (hipStream_t)stream, "seqtt_mid_bwd", SP_MID_NW);
}

// ---- sum of the per-workgroup weight-gradient partials (adt_seqbwd_tt.cuh: sb_dw_tiles) --------------------------------------------
// G[off[slot] + n * 64 + k] += sum_wg part[wg * stride + slot * 4096 + e], e = ((4 nt + kt) * 4 + r) * 64 + lane <-> n = 16 nt + 4 (lane >> 4) + r,
// k = 16 kt + (lane & 15).  grid (slots * 4, NSPLIT): a thread owns four consecutive elements (one float4 of the partial layout = four
// consecutive k of one row n) and a contiguous range of workgroups, summed in ascending order; the NSPLIT range sums meet in G by atomics.
struct PartReduceArgs {
  float* G; const float* part; size_t stride; int nwg; int nslots;
  int slot[256]; int off[256];      // slot index inside a workgroup's partial ; float offset of the 64 x 64 block in G
  int nwg_slot[256];                // workgroups that wrote this slot (kernels with several workgroups per sequence write B * S partials)
};
constexpr int PR_SPLIT = 4;
// bf16 partials (adt_seqbwd_tt.cuh: sb_dw_tiles): element (tile, lane, r) of a slot at (tile * 64 + lane) * 4 + r.  A thread owns 16
// consecutive elements (32 bytes: lanes l .. l + 3 of one tile, r = 0 .. 3 -> four rows n, four columns k) and a contiguous range of
// workgroups, summed in ascending order in fp32; the PR_SPLIT range sums meet in G by atomics.  grid (slots, PR_SPLIT), 256 threads.
__global__ __launch_bounds__(256) void k_dwpart_reduce(PartReduceArgs a) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const int j = blockIdx.x, item = threadIdx.x;
  const int nwg = a.nwg_slot[j];
  const int per = (nwg + PR_SPLIT - 1) / PR_SPLIT, w0 = blockIdx.y * per, w1 = min(nwg, w0 + per);
  const __bf16* p = reinterpret_cast<const __bf16*>(a.part + (size_t)a.slot[j] * 4096) + item * 16;
  const size_t strideb = a.stride * 2;
  float sum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sum[i] = 0.f;
  int wg = w0;
  for (; wg + 4 <= w1; wg += 4) {
    u4v lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const u4v* src = reinterpret_cast<const u4v*>(p + (size_t)(wg + u) * strideb);
      lo[u] = src[0]; hi[u] = src[1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned wds[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
#pragma unroll
      for (int i = 0; i < 8; ++i) { sum[2 * i] += __uint_as_float(wds[i] << 16); sum[2 * i + 1] += __uint_as_float(wds[i] & 0xFFFF0000u); }
    }
  }
  for (; wg < w1; ++wg) {
    const u4v* src = reinterpret_cast<const u4v*>(p + (size_t)wg * strideb);
    const u4v lo = src[0], hi = src[1];
    const unsigned wds[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) { sum[2 * i] += __uint_as_float(wds[i] << 16); sum[2 * i + 1] += __uint_as_float(wds[i] & 0xFFFF0000u); }
  }
  const int tile = item >> 4, lane0 = (item & 15) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* dst = a.G + a.off[j] + (16 * (tile >> 2) + 4 * (lane0 >> 4) + r) * 64 + 16 * (tile & 3) + (lane0 & 15);
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(dst + k, sum[4 * k + r]);
  }
}

int adt_dwpart_reduce(float* G, const float* part, size_t stride, int nwg, const int* slots, const int* offs, int nslots, void* stream) {
  return adt_dwpart_reduce_n(G, part, stride, nwg, nullptr, slots, offs, nslots, stream);
}
int adt_dwpart_reduce_n(float* G, const float* part, size_t stride, int nwg, const int* nwg_slot, const int* slots, const int* offs, int nslots, void* stream) {
  if (nslots < 1 || nslots > 256) return adt_set_error("dwpart_reduce: %d slots", nslots);
  PartReduceArgs a;
  a.G = G; a.part = part; a.stride = stride; a.nwg = nwg; a.nslots = nslots;
  for (int i = 0; i < nslots; ++i) { a.slot[i] = slots[i]; a.off[i] = offs[i]; a.nwg_slot[i] = nwg_slot ? nwg_slot[i] : nwg; }
  hipLaunchKernelGGL(k_dwpart_reduce, dim3(nslots, PR_SPLIT), dim3(256), 0, (hipStream_t)stream, a);
  return seq_check("dwpart_reduce");
}

// every block-weight gradient of the backward can go through private partials: all five per-sequence backward kernels cover the shape
int adt_seq_partials(int prec, int L, int d, int hd) {
  static int on = -1;
  if (on < 0) on = env_on("ADT_SEQ_PARTIALS") && env_on("ADT_SEQ_POST_BWD");
  return on && adt_seq_lean(prec, L, d, hd);
}
