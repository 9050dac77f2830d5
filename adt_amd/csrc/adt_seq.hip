// Launchers of the per-sequence fused layer kernels (adt_seqfwd.cuh, adt_seqbwd.cuh): one workgroup per user sequence.
#include "adt_host.h"
#include <stdlib.h>
#include "adt_seqfwd.cuh"

using namespace adt;

static int seq_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return adt_set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

static int seq_launch(const void* fn, size_t smem, bool& attr_done, int grid, const void* args_ptr, hipStream_t s, const char* what) {
  if (!attr_done) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return adt_set_error("%s: hipFuncSetAttribute(%zu)", what, smem);
    attr_done = true;
  }
  void* kargs[] = {const_cast<void*>(args_ptr)};
  if (hipLaunchKernel(fn, dim3(grid), dim3(SQ_NW * 64), kargs, smem, s) != hipSuccess) return adt_set_error("%s: launch failed", what);
  return seq_check(what);
}

// ---- pre-packed weight images (adt_wave.cuh: WPack) -----------------------------------------------------------------------------
struct PackArgs {
  const float* base;      // start of the packed parameter range
  __bf16* img;
  int n;
  int off[256];           // float offsets (relative to base) of the 64 x 64 blocks
};

__global__ __launch_bounds__(256) void k_pack_wimg(PackArgs a) {
  const int off = a.off[blockIdx.x];
  const float* W = a.base + off;
  __bf16* plain = a.img + 6 * (size_t)off;
  __bf16* trans = plain + WPACK_IMG;
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {
    const int n = i >> 4, k4 = (i & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(W + n * 64 + k4);
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      plain[n * 72 + k4 + j] = (__bf16)x[j];
      trans[(k4 + j) * 72 + n] = (__bf16)x[j];
    }
  }
}

int adt_pack_wimg(const float* base, void* img, const int* offs, int n, void* stream) {
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

static unsigned long long* g_stamps = nullptr;

extern "C" int adt_seq_stamps_read(unsigned long long* out, int n) {      // debugging aid of tools/seq_stamps.py, not part of include/adt_hip.h
  if (!g_stamps) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpy(out, g_stamps, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

static void seq_ablate(SeqFwdArgs& a) {
  static int st = -1;
  if (st < 0) {
    const char* e = getenv("ADT_SEQ_STAMPS");
    st = (e && atoi(e)) ? 1 : 0;
    if (st && hipMalloc(&g_stamps, 8 * 16 * sizeof(unsigned long long)) != hipSuccess) st = 0;
  }
  a.stamps = st ? g_stamps : nullptr;
  static int ab = -1;
  if (ab < 0) { const char* e = getenv("ADT_SEQ_ABLATE"); ab = e ? atoi(e) : 0; }
  a.ablate = ab;
  if (ab & 1) { a.qkv = nullptr; a.o = nullptr; a.h = nullptr; a.u = nullptr; a.a1 = nullptr; a.q2 = nullptr; a.kv2 = nullptr; a.o2 = nullptr; a.mask = nullptr; a.mask2 = nullptr; }
}

int adt_launch_seq_enc_fwd(int hd, const SeqFwdArgs& a, void* stream) {
  static bool done[3] = {false, false, false};
  SeqFwdArgs args = a;
  seq_ablate(args);
  const size_t smem = SeqFwdLds<6>::bytes;
  hipStream_t s = (hipStream_t)stream;
  if (hd == 64) return seq_launch((const void*)k_seq_enc_fwd<64, 2>, smem, done[0], a.B, &args, s, "seq_enc_fwd<64>");
  if (hd == 32) return seq_launch((const void*)k_seq_enc_fwd<32, 2>, smem, done[1], a.B, &args, s, "seq_enc_fwd<32>");
  if (hd == 16) return seq_launch((const void*)k_seq_enc_fwd<16, 4>, smem, done[2], a.B, &args, s, "seq_enc_fwd<16>");
  return adt_set_error("seq_enc_fwd: head size %d", hd);
}

int adt_launch_seq_dec_fwd(int hd, const SeqFwdArgs& a, void* stream) {
  static bool done[3] = {false, false, false};
  SeqFwdArgs args = a;
  seq_ablate(args);
  const size_t smem = SeqFwdLds<5>::bytes;
  hipStream_t s = (hipStream_t)stream;
  if (hd == 64) return seq_launch((const void*)k_seq_dec_fwd<64>, smem, done[0], a.B, &args, s, "seq_dec_fwd<64>");
  if (hd == 32) return seq_launch((const void*)k_seq_dec_fwd<32>, smem, done[1], a.B, &args, s, "seq_dec_fwd<32>");
  if (hd == 16) return seq_launch((const void*)k_seq_dec_fwd<16>, smem, done[2], a.B, &args, s, "seq_dec_fwd<16>");
  return adt_set_error("seq_dec_fwd: head size %d", hd);
}
