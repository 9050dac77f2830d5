// C ABI (include/adt_hip.h, "fused all-item logits + cross-entropy"): workspace layout and launch sequence of adt_lce.cuh.
#include <algorithm>
#include <cstdlib>
#include "adt_host.h"
#include "adt_lce.cuh"

using namespace adt;

static int lce_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return adt_set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

static int g_lce_slots = 0;
static int lce_slots() {
  if (!g_lce_slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_lce_slots = cus;
  }
  return g_lce_slots;
}

static size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct LceLayout {
  size_t Eb, biasp, Hb, nlse, part_m, part_s, dpart, total;
  int Vpad, Mpad;
};
static LceLayout lce_layout(int mcap, int V, int K, int slots) {
  LceLayout L;
  L.Vpad = (int)up((size_t)V, LCE_XR_BWD);
  L.Mpad = (int)up((size_t)(mcap > 0 ? mcap : 1), LCE_XR_FWD);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += up(bytes, 256); return at; };
  L.Eb = take((size_t)L.Vpad * K * 2);
  L.biasp = take((size_t)L.Vpad * 4);
  L.Hb = take((size_t)L.Mpad * K * 2);
  L.nlse = take((size_t)L.Mpad * 4);
  const size_t prow_f = std::max((size_t)slots * LCE_XR_FWD, (size_t)L.Mpad);
  L.part_m = take(prow_f * 4);
  L.part_s = take(prow_f * 4);
  const size_t prow_d = std::max((size_t)slots * LCE_XR_BWD, (size_t)std::max(L.Mpad, L.Vpad));
  L.dpart = take(prow_d * K * 4);
  L.total = o;
  return L;
}

template <int KD, int MODE>
static int lce_launch(const LceArgs& a, hipStream_t s, const char* what) {
  static bool done = false;
  const void* fn = (const void*)k_lce<KD, MODE>;
  if (!done) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LceGeo<KD>::LDS) != hipSuccess) return adt_set_error("%s: hipFuncSetAttribute", what);
    done = true;
  }
  hipLaunchKernelGGL((k_lce<KD, MODE>), dim3(a.slots), dim3(LCE_NTH), LceGeo<KD>::LDS, s, a);
  return lce_check(what);
}

template <int KD>
static int lce_reduce(const LceReduceArgs& a, int nx_cap, hipStream_t s, const char* what) {
  static bool done = false;
  const size_t smem = 32 * (KD + 4) * sizeof(float);
  const void* fn = (const void*)k_lce_reduce<KD>;
  if (!done) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return adt_set_error("%s: hipFuncSetAttribute", what);
    done = true;
  }
  const int nb = (nx_cap + 31) / 32;
  hipLaunchKernelGGL((k_lce_reduce<KD>), dim3(nb < 2048 ? nb : 2048), dim3(256), smem, s, a);
  return lce_check(what);
}

template <int KD>
static int lce_run(const float* h, int ldh, const int32_t* rows, const int32_t* labels, int mcap, const int32_t* m_dev, const float* E, int lde,
                   const float* bias, int V, const float* inv_count, float* loss64, float* lse_out, float* dh, int lddh, float* dE, int lddE,
                   float* dbias, unsigned char* ws, hipStream_t s) {
  const int slots = lce_slots();
  const LceLayout L = lce_layout(mcap, V, KD, slots);
  __bf16* Eb = reinterpret_cast<__bf16*>(ws + L.Eb);
  __bf16* Hb = reinterpret_cast<__bf16*>(ws + L.Hb);
  float* biasp = reinterpret_cast<float*>(ws + L.biasp);
  float* nlse = reinterpret_cast<float*>(ws + L.nlse);
  float* part_m = reinterpret_cast<float*>(ws + L.part_m);
  float* part_s = reinterpret_cast<float*>(ws + L.part_s);
  float* dpart = reinterpret_cast<float*>(ws + L.dpart);
  {
    LcePackArgs p{E, lde, nullptr, nullptr, V, LCE_XR_BWD, KD, Eb, bias, biasp};
    hipLaunchKernelGGL(k_lce_pack, dim3(2048), dim3(256), 0, s, p);
    LcePackArgs q{h, ldh, rows, m_dev, mcap, LCE_XR_FWD, KD, Hb, nullptr, nullptr};
    hipLaunchKernelGGL(k_lce_pack, dim3(1024), dim3(256), 0, s, q);
    if (int rc = lce_check("lce: pack")) return rc;
  }
  {
    LceArgs a{Hb, Eb, nullptr, biasp, m_dev, nullptr, mcap, V, part_m, part_s, nullptr, nullptr, slots};
    if (int rc = lce_launch<KD, LCE_FWD>(a, s, "lce: forward")) return rc;
    LceCombineArgs c{part_m, part_s, Hb, Eb, bias, labels, m_dev, mcap, V, KD, slots, inv_count, loss64, nlse, lse_out, dE, lddE, dbias};
    hipLaunchKernelGGL(k_lce_combine<KD>, dim3(std::min(4096, (L.Mpad + 15) / 16)), dim3(256), 0, s, c);
    if (int rc = lce_check("lce: combine")) return rc;
  }
  if (!dh) return 0;                                       // loss only
  {
    LceArgs a{Hb, Eb, nlse, biasp, m_dev, nullptr, mcap, V, nullptr, nullptr, dpart, nullptr, slots};
    if (int rc = lce_launch<KD, LCE_DH>(a, s, "lce: dh")) return rc;
    LceReduceArgs r{dpart, m_dev, mcap, nullptr, V, slots, KD, rows, labels, Eb, inv_count, dh, lddh};
    if (int rc = lce_reduce<KD>(r, mcap, s, "lce: dh reduce")) return rc;
  }
  {
    LceArgs a{Eb, Hb, biasp, nlse, nullptr, m_dev, V, mcap, nullptr, nullptr, dpart, dbias, slots};
    if (int rc = lce_launch<KD, LCE_DE>(a, s, "lce: dE")) return rc;
    LceReduceArgs r{dpart, nullptr, V, m_dev, mcap, slots, KD, nullptr, nullptr, nullptr, nullptr, dE, lddE};
    if (int rc = lce_reduce<KD>(r, V, s, "lce: dE reduce")) return rc;
  }
  return 0;
}

extern "C" {

int adt_lce_slots(int slots) {
  if (slots > 0) g_lce_slots = slots;
  return lce_slots();
}

int adt_lce_supported(int prec, int K) { return prec == PREC_BF16 && (K == 128 || K == 256); }

int64_t adt_lce_workspace_bytes(int mcap, int V, int K) { return (int64_t)lce_layout(mcap, V, K, lce_slots()).total; }

int adt_lce_fwd_bwd(const float* h, int ldh, const int32_t* rows, const int32_t* labels, int mcap, const int32_t* m_dev, const float* E, int lde,
                    const float* bias, int V, int K, const float* inv_count, float* loss64, float* lse_out, float* dh, int lddh, float* dE, int lddE,
                    float* dbias, void* workspace, int64_t workspace_bytes, void* stream) {
  if (mcap <= 0) return 0;
  if (!(K == 128 || K == 256)) return adt_set_error("lce: K=%d (128 or 256)", K);
  if ((ldh % 4) || (lde % 4) || (dh && (lddh % 4)) || (lddE % 4)) return adt_set_error("lce: leading dimensions %% 4");
  if (((uintptr_t)h | (uintptr_t)E | (uintptr_t)dh | (uintptr_t)dE | (uintptr_t)workspace) & 15u) return adt_set_error("lce: 16-byte alignment");
  if (!dE || !dbias || !bias || !inv_count || !loss64) return adt_set_error("lce: NULL argument");
  if (workspace_bytes < adt_lce_workspace_bytes(mcap, V, K)) return adt_set_error("lce: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)adt_lce_workspace_bytes(mcap, V, K));
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  return K == 256 ? lce_run<256>(h, ldh, rows, labels, mcap, m_dev, E, lde, bias, V, inv_count, loss64, lse_out, dh, lddh, dE, lddE, dbias, ws, s)
                  : lce_run<128>(h, ldh, rows, labels, mcap, m_dev, E, lde, bias, V, inv_count, loss64, lse_out, dh, lddh, dE, lddE, dbias, ws, s);
}

}  // extern "C"
