// Micro-benchmark: issue cost of single VALU instructions on gfx950 (one wave per SIMD and two waves per SIMD), cycles per instruction
// from s_memtime around a 4096-instruction unrolled stream of independent chains.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/mb_valu tools/mb_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
#define BODY(OP) \
  for (int it = 0; it < 64; ++it) { \
    REP8(asm volatile(OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8" \
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k));) }

template <int WHICH>
__global__ void k(uint32_t* out, unsigned long long* cyc) {
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7, kk = 0x7feb352d;
  uint32_t k = kk ^ blockIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (WHICH == 0) { BODY("v_mul_lo_u32") }
  if (WHICH == 1) { BODY("v_mul_u32_u24") }
  if (WHICH == 2) { BODY("v_xor_b32") }
  if (WHICH == 3) { BODY("v_add_u32") }
  if (WHICH == 4) { BODY("v_mul_hi_u32") }
  if (WHICH == 5) { BODY("v_mul_f32") }
  if (WHICH == 6) { BODY("v_lshrrev_b32") }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W>
void run(const char* name, int threads) {
  uint32_t* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<W>, dim3(256), dim3(threads), 0, 0, out, cyc);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  printf("%-16s %4d threads/WG: %.2f cycles per instruction per wave (memtime ticks / 4096)\n", name, threads, s / 256 / 4096.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int th : {256, 512}) {
    if (th == 256) { run<0>("v_mul_lo_u32", th); run<1>("v_mul_u32_u24", th); run<2>("v_xor_b32", th); run<3>("v_add_u32", th); run<4>("v_mul_hi_u32", th); run<5>("v_mul_f32", th); run<6>("v_lshrrev_b32", th); }
    else { run<0>("v_mul_lo_u32", th); run<1>("v_mul_u32_u24", th); run<2>("v_xor_b32", th); run<3>("v_add_u32", th); run<4>("v_mul_hi_u32", th); run<5>("v_mul_f32", th); run<6>("v_lshrrev_b32", th); }
  }
  return 0;
}
