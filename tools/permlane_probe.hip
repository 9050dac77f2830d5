#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xor32(float x) {
  const unsigned u = __float_as_uint(x);
  const u2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor16(float x) {
  const unsigned u = __float_as_uint(x);
  const u2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__global__ void k(float* out) {
  float v = (float)threadIdx.x;
  out[threadIdx.x] = xor32(v);
  out[64 + threadIdx.x] = xor16(v);
  out[128 + threadIdx.x] = v + xor16(v) + xor32(v + xor16(v));
}
#include <stdio.h>
int main() {
  float* d; hipMalloc(&d, 192 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; ++i) { if (h[i] != (float)(i ^ 32) || h[64 + i] != (float)(i ^ 16)) ok = 0; }
  printf("xor32/xor16 %s; lane0 gets %g %g; colsum lane 5: %g (want %d)\n", ok ? "OK" : "WRONG", h[0], h[64], h[128 + 5], 5 + 21 + 37 + 53);
  return 0;
}
