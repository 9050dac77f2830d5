// Probe of ds_read_b64_tr_b16 (gfx950): which element each lane receives.  Image [rows][72] of 16-bit values row*128+col.
// Lane 4q+p of each 16-lane group g supplies the address of row (R0 + q), columns C0 + 4p .. 4p+3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short img[64 * 72];
  for (int i = threadIdx.x; i < 64 * 72; i += 64) img[i] = (short)((i / 72) * 128 + (i % 72));
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
  const short* addr = img + (8 * grp + q) * 72 + 16 + 4 * p;       // block: rows 8g..8g+3, columns 16..31
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] / 128, h[l*4+j] % 128); printf("\n"); }
  return 0;
}
