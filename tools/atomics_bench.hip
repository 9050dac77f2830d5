// Same-address float-atomic chains on MI355X (cited in DESIGN.md, "Same-address float atomics and replicas").
//   hipcc --offload-arch=gfx950 -O3 -o atomics_bench tools/atomics_bench.hip && ./atomics_bench
// 256 workgroups x 256 threads each add a 64 x 64 fp32 block (16 atomics per thread) into replica rep(blockIdx) of the destination.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_flush(float* dst, int mode, int nrep, size_t stride) {
  int rep = 0;
  if (mode == 1) rep = blockIdx.x & (nrep - 1);            // by XCD (workgroups are dealt round-robin to the 8 XCDs)
  if (mode == 2) rep = blockIdx.x;                          // private copy
  if (mode == 3) rep = (blockIdx.x >> 3) & (nrep - 1);      // by position inside the XCD
  float* p = dst + rep * stride;
  for (int i = threadIdx.x; i < 4096; i += 256) atomicAdd(p + i, 1.0f);
}

int main() {
  const int G = 256;
  float* d;
  CHECK(hipMalloc(&d, (size_t)G * 4096 * 4));
  CHECK(hipMemset(d, 0, (size_t)G * 4096 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct { int mode, nrep; const char* name; } cfg[] = {{0, 1, "one copy"}, {1, 8, "8 replicas by bid&7"}, {3, 8, "8 replicas by (bid>>3)&7"},
                                                        {1, 2, "2 by bid&1"}, {1, 32, "32 by bid&31"}, {2, 256, "private"}};
  for (auto& c : cfg) {
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_flush, dim3(G), dim3(256), 0, 0, d, c.mode, c.nrep, (size_t)4096);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_flush, dim3(G), dim3(256), 0, 0, d, c.mode, c.nrep, (size_t)4096);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %7.2f us per launch (256 WGs x 4096 float atomics)\n", c.name, ms / 20 * 1e3);
  }
  return 0;
}
