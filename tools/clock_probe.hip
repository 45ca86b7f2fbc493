// Shader clock seen by a latency-bound single-workgroup kernel vs a chip-filling one (DVFS check):
// clock64() counts shader cycles, wall_clock64() a constant 100 MHz reference.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out, long long* t, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  double a = threadIdx.x * 1e-9 + 1.0;
  for (int i = 0; i < iters; ++i) a = fma(a, 1.0000001, 1e-9);
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  double* out; long long* t; long long h[2];
  hipMalloc(&out, 8 * 1024 * 4096); hipMalloc(&t, 16);
  for (int blocks : {1, 1, 64, 2048, 1, 1}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 0, 0, out, t, 200000); hipEventRecord(e1);
      hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      printf("blocks %5d: %.3f ms, clock64 %lld, wall(100MHz) %lld -> shader clock %.0f MHz, %.1f cycles per dependent fma\n", blocks, ms,
             h[0], h[1], h[0] / (h[1] / 100.0), (double)h[0] / 200000);
    }
  }
  return 0;
}
