#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { double a[12]; long b[6]; int c[10]; };
__global__ void knop(Big p, double* out) { if (p.c[0] == 12345) out[0] = p.a[0]; }
template <typename F> double host_us(F f, int n) {
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  double* out; (void)hipMalloc(&out, 64);
  Big p{}; hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreate(&s2);
  for (int rep = 0; rep < 2; ++rep) {
    double a = host_us([&] { hipLaunchKernelGGL(knop, dim3(1), dim3(64), 0, 0, p, out); }, 2000); (void)hipDeviceSynchronize();
    double b = host_us([&] { hipLaunchKernelGGL(knop, dim3(1), dim3(64), 0, s1, p, out); }, 2000); (void)hipDeviceSynchronize();
    double c = host_us([&] { hipLaunchKernelGGL(knop, dim3(1), dim3(64), 0, s2, p, out); }, 2000); (void)hipDeviceSynchronize();
    double d = host_us([&] { hipLaunchKernelGGL(knop, dim3(1), dim3(64), 0, s1, p, out); (void)hipGetLastError(); }, 2000); (void)hipDeviceSynchronize();
    printf("host us/launch: null %.2f  nonblocking %.2f  blocking-stream %.2f  nonblocking+getlasterror %.2f\n", a, b, c, d);
  }
  // end-to-end time of 2000 dependent tiny kernels (GPU-side throughput)
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(knop, dim3(1), dim3(64), 0, s1, p, out);
  (void)hipStreamSynchronize(s1);
  auto t1 = std::chrono::steady_clock::now();
  printf("2000 launches + sync: %.2f us each\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
  // graph replay
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(knop, dim3(1), dim3(64), 0, s1, p, out);
  (void)hipStreamEndCapture(s1, &g);
  auto t2 = std::chrono::steady_clock::now();
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  auto t3 = std::chrono::steady_clock::now();
  printf("instantiate 400-node graph: %.1f us\n", std::chrono::duration<double, std::micro>(t3 - t2).count());
  for (int rep = 0; rep < 3; ++rep) {
    auto t4 = std::chrono::steady_clock::now();
    (void)hipGraphLaunch(ge, s1);
    auto t5 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(s1);
    auto t6 = std::chrono::steady_clock::now();
    printf("graph launch host %.1f us, total %.1f us (%.2f us/node)\n", std::chrono::duration<double, std::micro>(t5 - t4).count(),
           std::chrono::duration<double, std::micro>(t6 - t4).count(), std::chrono::duration<double, std::micro>(t6 - t4).count() / 400);
  }
  return 0;
}
