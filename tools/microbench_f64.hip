// Micro-benchmarks for gfx950 fp64 rates (MFMA f64 16x16x4, VALU v_fma_f64, mixed) and HBM copy.
// Build: hipcc --offload-arch=gfx950 -O3 microbench_f64.hip -o microbench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template<int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i=0;i<NACC;i++) acc[i] = d4{0,0,0,0};
  double a = a0 + threadIdx.x*1e-9, b = b0 - threadIdx.x*1e-9;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0,0,0);
  }
  double s=0; for (int i=0;i<NACC;i++) s += acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}

template<int NACC>
__global__ void __launch_bounds__(256) fma_loop(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i=0;i<NACC;i++) acc[i] = i;
  double a = a0 + threadIdx.x*1e-9, b = b0 - threadIdx.x*1e-9;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s=0; for (int i=0;i<NACC;i++) s += acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}

// sub + fma (the direct-difference distance inner op)
template<int NACC>
__global__ void __launch_bounds__(256) subfma_loop(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i=0;i<NACC;i++) acc[i] = i;
  double a = a0 + threadIdx.x*1e-9, b = b0 - threadIdx.x*1e-9;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) { double d = a - (b + i); acc[i] = __builtin_fma(d, d, acc[i]); }
    a += 1e-9;
  }
  double s=0; for (int i=0;i<NACC;i++) s += acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}

// half of the waves MFMA, half VALU fma (co-issue test)
__global__ void __launch_bounds__(512) mixed_loop(double* out, int iters, double a0, double b0) {
  int wave = threadIdx.x >> 6;
  double a = a0 + threadIdx.x*1e-9, b = b0 - threadIdx.x*1e-9;
  double s = 0;
  if (wave < 4) {
    d4 acc[4]; for (int i=0;i<4;i++) acc[i] = d4{0,0,0,0};
    for (int it=0; it<iters; ++it) {
#pragma unroll
      for (int i=0;i<4;i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0,0,0);
    }
    for (int i=0;i<4;i++) s += acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  } else {
    double acc[16]; for (int i=0;i<16;i++) acc[i] = i;
    for (int it=0; it<iters; ++it) {
#pragma unroll
      for (int i=0;i<16;i++) acc[i] = __builtin_fma(a, acc[i], b);
    }
    for (int i=0;i<16;i++) s += acc[i];
  }
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}

__global__ void copy_k(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x*blockDim.x;
  for (; i<n; i+=stride) out[i] = in[i];
}

template<typename F> float timeit(F f, int reps=5) {
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best=1e30f;
  for (int r=0;r<reps;r++){ hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms; }
  return best;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int ncu = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, sizeof(double)*ncu*8*512));
  int iters = 20000;
  // MFMA: blocks per CU 1 (4 waves = 1 per SIMD) and 2
  for (int bpc=1; bpc<=2; ++bpc) {
    { float ms = timeit([&]{ mfma_loop<1><<<ncu*bpc,256>>>(out, iters, 1.0, 2.0); });
      double fl = (double)ncu*bpc*4*iters*1*2048.0; printf("mfma_f64 NACC=1 bpc=%d: %.3f ms  %.2f TF\n", bpc, ms, fl/ms*1e-9); }
    { float ms = timeit([&]{ mfma_loop<4><<<ncu*bpc,256>>>(out, iters, 1.0, 2.0); });
      double fl = (double)ncu*bpc*4*iters*4*2048.0; printf("mfma_f64 NACC=4 bpc=%d: %.3f ms  %.2f TF\n", bpc, ms, fl/ms*1e-9); }
    { float ms = timeit([&]{ mfma_loop<16><<<ncu*bpc,256>>>(out, iters/4, 1.0, 2.0); });
      double fl = (double)ncu*bpc*4*(iters/4)*16*2048.0; printf("mfma_f64 NACC=16 bpc=%d: %.3f ms  %.2f TF\n", bpc, ms, fl/ms*1e-9); }
  }
  for (int bpc=1; bpc<=4; bpc*=2) {
    { float ms = timeit([&]{ fma_loop<16><<<ncu*bpc,256>>>(out, iters, 1.0000001, 1e-9); });
      double fl = (double)ncu*bpc*256*(double)iters*16*2.0; printf("v_fma_f64 NACC=16 bpc=%d: %.3f ms  %.2f TF\n", bpc, ms, fl/ms*1e-9); }
    { float ms = timeit([&]{ subfma_loop<16><<<ncu*bpc,256>>>(out, iters, 1.0000001, 1e-9); });
      double fl = (double)ncu*bpc*256*(double)iters*16; printf("sub+fma f64 NACC=16 bpc=%d: %.3f ms  %.2f Gpair-k/s (x3 flop = %.2f TF)\n", bpc, ms, fl/ms*1e-6, 3*fl/ms*1e-9); }
  }
  { float ms = timeit([&]{ mixed_loop<<<ncu,512>>>(out, iters, 1.0000001, 1e-9); });
    double fm = (double)ncu*4*iters*4*2048.0, fv = (double)ncu*256*(double)iters*16*2.0;
    printf("mixed (4 mfma waves + 4 valu waves / CU): %.3f ms  mfma %.2f TF + valu %.2f TF = %.2f TF\n", ms, fm/ms*1e-9, fv/ms*1e-9, (fm+fv)/ms*1e-9); }
  // HBM copy 2 GiB
  size_t nb = (size_t)2<<30; double2 *a,*b; CK(hipMalloc(&a,nb)); CK(hipMalloc(&b,nb)); CK(hipMemset(a,1,nb));
  { float ms = timeit([&]{ copy_k<<<ncu*8,256>>>(a,b,nb/16); }); printf("copy 2GiB: %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0*nb/ms*1e-9); }
  // pinned H2D bandwidth
  void* h; CK(hipHostMalloc(&h, (size_t)1<<30));
  { float ms = timeit([&]{ hipMemcpyAsync(a,h,(size_t)1<<30,hipMemcpyHostToDevice,0); }, 3); printf("H2D pinned 1GiB: %.3f ms  %.2f GB/s\n", ms, ((size_t)1<<30)/ms*1e-6); }
  return 0;
}
