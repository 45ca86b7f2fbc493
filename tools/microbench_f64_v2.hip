// v2: inline-asm MFMA f64 loops (compiler-independent), with in-kernel clock measurement.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MF(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));

template<int NACC, int ZERO>
__global__ void __launch_bounds__(256) mfma_asm(double* out, unsigned long long* clk, int iters, double a0, double b0) {
  d4 c0={0,0,0,0}, c1=c0, c2=c0, c3=c0, c4=c0, c5=c0, c6=c0, c7=c0;
  double a = ZERO ? 0.0 : a0 + threadIdx.x*1.37e-3, b = ZERO ? 0.0 : b0 - threadIdx.x*0.77e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it=0; it<iters; ++it) {
    if (NACC==1) { MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) }
    if (NACC==2) { MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) }
    if (NACC==4) { MF(c0) MF(c1) MF(c2) MF(c3) MF(c0) MF(c1) MF(c2) MF(c3) }
    if (NACC==8) { MF(c0) MF(c1) MF(c2) MF(c3) MF(c4) MF(c5) MF(c6) MF(c7) }
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  d4 s = c0+c1+c2+c3+c4+c5+c6+c7;
  out[blockIdx.x*blockDim.x+threadIdx.x] = s[0]+s[1]+s[2]+s[3];
  if (threadIdx.x==0) { clk[blockIdx.x*2] = t1-t0; clk[blockIdx.x*2+1] = r1-r0; }
}

template<typename F> float timeit(F f, int reps=5) {
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  float best=1e30f;
  for (int r=0;r<reps;r++){ (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms; }
  return best;
}

template<int NACC, int ZERO> void run(int ncu, int bpc, double* out, unsigned long long* clk, int iters) {
  float ms = timeit([&]{ mfma_asm<NACC,ZERO><<<ncu*bpc,256>>>(out, clk, iters, 1.0, 2.0); });
  unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double nm = (double)iters*8;  // per wave
  double fl = (double)ncu*bpc*4*nm*2048.0;
  double ghz = (double)h[0]/((double)h[1]*10.0);   // memrealtime ticks at 100 MHz
  printf("mfma_f64 asm NACC=%d zero=%d bpc=%d: %.3f ms %.2f TF | %.1f cyc/MFMA/wave, clock %.2f GHz\n", NACC, ZERO, bpc, ms, fl/ms*1e-9, (double)h[0]/nm, ghz);
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p,0);
  int ncu = p.multiProcessorCount;
  double* out; (void)hipMalloc(&out, sizeof(double)*ncu*8*512);
  unsigned long long* clk; (void)hipMalloc(&clk, 16*ncu*8);
  int iters = 40000;
  for (int bpc=1; bpc<=2; ++bpc) {
    run<1,0>(ncu,bpc,out,clk,iters); run<2,0>(ncu,bpc,out,clk,iters); run<4,0>(ncu,bpc,out,clk,iters); run<8,0>(ncu,bpc,out,clk,iters);
    run<4,1>(ncu,bpc,out,clk,iters);
  }
  // one block only (single CU): is the per-instruction cycle count the same when the chip is idle?
  { float ms = timeit([&]{ mfma_asm<4,0><<<1,256>>>(out, clk, iters, 1.0, 2.0); });
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("single block: %.3f ms, %.1f cyc/MFMA/wave clock %.2f GHz\n", ms, (double)h[0]/(iters*8.0), (double)h[0]/((double)h[1]*10.0)); }
  { float ms = timeit([&]{ mfma_asm<4,0><<<1,64>>>(out, clk, iters, 1.0, 2.0); });
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("single wave: %.3f ms, %.1f cyc/MFMA/wave clock %.2f GHz\n", ms, (double)h[0]/(iters*8.0), (double)h[0]/((double)h[1]*10.0)); }
  return 0;
}
