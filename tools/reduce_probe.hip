// Check of the cross-lane reduction used by the single-workgroup recursion (nk_rollout.hip): 4 values per lane summed
// over the 32 lanes of a half wave with v_permlane16_swap + DPP (no LDS traffic).
// hipcc -O3 --offload-arch=gfx950 tools/reduce_probe.hip -o /tmp/reduce_probe && /tmp/reduce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void swap16_f64(double& a, double& b) {
  const u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x);
  b = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double reduce_4rows_32parts(double a0, double a1, double a2, double a3, int lane) {
  swap16_f64(a0, a2);
  const double e0 = a0 + a2;
  swap16_f64(a1, a3);
  const double e1 = a1 + a3;
  const bool hi8 = (lane & 8) != 0;
  const double keep = hi8 ? e1 : e0, send = hi8 ? e0 : e1;
  double c = keep + dpp_f64<0x140>(send);  // row_mirror
  c += dpp_f64<0xB1>(c);                   // quad_perm [1,0,3,2]
  c += dpp_f64<0x4E>(c);                   // quad_perm [2,3,0,1]
  c += dpp_f64<0x141>(c);                  // row_half_mirror
  return c;
}
__device__ __forceinline__ void swap32_f64(double& a, double& b) {
  const u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x);
  b = __hiloint2double((int)hi.y, (int)lo.y);
}
// sum of one value over the 64 lanes, in every lane
__device__ __forceinline__ double wave_sum64_dpp(double c) {
  c += dpp_f64<0xB1>(c);
  c += dpp_f64<0x4E>(c);
  c += dpp_f64<0x141>(c);
  c += dpp_f64<0x140>(c);
  double a = c, b = c;
  swap16_f64(a, b);
  c = a + b;
  a = c; b = c;
  swap32_f64(a, b);
  return a + b;
}
__global__ void k2(const double* in, double* out) { out[threadIdx.x] = wave_sum64_dpp(in[threadIdx.x * 4]); }
__global__ void k(const double* in, double* out) {  // in[lane][4], out[lane]
  const int lane = threadIdx.x;
  out[lane] = reduce_4rows_32parts(in[lane * 4], in[lane * 4 + 1], in[lane * 4 + 2], in[lane * 4 + 3], lane);
}
int main() {
  double h[256], o[64], *d, *dout;
  for (int i = 0; i < 256; ++i) h[i] = (double)((i * 37) % 101) + 0.25 * (i % 4);
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int half = lane >> 5, r = 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1);
    double ref = 0.0;
    for (int p = 0; p < 32; ++p) ref += h[(half * 32 + p) * 4 + r];
    if (std::fabs(ref - o[lane]) > 1e-9) { ++bad; printf("lane %d row %d: got %g want %g\n", lane, r, o[lane], ref); }
  }
  printf(bad ? "reduce_probe: %d lanes WRONG\n" : "reduce_probe: all 64 lanes correct\n", bad);
  k2<<<1, 64>>>(d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  double tot = 0.0;
  for (int lane = 0; lane < 64; ++lane) tot += h[lane * 4];
  int bad2 = 0;
  for (int lane = 0; lane < 64; ++lane)
    if (std::fabs(o[lane] - tot) > 1e-9) { ++bad2; printf("wave sum lane %d: got %g want %g\n", lane, o[lane], tot); }
  printf(bad2 ? "wave_sum64_dpp: %d lanes WRONG\n" : "wave_sum64_dpp: all 64 lanes correct\n", bad2);
  return bad != 0 || bad2 != 0;
}
