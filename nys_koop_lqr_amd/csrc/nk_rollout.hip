// Lifted-state recursion z_{t+1} = A z_t + B u_t (validate_dyn_sys, benchmark_lqr_cloth.py:29-32) and the lifted closed loop
// (lqr_control, benchmark_lqr_cloth.py:79-84) for a FEW trajectories: a matrix-vector chain, latency bound, where the
// 128x128 MFMA tile of the GEMM engine would idle 127/128 of its rows.  One wave per output row: the 64 lanes stride
// over the row of G = [A | B] (coalesced 512-B segments), up to 8 trajectories share each load of G, wavefront
// reduction, lane 0 stores.  G (32 MB at m = 2000) is re-read every step and stays in L2 / Infinity Cache.
// Larger batches use the GEMM engine (nk_api.hip).
#include "nk_common.h"

namespace nk {

constexpr int STEP_MAX_BATCH = 8;

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// out[b][row] = sum_k G[row][k] z[b][k]  (k < mz)  +  sum_k G[row][mz + k] u[b][k]  (k < pu)  + bias[row]
__global__ void __launch_bounds__(256) lifted_step_kernel(const double* __restrict__ G, int64_t ldg, int m, int mz, int pu,
                                                          const double* __restrict__ z, int64_t zstride,
                                                          const double* __restrict__ u, int64_t ustride,
                                                          const double* __restrict__ bias, double* __restrict__ out,
                                                          int64_t ostride, int batch) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const double* g = G + (int64_t)row * ldg;
  double acc[STEP_MAX_BATCH];
#pragma unroll
  for (int b = 0; b < STEP_MAX_BATCH; ++b) acc[b] = 0.0;
  for (int k = lane; k < mz; k += 64) {
    const double gv = g[k];
#pragma unroll
    for (int b = 0; b < STEP_MAX_BATCH; ++b)
      if (b < batch) acc[b] = fma(gv, z[(int64_t)b * zstride + k], acc[b]);
  }
  for (int k = lane; k < pu; k += 64) {
    const double gv = g[mz + k];
#pragma unroll
    for (int b = 0; b < STEP_MAX_BATCH; ++b)
      if (b < batch) acc[b] = fma(gv, u[(int64_t)b * ustride + k], acc[b]);
  }
#pragma unroll
  for (int b = 0; b < STEP_MAX_BATCH; ++b) {
    if (b < batch) {
      const double s = wave_sum64(acc[b]);
      if (lane == 0) out[(int64_t)b * ostride + row] = s + (bias ? bias[row] : 0.0);
    }
  }
}

int launch_lifted_step(nk_ctx* ctx, const double* G, int64_t ldg, int m, int mz, int pu, const double* z, int64_t zstride,
                       const double* u, int64_t ustride, const double* bias, double* out, int64_t ostride, int batch) {
  NK_REQUIRE(batch >= 1 && batch <= STEP_MAX_BATCH, "lifted_step: batch 1..8");
  hipLaunchKernelGGL(lifted_step_kernel, dim3((m + 3) / 4), dim3(256), 0, ctx->stream, G, ldg, m, mz, pu, z, zstride, u,
                     ustride, bias, out, ostride, batch);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
