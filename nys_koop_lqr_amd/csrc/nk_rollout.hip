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

// ---------------------------------------------------------------------------------------------------------------
// The whole recursion in ONE launch when G = [A | B] fits in LDS (m <= 128): one 1024-thread workgroup per trajectory
// keeps G resident in LDS (row-major, lanes read consecutive k: conflict-free) and walks all T steps with one workgroup
// barrier per step; trajectories of a batch run side by side on different CUs.  The arithmetic per output row is that
// of lifted_step_kernel (lane-strided partial sums over z then u, shuffle tree, + bias), so both paths give the same
// bits.  Optionally the lift of the initial state, z_0 = K_mm^{-1/2} k(Z, x_0) (regressors.py:171-178), is done by the
// same workgroup first (wave per landmark / per row), so that a rollout is one kernel + one product with C.
// ---------------------------------------------------------------------------------------------------------------
struct ChainParams {
  const double* G; int64_t ldg; int m, pu;          // z' = G [z; u] + bias
  const double* z0; int64_t z0_stride;               // batch x m initial lifted states (lift == 0)
  int lift;                                          // 1: z0 = Sinv k(Zl, x0)
  const double* x0; int64_t x0_stride;               // batch x d states
  const double* Zl; int d; const double* winv; const double* Sinv; int ktype; double sigma0sq;
  const double* U; int64_t u_stride;                 // [b][t][pu] (u_stride = doubles per trajectory); may be null (pu == 0)
  const double* bias; int64_t bias_stride;           // [b][m] or shared (stride 0); may be null
  double* Zall; int64_t z_stride;                    // [b][t][m]
  int T;
};

constexpr int CHAIN_THREADS = 1024;

__device__ __forceinline__ double chain_kfun(int ktype, double acc, double sigma0sq) {
  if (ktype == NK_KERNEL_RBF) return exp(-0.5 * acc);
  if (ktype == NK_KERNEL_MATERN52) {
    const double t = sqrt(acc) * 2.23606797749978969641;
    return (1.0 + t + t * t / 3.0) * exp(-t);
  }
  return acc + sigma0sq;
}

__global__ void __launch_bounds__(CHAIN_THREADS) lifted_chain_kernel(ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int m = P.m, pu = P.pu, mpu = m + pu;
  const int ldgs = mpu + (mpu & 1);
  double* Gs = lds;                          // m x ldgs
  double* zu0 = Gs + (size_t)m * ldgs;       // two buffers of [z ; u]
  double* zu1 = zu0 + ldgs;
  double* kv = zu1 + ldgs;                   // m kernel values (lift)
  double* xw = kv + m + (m & 1);             // d scaled coordinates of x0 (lift)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = CHAIN_THREADS / 64;
  const int b = blockIdx.x;
  for (int e = tid; e < m * mpu; e += CHAIN_THREADS) {
    const int r = e / mpu, c = e - r * mpu;
    Gs[(size_t)r * ldgs + c] = P.G[(int64_t)r * P.ldg + c];
  }
  double* zall = P.Zall + (int64_t)b * P.z_stride;
  if (P.lift) {
    const double* x0 = P.x0 + (int64_t)b * P.x0_stride;
    const bool linear = P.ktype == NK_KERNEL_LINEAR;
    for (int k = tid; k < P.d; k += CHAIN_THREADS) xw[k] = x0[k] * (linear ? 1.0 : P.winv[k]);
    __syncthreads();
    for (int j = wave; j < m; j += nwaves) {
      const double* zj = P.Zl + (int64_t)j * P.d;
      double acc = 0.0;
      if (linear) {
        for (int k = lane; k < P.d; k += 64) acc = fma(zj[k], xw[k], acc);
      } else {
        for (int k = lane; k < P.d; k += 64) {
          const double t = zj[k] * P.winv[k] - xw[k];
          acc = fma(t, t, acc);
        }
      }
      acc = wave_sum64(acc);
      if (lane == 0) kv[j] = chain_kfun(P.ktype, acc, P.sigma0sq);
    }
    __syncthreads();
    // phi = k(x0, Z) Sinv: the product nk_lift forms (column r of Sinv; consecutive threads read consecutive addresses)
    if (tid < m) {
      double acc = 0.0;
      for (int k = 0; k < m; ++k) acc = fma(kv[k], P.Sinv[(int64_t)k * m + tid], acc);
      zu0[tid] = acc;
      zall[tid] = acc;
    }
  } else {
    const double* z0 = P.z0 + (int64_t)b * P.z0_stride;
    for (int k = tid; k < m; k += CHAIN_THREADS) { const double v = z0[k]; zu0[k] = v; zall[k] = v; }
  }
  const double* U = P.U ? P.U + (int64_t)b * P.u_stride : nullptr;
  const double* bias = P.bias ? P.bias + (int64_t)b * P.bias_stride : nullptr;
  if (tid < pu && P.T > 1) zu0[m + tid] = U[tid];
  __syncthreads();
  double* cur = zu0;
  double* nxt = zu1;
  for (int t = 0; t + 1 < P.T; ++t) {
    for (int r = wave; r < m; r += nwaves) {
      const double* g = Gs + (size_t)r * ldgs;
      double acc = 0.0;
      for (int k = lane; k < m; k += 64) acc = fma(g[k], cur[k], acc);
      for (int k = lane; k < pu; k += 64) acc = fma(g[m + k], cur[m + k], acc);
      const double sres = wave_sum64(acc);
      if (lane == 0) {
        const double v = sres + (bias ? bias[r] : 0.0);
        nxt[r] = v;
        zall[(int64_t)(t + 1) * m + r] = v;
      }
    }
    if (tid < pu && t + 2 < P.T) nxt[m + tid] = U[(int64_t)(t + 1) * pu + tid];
    __syncthreads();
    double* sw = cur; cur = nxt; nxt = sw;
  }
}

// D[b][t][:] = ref[b][:] - Phi[b][t][:]   (the argument of the feedback law u_t = K (phi_ref - phi_t) for all steps)
__global__ void ref_minus_traj_kernel(const double* __restrict__ ref, int64_t ref_stride, const double* __restrict__ Phi,
                                      int64_t phi_stride, double* __restrict__ D, int64_t d_stride, int steps, int m,
                                      int batch) {
  const int64_t total = (int64_t)batch * steps * m;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / ((int64_t)steps * m), r = e - b * steps * m;
    const int k = (int)(r % m);
    D[b * d_stride + r] = ref[b * ref_stride + k] - Phi[b * phi_stride + r];
  }
}
int launch_ref_minus_traj(nk_ctx* ctx, const double* ref, int64_t ref_stride, const double* Phi, int64_t phi_stride,
                          double* D, int64_t d_stride, int steps, int m, int batch) {
  const int64_t total = (int64_t)batch * steps * m;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ref_minus_traj_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ref, ref_stride, Phi,
                     phi_stride, D, d_stride, steps, m, batch);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

size_t lifted_chain_lds_bytes(int m, int pu, int d_lift) {
  const int mpu = m + pu, ldgs = mpu + (mpu & 1);
  return ((size_t)m * ldgs + 2 * (size_t)ldgs + (size_t)(m + (m & 1)) + (size_t)(d_lift > 0 ? d_lift : 0) + 2) * sizeof(double);
}
bool lifted_chain_ok(int m, int pu, int d_lift) {
  return m >= 1 && m <= 128 && pu >= 0 && pu <= CHAIN_THREADS && lifted_chain_lds_bytes(m, pu, d_lift) <= 160 * 1024;
}

static bool g_chain_attr_set = false;

int launch_lifted_chain(nk_ctx* ctx, const ChainArgs& a) {
  NK_REQUIRE(lifted_chain_ok(a.m, a.pu, a.lift ? a.d : 0), "lifted_chain: operators do not fit in LDS");
  NK_REQUIRE(a.batch >= 1 && a.T >= 1, "lifted_chain: bad sizes");
  const size_t bytes = lifted_chain_lds_bytes(a.m, a.pu, a.lift ? a.d : 0);
  if (!g_chain_attr_set) {
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lifted_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    g_chain_attr_set = true;
  }
  ChainParams P;
  P.G = a.G; P.ldg = a.ldg; P.m = a.m; P.pu = a.pu; P.z0 = a.z0; P.z0_stride = a.z0_stride; P.lift = a.lift ? 1 : 0;
  P.x0 = a.x0; P.x0_stride = a.x0_stride; P.Zl = a.Zl; P.d = a.d; P.winv = a.winv; P.Sinv = a.Sinv; P.ktype = a.ktype;
  P.sigma0sq = a.sigma0 * a.sigma0; P.U = a.pu > 0 ? a.U : nullptr; P.u_stride = a.u_stride; P.bias = a.bias;
  P.bias_stride = a.bias_stride; P.Zall = a.Zall; P.z_stride = a.z_stride; P.T = a.T;
  hipLaunchKernelGGL(lifted_chain_kernel, dim3(a.batch), dim3(CHAIN_THREADS), bytes, ctx->stream, P);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
