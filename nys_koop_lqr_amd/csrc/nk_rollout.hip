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
// The whole recursion in ONE launch when G = [A | B] fits in the REGISTERS of one workgroup (m <= 128, m + p <= 136):
// one 1024-thread workgroup per trajectory, thread (row = tid / 8, part = tid % 8) keeps G[row][part + 8 j], j < 17, in
// VGPRs for all T steps -- G is loop invariant, so a step touches LDS only for the current vector [z ; u] (8 distinct
// addresses per wave instruction, broadcast to the 8 rows of the wave: conflict-free), does 17 FMAs, three DPP adds
// across the 8 parts and one workgroup barrier.  Trajectories of a batch run side by side on different CUs.  Controls
// and bias are staged in LDS (a global / page-locked read inside the step loop would put 1-3 us on the critical path of
// every step).  Optionally the lift of the initial state, z_0 = K_mm^{-1/2} k(Z, x_0) (regressors.py:171-178), is done by
// the same workgroup first (wave per landmark), so that a rollout is one kernel + one product with C.
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
  int tb;                                            // steps per staged block of controls
  int d_pad;                                         // LDS doubles reserved for the scaled x0 (lift)
};

constexpr int CHAIN_THREADS = 1024;
constexpr int CHAIN_KPT = 17;               // G entries per thread
constexpr int CHAIN_ZU = 8 * CHAIN_KPT;     // padded length of [z ; u] in LDS (136)

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
  double* zu0 = lds;                         // two buffers of [z ; u], zero padded to CHAIN_ZU
  double* zu1 = zu0 + CHAIN_ZU;
  double* kv = zu1 + CHAIN_ZU;               // m kernel values (lift)
  double* biasS = kv + 128;                  // m
  double* xw = biasS + 128;                  // d scaled coordinates of x0 (lift)
  double* Ublk = xw + P.d_pad;               // P.tb x pu
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = CHAIN_THREADS / 64;
  const int row = tid >> 3, part = tid & 7;
  const int b = blockIdx.x;
  double g[CHAIN_KPT];
#pragma unroll
  for (int j = 0; j < CHAIN_KPT; ++j) {
    const int k = part + 8 * j;
    g[j] = (row < m && k < mpu) ? P.G[(int64_t)row * P.ldg + k] : 0.0;
  }
  if (tid < 2 * CHAIN_ZU) zu0[tid] = 0.0;    // both buffers: the padding must read as zero
  double* zall = P.Zall + (int64_t)b * P.z_stride;
  if (P.bias) {
    const double* bias = P.bias + (int64_t)b * P.bias_stride;
    if (tid < m) biasS[tid] = bias[tid];
  } else if (tid < m) {
    biasS[tid] = 0.0;
  }
  __syncthreads();
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
    if (tid < m) { const double v = z0[tid]; zu0[tid] = v; zall[tid] = v; }
  }
  const double* U = P.U ? P.U + (int64_t)b * P.u_stride : nullptr;
  const int kpt = (mpu + 7) >> 3;
  const bool wave_active = (wave << 3) < m;
  double* cur = zu0;
  double* nxt = zu1;
  for (int t0 = 0; t0 + 1 < P.T; t0 += P.tb) {
    const int steps = min(P.tb, P.T - 1 - t0);
    __syncthreads();  // the previous block's last reads of Ublk are done
    if (U) {
      for (int e = tid; e < steps * pu; e += CHAIN_THREADS) Ublk[e] = U[(int64_t)t0 * pu + e];
    }
    __syncthreads();
    if (U && tid < pu) cur[m + tid] = Ublk[tid];
    __syncthreads();
    for (int ts = 0; ts < steps; ++ts) {
      double acc = 0.0;
      if (wave_active) {  // waves whose 8 rows all lie beyond m idle (uniform branch); kpt = ceil((m + pu) / 8) <= 17
#pragma unroll
        for (int j = 0; j < CHAIN_KPT; ++j)
          if (j < kpt) acc = fma(g[j], cur[part + 8 * j], acc);
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      if (part == 0 && row < m) {
        const double v = acc + biasS[row];
        nxt[row] = v;
        zall[(int64_t)(t0 + ts + 1) * m + row] = v;
      }
      if (U && tid < pu && ts + 1 < steps) nxt[m + tid] = Ublk[(ts + 1) * pu + tid];
      // LDS-only barrier: __syncthreads() would also wait for the global store above to retire (vmcnt(0), ~1 us per step);
      // nothing in this workgroup reads zall back, so only the LDS writes need to be visible
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      double* sw = cur; cur = nxt; nxt = sw;
    }
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

static inline int chain_d_pad(int d_lift) { return d_lift > 0 ? d_lift + (d_lift & 1) : 0; }
// LDS doubles without the control block
static size_t chain_fixed_doubles(int d_lift) { return 2 * (size_t)CHAIN_ZU + 256 + (size_t)chain_d_pad(d_lift) + 2; }
constexpr size_t CHAIN_LDS_MAX = 64 * 1024;
bool lifted_chain_ok(int m, int pu, int d_lift) {
  if (!(m >= 1 && m <= 128 && pu >= 0 && m + pu <= CHAIN_ZU)) return false;
  return (chain_fixed_doubles(d_lift) + (size_t)pu * 16) * sizeof(double) <= CHAIN_LDS_MAX;  // >= 16 steps of controls per block
}

static bool g_chain_attr_set = false;

int launch_lifted_chain(nk_ctx* ctx, const ChainArgs& a) {
  const int dl = a.lift ? a.d : 0;
  NK_REQUIRE(lifted_chain_ok(a.m, a.pu, dl), "lifted_chain: operators do not fit in LDS");
  NK_REQUIRE(a.batch >= 1 && a.T >= 1, "lifted_chain: bad sizes");
  const int pu = a.U ? a.pu : 0;
  const size_t fixed = chain_fixed_doubles(dl) * sizeof(double);
  int tb = a.T > 1 ? a.T - 1 : 1;
  if (pu > 0) {
    const size_t room = (CHAIN_LDS_MAX - fixed) / (sizeof(double) * (size_t)pu);
    if ((size_t)tb > room) tb = (int)room;
    if (tb > 1024) tb = 1024;
  }
  const size_t bytes = fixed + (size_t)tb * (size_t)pu * sizeof(double);
  if (!g_chain_attr_set) {
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lifted_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)CHAIN_LDS_MAX));
    g_chain_attr_set = true;
  }
  ChainParams P;
  P.G = a.G; P.ldg = a.ldg; P.m = a.m; P.pu = a.pu; P.z0 = a.z0; P.z0_stride = a.z0_stride; P.lift = a.lift ? 1 : 0;
  P.x0 = a.x0; P.x0_stride = a.x0_stride; P.Zl = a.Zl; P.d = a.d; P.winv = a.winv; P.Sinv = a.Sinv; P.ktype = a.ktype;
  P.sigma0sq = a.sigma0 * a.sigma0; P.U = a.pu > 0 ? a.U : nullptr; P.u_stride = a.u_stride; P.bias = a.bias;
  P.bias_stride = a.bias_stride; P.Zall = a.Zall; P.z_stride = a.z_stride; P.T = a.T; P.tb = tb; P.d_pad = chain_d_pad(dl);
  hipLaunchKernelGGL(lifted_chain_kernel, dim3(a.batch), dim3(CHAIN_THREADS), bytes, ctx->stream, P);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
