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
// The whole recursion in ONE launch when G = [A | B] fits in the REGISTERS of one workgroup (m <= 128, m + p <= 160):
// one 1024-thread workgroup per trajectory.  Thread (quad = tid / 32, part = tid % 32) keeps the 4 x 5 entries
// G[4 quad + r][part + 32 j] in VGPRs for all T steps -- G is loop invariant, so a step touches LDS only for the current
// vector [z ; u]: 5 conflict-free ds_read_b64 per lane (the 32 lanes of a half wave read 32 consecutive doubles),
// 20 FMAs in four independent chains, and a cross-lane reduction of the four partial sums over the 32 parts that
// never touches LDS: v_permlane16_swap (gfx950) exchanges the halves of the two row pairs between the 16-lane rows, a
// row_mirror DPP move halves once more, three DPP moves finish; one workgroup barrier ends the step.  (The first
// version of this kernel kept one row per 8 lanes and read 17 values of [z ; u] per lane and step: with 16 waves the
// LDS pipe alone was busy ~1100 cycles of the 2250 a step took; now ~320, and a step takes ~1100.)  Trajectories of a
// batch run side by side on different CUs.  Controls and bias are staged in LDS (a global / page-locked read inside the
// step loop would put 1-3 us on the critical path of every step).  Optionally the lift of the initial state,
// z_0 = k(x_0, Z) K_mm^{-1/2} (regressors.py:171-178), is done by the same workgroup first (wave per landmark for the
// kernel values, then the same register-tile product with S^-1), so that a rollout is one kernel + one product with C.
// ---------------------------------------------------------------------------------------------------------------
struct ChainParams {
  const double* G; int64_t ldg; int m, pu;          // z' = G [z; u] + bias
  const double* z0; int64_t z0_stride;               // batch x m initial lifted states (lift == 0)
  int lift;                                          // 1: z0 = k(x0, Zl) Sinv
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
constexpr int CHAIN_KPT = 5;                // columns of G per thread (stride 32)
constexpr int CHAIN_ZU = 32 * CHAIN_KPT;    // padded length of [z ; u] in LDS (160)

__device__ __forceinline__ double chain_kfun(int ktype, double acc, double sigma0sq) {
  if (ktype == NK_KERNEL_RBF) return exp_nonpos(-0.5 * acc);
  if (ktype == NK_KERNEL_MATERN52) {
    const double t = sqrt(acc) * 2.23606797749978969641;
    return (1.0 + t + t * t / 3.0) * exp_nonpos(-t);
  }
  return acc + sigma0sq;
}

// Four values per lane, summed over the 32 lanes of a half wave: afterwards every lane holds the total of value
// r = 2 * bit4(lane) + bit3(lane) (checked lane by lane on the device by tools/reduce_probe.hip).  Fixed order: the
// result does not depend on anything but the inputs.
__device__ __forceinline__ double reduce_4rows_32parts(double a0, double a1, double a2, double a3, int lane) {
  swap16_f64(a0, a2);
  const double e0 = a0 + a2;  // lanes with bit4 = 0: value 0, bit4 = 1: value 2 (each over parts {q, q + 16})
  swap16_f64(a1, a3);
  const double e1 = a1 + a3;  // value 1 / value 3
  const bool hi8 = (lane & 8) != 0;
  const double keep = hi8 ? e1 : e0, send = hi8 ? e0 : e1;
  double c = keep + dpp_f64<0x140>(send);  // row_mirror: lane i <-> 15 - i
  c += dpp_f64<0xB1>(c);                   // quad_perm [1,0,3,2]
  c += dpp_f64<0x4E>(c);                   // quad_perm [2,3,0,1]
  c += dpp_f64<0x141>(c);                  // row_half_mirror: lane i <-> 7 - i
  return c;
}

__global__ void __launch_bounds__(CHAIN_THREADS) lifted_chain_kernel(ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int m = P.m, pu = P.pu, mpu = m + pu;
  double* zu0 = lds;                         // two buffers of [z ; u], zero padded to CHAIN_ZU
  double* zu1 = zu0 + CHAIN_ZU;
  double* kv = zu1 + CHAIN_ZU;               // m kernel values (lift), zero padded to CHAIN_ZU
  double* biasS = kv + CHAIN_ZU;             // m
  double* xw = biasS + 128;                  // d scaled coordinates of x0 (lift)
  double* Ublk = xw + P.d_pad;               // P.tb x pu
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = CHAIN_THREADS / 64;
  const int part = tid & 31, row0 = (tid >> 5) * 4;
  // the lane that ends up with the total of row `myrow` after the reduction, and whether it is the one that stores it
  const int myrow = row0 + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1);
  const bool writer = (lane & 7) == 0 && myrow < m;
  const bool wave_active = row0 - (lane >> 5) * 4 < m;  // wave-uniform: the first row of the wave exists
  const int b = blockIdx.x;
  if (tid < 3 * CHAIN_ZU) zu0[tid] = 0.0;    // both vectors and kv: the padding must read as zero
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
    // phi[r] = sum_k kv[k] Sinv[k][r]: the product nk_lift forms, with the register tiling of the steps below
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (wave_active) {
#pragma unroll
      for (int j = 0; j < CHAIN_KPT - 1; ++j) {  // k < 128
        const int k = part + 32 * j;
        if (k < m) {
          const double v = kv[k];
          const double* srow = P.Sinv + (int64_t)k * m + row0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < m) a[r] = fma(v, srow[r], a[r]);
        }
      }
    }
    const double c = reduce_4rows_32parts(a[0], a[1], a[2], a[3], lane);
    if (writer) { zu0[myrow] = c; zall[myrow] = c; }
  } else {
    const double* z0 = P.z0 + (int64_t)b * P.z0_stride;
    if (tid < m) { const double v = z0[tid]; zu0[tid] = v; zall[tid] = v; }
  }
  double g[4][CHAIN_KPT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int j = 0; j < CHAIN_KPT; ++j) {
      const int k = part + 32 * j;
      g[r][j] = (row0 + r < m && k < mpu) ? P.G[(int64_t)(row0 + r) * P.ldg + k] : 0.0;
    }
  const double mybias = writer ? biasS[myrow] : 0.0;
  const double* U = P.U ? P.U + (int64_t)b * P.u_stride : nullptr;
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
      double a[4] = {0.0, 0.0, 0.0, 0.0};
      // all LDS reads of the step are issued together (one wait): the padding of [z ; u] is zero and so are the entries
      // of g beyond m + pu, so no read is conditional
      const bool ucopy = U != nullptr && tid < pu && ts + 1 < steps;
      double unext = 0.0;
      if (ucopy) unext = Ublk[(ts + 1) * pu + tid];
      if (wave_active) {  // waves whose rows all lie beyond m idle (uniform branch)
        double v[CHAIN_KPT];
#pragma unroll
        for (int j = 0; j < CHAIN_KPT; ++j) v[j] = cur[part + 32 * j];
#pragma unroll
        for (int j = 0; j < CHAIN_KPT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] = fma(g[r][j], v[j], a[r]);
      }
      const double c = reduce_4rows_32parts(a[0], a[1], a[2], a[3], lane);
      if (writer) {
        const double v = c + mybias;
        nxt[myrow] = v;
        zall[(int64_t)(t0 + ts + 1) * m + myrow] = v;
      }
      if (ucopy) nxt[m + tid] = unext;
      // LDS-only barrier: __syncthreads() would also wait for the global store above to retire (vmcnt(0), ~1 us per step);
      // nothing in this workgroup reads zall back, so only the LDS writes need to be visible
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      double* sw = cur; cur = nxt; nxt = sw;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The recursion in ONE launch for m > 128 (the real cloth model has m = 500, the headline model m = 2000): G = [A | B]
// is spread over the registers of W = ceil(m / 16) workgroups of 16 waves -- wave r of the launch keeps row r of G (lane
// l holds G[r][l + 64 j]) -- and the workgroups exchange the state through memory without any grid barrier: the rows
// t >= 1 of Zall are pre-filled with a sentinel (a signalling-NaN bit pattern, which no arithmetic result can have);
// in step t wave w of a workgroup fetches chunk(s) w, w + 16, ... of z_t with agent-scope (cross-XCD coherent) loads,
// repeating until no entry is the sentinel, and parks them in LDS; after one workgroup barrier every wave forms its dot
// product from LDS and publishes z_{t+1}[r] with one agent-scope store.  Every 8-byte slot of Zall is written exactly
// once, so a slot is its own "ready" flag: one memory round trip per step instead of a launch per step, and only
// W * ceil(m / 64) polling loads per round (all waves polling all of z_t made the 4 KB of z_t a hot spot: 4.7 us per
// step at m = 500).
// Progress: a workgroup only waits for workgroups of the same trajectory and the host never puts more workgroups in a
// launch than the device holds at once (W * trajectories <= number of CUs; with more, workgroups of later trajectories
// can fill an XCD's slots while they wait for siblings that then never get a slot), so every workgroup of the launch is
// resident once the kernels ahead of it in other queues have drained.  Two such launches side by side could still
// starve each other, so the host serialises them (nk_api.hip) and a wave that has polled MW_POLL_LIMIT times gives up:
// it raises *status and publishes NaN, which releases everybody behind it -- the grid always drains, and the caller
// repeats the recursion with one launch per step.
// ---------------------------------------------------------------------------------------------------------------
constexpr int MW_KPT = 32;                                     // most 64-wide chunks of the state per lane: m <= 2048
constexpr int MW_ROWS = 16;                                    // rows (waves) per workgroup
constexpr unsigned long long MW_SENTINEL = 0x7FF4A5C3E1D2B697ull;  // exponent all ones, quiet bit clear: signalling NaN
constexpr unsigned long long MW_QNAN = 0x7FF8000000000000ull;
constexpr int MW_POLL_LIMIT = 1 << 19;

struct ChainMwParams {
  const double* G; int64_t ldg; int m, pu, kpt;
  const double* U; int64_t u_stride;                 // device memory, [b][t][pu]
  const double* bias; int64_t bias_stride;
  double* Zall; int64_t z_stride;                    // [b][t][m]; row 0 holds z_0, rows >= 1 the sentinel
  int T;
  int batch;                                         // trajectories of the launch (grid.y = ceil(batch / NT))
  int* status;                                       // [0] raised on a timeout, [1..3] = row, step, trajectory of the first
};

__global__ void mw_sentinel_fill_kernel(double* __restrict__ Zall, int64_t z_stride, int m, int T, int batch) {
  const int64_t per = (int64_t)(T - 1) * m, total = per * batch;
  unsigned long long* z = reinterpret_cast<unsigned long long*>(Zall);
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / per, r = e - b * per;
    z[b * z_stride + m + r] = MW_SENTINEL;
  }
}

// KPT = number of 64-wide chunks of z_t (compile time: the LDS reads and FMAs of a step are straight-line code).  The
// chunk past m re-reads z[m - 1] (a valid address) and parks zeros.  The controls are a separate term: lane l < pu holds
// G[r][m + l].
// NT = trajectories a workgroup advances together (grid.y = ceil(batch / NT)): the rows of G in registers are shared by
// the NT dot products of a step, so a batch needs NT times fewer workgroups (20 test trajectories at m = 500: 5 x 32
// workgroups, one launch, where 20 x 32 would not be resident together and went through one GEMM per step).
template <int KPT, int NT>
__global__ void __launch_bounds__(64 * MW_ROWS) lifted_chain_mw_kernel(ChainMwParams P) {
  __shared__ double zs[2][NT][KPT * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * MW_ROWS + wave;
  const int m = P.m, pu = P.pu;
  const bool active = row < m;  // idle waves of the last workgroup still fetch their chunks and meet the barriers
  const int b0 = blockIdx.y * NT;
  double g[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int k = lane + 64 * j;
    g[j] = (active && k < m) ? P.G[(int64_t)row * P.ldg + k] : 0.0;
  }
  const bool have_u = P.U != nullptr && pu > 0;
  const double gu = (active && have_u && lane < pu) ? P.G[(int64_t)row * P.ldg + m + lane] : 0.0;
  unsigned long long* zall[NT];
  const double* U[NT];
  double bias[NT];
  bool live[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    live[i] = b0 + i < P.batch;
    const int b = live[i] ? b0 + i : b0;  // a padding slot shadows trajectory b0 (reads only)
    zall[i] = reinterpret_cast<unsigned long long*>(P.Zall + (int64_t)b * P.z_stride);
    U[i] = have_u ? P.U + (int64_t)b * P.u_stride : nullptr;
    bias[i] = (active && P.bias) ? P.bias[(int64_t)b * P.bias_stride + row] : 0.0;
  }
  for (int t = 0; t + 1 < P.T; ++t) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (!live[i]) continue;  // uniform over the workgroup
      const unsigned long long* zt = zall[i] + (int64_t)t * m;
      double* buf = zs[t & 1][i];
      for (int j = wave; j < KPT; j += MW_ROWS) {  // wave-uniform
        const int k = lane + 64 * j;
        const unsigned long long* src = zt + (k < m ? k : m - 1);
        unsigned long long bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t > 0) {  // z_0 is the caller's data: taken as it is
          int polls = 0;
          while (__any(bits == MW_SENTINEL)) {
            if (++polls >= MW_POLL_LIMIT) {
              if (lane == 0 && atomicCAS(P.status, 0, 1) == 0) { P.status[1] = row; P.status[2] = t; P.status[3] = b0 + i; }
              bits = MW_QNAN;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        buf[k] = k < m ? __longlong_as_double((long long)bits) : 0.0;  // (0 x an overflowed z[m - 1] would be NaN)
      }
    }
    // LDS-only barrier (the global store of the previous step need not have retired); the other buffer is free again
    // once every wave has passed THIS barrier, i.e. before anybody writes it in step t + 1
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (!live[i]) continue;
      const double* buf = zs[t & 1][i];
      double a0 = (have_u && lane < pu) ? gu * U[i][(int64_t)t * pu + lane] : 0.0;
      double a1 = 0.0;
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const double v = buf[lane + 64 * j];
        if (j & 1) a1 = fma(g[j], v, a1);
        else a0 = fma(g[j], v, a0);
      }
      const double acc = wave_sum64_dpp(a0 + a1);
      if (active && lane == 0) {
        unsigned long long out = (unsigned long long)__double_as_longlong(acc + bias[i]);
        if (out == MW_SENTINEL) out = MW_QNAN;
        __hip_atomic_store(zall[i] + (int64_t)(t + 1) * m + row, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// trajectories per workgroup for a batch: NT * ceil(m / 64) <= 32 (the LDS image of the NT state vectors, two buffers)
int chain_mw_group(int m, int batch) {
  const int kpt = (m + 63) / 64;
  int nt = 1;
  while (nt < 4 && 2 * nt * (kpt <= 4 ? 4 : kpt <= 8 ? 8 : kpt <= 16 ? 16 : 32) <= 32 && 2 * nt <= batch) nt *= 2;
  return nt;
}
int chain_mw_workgroups(int m) { return (m + MW_ROWS - 1) / MW_ROWS; }
bool lifted_chain_mw_ok(const nk_ctx* ctx, int m, int pu) {
  return m >= 1 && m <= 64 * MW_KPT && pu >= 0 && pu <= 64 && chain_mw_workgroups(m) <= ctx->num_cu;
}

// Zall row 0 of every trajectory holds z_0; U (if any) and bias are in DEVICE memory; batch * workgroups <= CUs.  The
// verdict (a wave gave up waiting) lands in ctx->h_info[12..15] with the caller's next synchronisation of ctx->stream:
// lifted_chain_mw_timed_out.
int launch_lifted_chain_mw(nk_ctx* ctx, const ChainArgs& a) {
  NK_REQUIRE(lifted_chain_mw_ok(ctx, a.m, a.U ? a.pu : 0), "lifted_chain_mw: operators do not fit");
  NK_REQUIRE(a.batch >= 1 && a.T >= 1 && !a.lift && a.z0 == nullptr, "lifted_chain_mw: z_0 must be in place");
  const int nt = chain_mw_group(a.m, a.batch);
  const int groups = (a.batch + nt - 1) / nt;
  NK_REQUIRE((int64_t)groups * chain_mw_workgroups(a.m) <= ctx->num_cu, "lifted_chain_mw: more workgroups than CUs");
  if (a.T < 2) return NK_OK;
  const int pu = a.U ? a.pu : 0;
  ChainMwParams P;
  P.G = a.G; P.ldg = a.ldg; P.m = a.m; P.pu = pu; P.kpt = (a.m + 63) / 64; P.U = pu > 0 ? a.U : nullptr;
  P.u_stride = a.u_stride; P.bias = a.bias; P.bias_stride = a.bias_stride; P.Zall = a.Zall; P.z_stride = a.z_stride;
  P.T = a.T; P.batch = a.batch; P.status = ctx->d_info + 12;
  const int64_t total = (int64_t)a.batch * (a.T - 1) * a.m;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mw_sentinel_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a.Zall, a.z_stride, a.m,
                     a.T, a.batch);
  const dim3 grid((unsigned)chain_mw_workgroups(a.m), (unsigned)groups), block(64 * MW_ROWS);
#define NK_MW_LAUNCH(K, N) hipLaunchKernelGGL((lifted_chain_mw_kernel<K, N>), grid, block, 0, ctx->stream, P)
  if (P.kpt <= 4) {
    if (nt == 4) NK_MW_LAUNCH(4, 4); else if (nt == 2) NK_MW_LAUNCH(4, 2); else NK_MW_LAUNCH(4, 1);
  } else if (P.kpt <= 8) {
    if (nt == 4) NK_MW_LAUNCH(8, 4); else if (nt == 2) NK_MW_LAUNCH(8, 2); else NK_MW_LAUNCH(8, 1);
  } else if (P.kpt <= 16) {
    if (nt == 2) NK_MW_LAUNCH(16, 2); else NK_MW_LAUNCH(16, 1);
  } else {
    NK_MW_LAUNCH(MW_KPT, 1);
  }
#undef NK_MW_LAUNCH
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int lifted_chain_mw_reset(nk_ctx* ctx) {
  NK_HIP(hipMemsetAsync(ctx->d_info + 12, 0, 4 * sizeof(int), ctx->stream));
  return NK_OK;
}
int lifted_chain_mw_fetch_status(nk_ctx* ctx) {
  NK_HIP(hipMemcpyAsync(ctx->h_info + 12, ctx->d_info + 12, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  return NK_OK;
}
bool lifted_chain_mw_timed_out(nk_ctx* ctx, int* row, int* step, int* traj) {
  if (ctx->h_info[12] == 0) return false;
  ctx->h_info[12] = 0;
  if (row) *row = ctx->h_info[13];
  if (step) *step = ctx->h_info[14];
  if (traj) *traj = ctx->h_info[15];
  return true;
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
static size_t chain_fixed_doubles(int d_lift) { return 3 * (size_t)CHAIN_ZU + 128 + (size_t)chain_d_pad(d_lift) + 2; }
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
