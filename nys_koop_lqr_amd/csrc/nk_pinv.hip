// Rank-revealing fallback of the two regularised solves (regressors.py:155,165): the reference calls
// scipy.linalg.lstsq, i.e. LAPACK gelsd, which returns the MINIMUM-NORM solution with every singular value below a
// cut-off relative to the largest one treated as zero.  The fast path (blocked Cholesky, nk_linalg.hip) has no such
// notion: on a numerically rank-deficient system it meets a non-positive (or rounding-level) pivot.  This file provides
// the truncated pseudo-inverse for that case, entirely on the device:
//
//   one-sided (Hestenes) Jacobi SVD of the symmetric m x m matrix P.  W starts as P, V as I, both stored so that logical
//   column j is memory row j (P is symmetric, so W_0 = P needs no transpose).  A rotation of the pair (p, q) makes rows p
//   and q of W orthogonal and is applied to the same rows of V; at convergence  P V = W^T  with mutually orthogonal
//   columns u_j = row j of W, sigma_j = |u_j|, hence
//        P^+ = sum_{sigma_j > rcond * sigma_max}  sigma_j^-2  v_j u_j^T ,     E P^+ = ((E V) diag(sigma^-2 | 0)) W .
//   A sweep is m-1 rounds of m/2 independent pairs (round-robin tournament order), one launch per round and one
//   workgroup per pair; the host reads one counter per sweep.  The iteration runs to full convergence (every pair
//   orthogonal to m * eps): one-sided Jacobi then resolves an exact null space to ~1e-20 sigma_max (measured on the
//   duplicated-landmark systems of tests/golden/f9), far below the cut-off, where bidiagonalisation-based LAPACK drivers
//   leave it at (0.1..5) eps sigma_max, on either side of it (the same golden records gelsd keeping 41 singular values
//   of a rank-40 matrix).  Columns a factor 1000 below the cut-off are left out of further rotations.
//
// Cut-off: gelsd's, as the reference calls it (scipy.linalg.lstsq default): singular values <= eps * sigma_max are dropped
// -- plus the isolated-cluster rule documented at pinv_scale_kernel for matrices with an exact null space, whose
// rounding-level singular values a general symmetric matrix leaves at (1..50) eps sigma_max even under Jacobi.
//
// This is the rare path: a 500 x 500 system of the cloth grid takes 20-35 sweeps (the iteration works on P itself, i.e. on
// the squared spectrum of a factor, and converges linearly for most of them) of 505 rounds, each of which reads and
// writes all of W and V (4 MB): 0.14 s alone -- bound by that traffic, not by the launches; a 2000 x 2000 one takes seconds.
// It is never entered when the Cholesky succeeds with pivots above the rounding level.
#include "nk_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace nk {

__device__ __forceinline__ double wave_sum_p(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// round-robin tournament over N players (N even): round r in [0, N-1), pair k in [0, N/2)
__device__ __forceinline__ void rr_pair(int N, int r, int k, int* a, int* b) {
  if (k == 0) {
    *a = N - 1;
    *b = r;
  } else {
    *a = (r + k) % (N - 1);
    *b = (r - k + (N - 1)) % (N - 1);
  }
}

constexpr int JAC_KPT = 8;  // entries of a row per thread held in registers (m <= 2048)
// Rotation that makes two rows with squared norms a, b and inner product c orthogonal.  False: leave the pair alone (one of
// them is below the dead-column threshold, or they are orthogonal to `tol` already: c^2 <= tol^2 a b, tested without square
// roots).  The dependent chain is what an inner round of the block kernel waits for: one division, one square root, one
// more division and one reciprocal square root.
__device__ __forceinline__ bool jacobi_rotation(double a, double b, double c, double tol, double small2, double* cs, double* sn) {
  if (a < small2 || b < small2 || c == 0.0 || !(c * c > (tol * tol) * a * b)) return false;
  const double zeta = (b - a) / (2.0 * c);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
  *cs = rsqrt(fma(t, t, 1.0));
  *sn = *cs * t;
  return true;
}

__device__ __forceinline__ void jacobi_round_kernel_body(double* __restrict__ W, double* __restrict__ V, int m, int N, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_count) {
  __shared__ double sh[3][4];
  __shared__ double cs_sn[2];
  int p, q;
  rr_pair(N, round, blockIdx.x, &p, &q);
  if (p >= m || q >= m) return;  // the dummy player of an odd m
  if (p > q) { const int t = p; p = q; q = t; }
  double* wp = W + (int64_t)p * m;
  double* wq = W + (int64_t)q * m;
  double* vp = V + (int64_t)p * m;
  double* vq = V + (int64_t)q * m;
  // The launch is latency bound (one workgroup per pair, a few hundred workgroups): the four rows are fetched ONCE, all
  // loads in flight together, and stay in registers for the rotation (m <= 2048; longer rows are read again below).
  const bool in_regs = m <= 256 * JAC_KPT;
  double x[JAC_KPT], y[JAC_KPT], vx[JAC_KPT], vy[JAC_KPT];
  double a = 0.0, b = 0.0, c = 0.0;
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < JAC_KPT; ++j) {
      const int k = threadIdx.x + 256 * j;
      const bool ok = k < m;
      x[j] = ok ? wp[k] : 0.0;
      y[j] = ok ? wq[k] : 0.0;
      vx[j] = ok ? vp[k] : 0.0;
      vy[j] = ok ? vq[k] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < JAC_KPT; ++j) {  // same order of accumulation as the strided loop below
      a = fma(x[j], x[j], a);
      b = fma(y[j], y[j], b);
      c = fma(x[j], y[j], c);
    }
  } else {
    for (int k = threadIdx.x; k < m; k += 256) {
      const double xx = wp[k], yy = wq[k];
      a = fma(xx, xx, a);
      b = fma(yy, yy, b);
      c = fma(xx, yy, c);
    }
  }
  a = wave_sum_p(a); b = wave_sum_p(b); c = wave_sum_p(c);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = b; sh[2][w] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    b = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    c = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    double cs = 1.0, sn = 0.0;
    if (jacobi_rotation(a, b, c, tol, d_small2[0], &cs, &sn)) {  // (a column below the cut-off takes no further part)
      rot_count[blockIdx.x] += 1;  // one slot per pair of the round: no contended atomic (250 same-address atomics from
                                   // all XCDs cost ~25 us per launch, more than everything else in the round)
    }
    cs_sn[0] = cs;
    cs_sn[1] = sn;
  }
  __syncthreads();
  const double cs = cs_sn[0], sn = cs_sn[1];
  if (sn == 0.0) return;
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < JAC_KPT; ++j) {
      const int k = threadIdx.x + 256 * j;
      if (k < m) {
        wp[k] = cs * x[j] - sn * y[j];
        wq[k] = sn * x[j] + cs * y[j];
        vp[k] = cs * vx[j] - sn * vy[j];
        vq[k] = sn * vx[j] + cs * vy[j];
      }
    }
    return;
  }
  for (int k = threadIdx.x; k < m; k += 256) {
    const double xx = wp[k], yy = wq[k];
    wp[k] = cs * xx - sn * yy;
    wq[k] = sn * xx + cs * yy;
    const double vxx = vp[k], vyy = vq[k];
    vp[k] = cs * vxx - sn * vyy;
    vq[k] = sn * vxx + cs * vyy;
  }
}
__global__ void __launch_bounds__(256) jacobi_round_kernel(double* __restrict__ W, double* __restrict__ V, int m, int N, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_count) { jacobi_round_kernel_body(W, V, m, N, round, tol, d_small2, rot_count); }
NK_BATCHED_TWIN(jacobi_round_kernel, (256), double*, double*, int, int, int, double, const double*, int*)

// m <= 512: one WAVE per pair (four pairs per workgroup), the rows in registers, no LDS and no workgroup barrier.  With a
// workgroup per pair a launch over several lock-step units (9 x 253 workgroups of 256 threads) did not fit the chip at once
// and ran as 2-3 rounds of latency-bound workgroups (26-33 us per launch, profiles of the cloth grid); as waves all pairs
// are resident together.
__device__ __forceinline__ double wave_allsum_p(double v) { return wave_sum64_dpp(v); }  // every lane gets the total
__device__ __forceinline__ void jacobi_round_wave_kernel_body(double* __restrict__ W, double* __restrict__ V, int m, int N, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_count) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= N / 2) return;
  int p, q;
  rr_pair(N, round, pair, &p, &q);
  if (p >= m || q >= m) return;  // the dummy player of an odd m
  if (p > q) { const int t = p; p = q; q = t; }
  double* wp = W + (int64_t)p * m;
  double* wq = W + (int64_t)q * m;
  double* vp = V + (int64_t)p * m;
  double* vq = V + (int64_t)q * m;
  double x[JAC_KPT], y[JAC_KPT], vx[JAC_KPT], vy[JAC_KPT];
#pragma unroll
  for (int j = 0; j < JAC_KPT; ++j) {
    const int k = lane + 64 * j;
    const bool ok = k < m;
    x[j] = ok ? wp[k] : 0.0;
    y[j] = ok ? wq[k] : 0.0;
    vx[j] = ok ? vp[k] : 0.0;
    vy[j] = ok ? vq[k] : 0.0;
  }
  double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
  for (int j = 0; j < JAC_KPT; ++j) {
    a = fma(x[j], x[j], a);
    b = fma(y[j], y[j], b);
    c = fma(x[j], y[j], c);
  }
  a = wave_allsum_p(a); b = wave_allsum_p(b); c = wave_allsum_p(c);
  double cs = 1.0, sn = 0.0;
  if (!jacobi_rotation(a, b, c, tol, d_small2[0], &cs, &sn)) return;  // wave-uniform
  if (lane == 0) rot_count[pair] += 1;  // one slot per pair of the round (see jacobi_round_kernel)
  if (sn == 0.0) return;
#pragma unroll
  for (int j = 0; j < JAC_KPT; ++j) {
    const int k = lane + 64 * j;
    if (k < m) {
      wp[k] = cs * x[j] - sn * y[j];
      wq[k] = sn * x[j] + cs * y[j];
      vp[k] = cs * vx[j] - sn * vy[j];
      vq[k] = sn * vx[j] + cs * vy[j];
    }
  }
}
__global__ void __launch_bounds__(256) jacobi_round_wave_kernel(double* __restrict__ W, double* __restrict__ V, int m, int N, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_count) { jacobi_round_wave_kernel_body(W, V, m, N, round, tol, d_small2, rot_count); }
NK_BATCHED_TWIN(jacobi_round_wave_kernel, (256), double*, double*, int, int, int, double, const double*, int*)

// A whole SWEEP (N - 1 rounds) in one launch for m <= 512: 16 pairs per 1024-thread workgroup, at most 16 workgroups, a
// software barrier between rounds (one agent-scope atomic per workgroup and round; every workgroup is resident: 16
// workgroups of this size fit 16 CUs, and a merged lock-step launch of 32 units 512 slots of the chip's 512).  A kernel
// boundary per round costs ~25 us here -- not the launch itself but the L2 write-back / invalidate of the 4 MB of W and V
// that every round rewrites (profiles of the cloth grid: 27 us per round launch whatever the kernel does) -- so the rows
// are read and written with agent-scope accesses instead, which stay coherent across XCDs without flushing anything.
// sync[0] = arrival counter (zeroed by the host before the launch), sync[1] = raised when a wait gives up (the host then
// continues with one launch per round; the matrices are left in a consistent state: a rotation is either fully applied
// or not at all, because a wave only gives up between rounds).
constexpr int JAC_SPIN_LIMIT = 1 << 20;
__device__ __forceinline__ double ld_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void jacobi_sweep_kernel_body(double* __restrict__ W, double* __restrict__ V, int m, int N, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots, int* __restrict__ sync) {
  __shared__ int give_up;
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 16 + (threadIdx.x >> 6);
  const int nwg = gridDim.x;
  const double small2 = d_small2[0];
  int rotations = 0;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  for (int round = 0; round < N - 1; ++round) {
    int p = m, q = m;
    if (pair < N / 2) rr_pair(N, round, pair, &p, &q);
    if (p < m && q < m) {  // wave-uniform (the dummy player of an odd m, the padding waves of the last workgroup)
      if (p > q) { const int t = p; p = q; q = t; }
      double* wp = W + (int64_t)p * m;
      double* wq = W + (int64_t)q * m;
      double* vp = V + (int64_t)p * m;
      double* vq = V + (int64_t)q * m;
      double x[JAC_KPT], y[JAC_KPT], vx[JAC_KPT], vy[JAC_KPT];
#pragma unroll
      for (int j = 0; j < JAC_KPT; ++j) {
        const int k = lane + 64 * j;
        const bool ok = k < m;
        x[j] = ok ? ld_agent(wp + k) : 0.0;
        y[j] = ok ? ld_agent(wq + k) : 0.0;
        vx[j] = ok ? ld_agent(vp + k) : 0.0;
        vy[j] = ok ? ld_agent(vq + k) : 0.0;
      }
      double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
      for (int j = 0; j < JAC_KPT; ++j) {
        a = fma(x[j], x[j], a);
        b = fma(y[j], y[j], b);
        c = fma(x[j], y[j], c);
      }
      a = wave_allsum_p(a); b = wave_allsum_p(b); c = wave_allsum_p(c);
      double cs = 1.0, sn = 0.0;
      if (jacobi_rotation(a, b, c, tol, small2, &cs, &sn)) {
        ++rotations;
        if (sn != 0.0) {
#pragma unroll
          for (int j = 0; j < JAC_KPT; ++j) {
            const int k = lane + 64 * j;
            if (k < m) {
              st_agent(wp + k, cs * x[j] - sn * y[j]);
              st_agent(wq + k, sn * x[j] + cs * y[j]);
              st_agent(vp + k, cs * vx[j] - sn * vy[j]);
              st_agent(vq + k, sn * vx[j] + cs * vy[j]);
            }
          }
        }
      }
    }
    if (round + 1 == N - 1) break;  // nobody reads after the last round: the kernel boundary publishes it
    // ---- barrier over the launch's workgroups (of this unit): the stores above have been acknowledged (vmcnt(0) in
    //      __syncthreads) before thread 0 announces the workgroup
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const int target = nwg * (round + 1);
      int polls = 0;
      while (__hip_atomic_load(sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || ++polls >= JAC_SPIN_LIMIT) {
          __hip_atomic_store(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          give_up = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (give_up) break;
  }
  if (lane == 0 && pair < N / 2) rot_slots[pair] += rotations;
}
__global__ void __launch_bounds__(1024) jacobi_sweep_kernel(double* __restrict__ W, double* __restrict__ V, int m, int N, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots, int* __restrict__ sync) { jacobi_sweep_kernel_body(W, V, m, N, tol, d_small2, rot_slots, sync); }
NK_BATCHED_TWIN(jacobi_sweep_kernel, (1024), double*, double*, int, int, double, const double*, int*, int*)

// ---------------------------------------------------------------------------------------------------------------
// Block Jacobi (m <= 2048): the scalar rounds above move ALL of W and V through memory once per round, 505 times per
// sweep at m = 506 -- the fallback is bound by that traffic (4 GB per sweep and unit), not by launches or arithmetic.
// Here the rows are grouped in blocks of B; a workgroup takes a PAIR of blocks (2B rows of W and of V, 128 KB) into LDS
// and runs a full round-robin sweep over the 2B rows there (2B - 1 inner rounds, one wave per pair, the four rows of a
// pair in registers between the dot products and the rotation), then writes the rows back.  An outer sweep is a
// round-robin over the blocks: NB - 1 launches of NB / 2 workgroups, each of which moves W and V once -- 8 x fewer bytes
// and launches per sweep at B = 8.  Every pair of rows meets at least once per outer sweep (pairs inside a block meet in
// every round their block takes part in, where they are already orthogonal and are skipped), so "no rotation in a whole
// outer sweep" is the same convergence criterion as before.  B x row length <= 4096 doubles: B = 8 for m <= 512, 4 for
// m <= 1024, 2 for m <= 2048.
// ---------------------------------------------------------------------------------------------------------------
template <int B, int KPT>
__device__ __forceinline__ void jacobi_block_body(double* __restrict__ W, double* __restrict__ V, int m, int NB, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots) {
  extern __shared__ __attribute__((aligned(16))) double jb_lds[];
  constexpr int LDW = 64 * KPT;
  double* Ws = jb_lds;
  double* Vs = jb_lds + 2 * B * LDW;
  __shared__ int rot_wg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bi, bj;
  rr_pair(NB, round, blockIdx.x, &bi, &bj);
  if (bi > bj) { const int t = bi; bi = bj; bj = t; }
  if (bi * B >= m) return;  // both blocks are padding
  if (threadIdx.x == 0) rot_wg = 0;
  // global row of local row r
  auto grow = [&](int r) { return r < B ? bi * B + r : bj * B + (r - B); };
  for (int r = wave; r < 2 * B; r += B) {
    const int gi = grow(r);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = lane + 64 * j;
      const bool ok = gi < m && k < m;
      Ws[r * LDW + k] = ok ? W[(int64_t)gi * m + k] : 0.0;
      Vs[r * LDW + k] = ok ? V[(int64_t)gi * m + k] : 0.0;
    }
  }
  __syncthreads();
  const double small2 = d_small2[0];
  int rotations = 0;
  for (int ir = 0; ir < 2 * B - 1; ++ir) {
    int p, q;
    rr_pair(2 * B, ir, wave, &p, &q);
    if (p > q) { const int t = p; p = q; q = t; }
    if (grow(p) < m && grow(q) < m) {  // wave-uniform
      double* wp = Ws + p * LDW;
      double* wq = Ws + q * LDW;
      double x[KPT], y[KPT];
      double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        x[j] = wp[lane + 64 * j];
        y[j] = wq[lane + 64 * j];
        a = fma(x[j], x[j], a);
        b = fma(y[j], y[j], b);
        c = fma(x[j], y[j], c);
      }
      a = wave_allsum_p(a); b = wave_allsum_p(b); c = wave_allsum_p(c);
      double cs = 1.0, sn = 0.0;
      if (jacobi_rotation(a, b, c, tol, small2, &cs, &sn)) {
        ++rotations;
        if (sn != 0.0) {
          double* vp = Vs + p * LDW;
          double* vq = Vs + q * LDW;
#pragma unroll
          for (int j = 0; j < KPT; ++j) {
            const int k = lane + 64 * j;
            wp[k] = cs * x[j] - sn * y[j];
            wq[k] = sn * x[j] + cs * y[j];
            const double vx = vp[k], vy = vq[k];
            vp[k] = cs * vx - sn * vy;
            vq[k] = sn * vx + cs * vy;
          }
        }
      }
    }
    __syncthreads();
  }
  for (int r = wave; r < 2 * B; r += B) {
    const int gi = grow(r);
    if (gi < m) {
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = lane + 64 * j;
        if (k < m) {
          W[(int64_t)gi * m + k] = Ws[r * LDW + k];
          V[(int64_t)gi * m + k] = Vs[r * LDW + k];
        }
      }
    }
  }
  if (lane == 0 && rotations != 0) atomicAdd(&rot_wg, rotations);
  __syncthreads();
  if (threadIdx.x == 0) rot_slots[blockIdx.x] += rot_wg;
}
constexpr int JB_LDS_BYTES = 2 * 2 * 4096 * 8;  // W and V rows of a block pair: 2 x 2B x (64 KPT) doubles with B KPT = 64
__device__ __forceinline__ void jacobi_block8_kernel_body(double* __restrict__ W, double* __restrict__ V, int m, int NB, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots) { jacobi_block_body<8, 8>(W, V, m, NB, round, tol, d_small2, rot_slots); }
__device__ __forceinline__ void jacobi_block4_kernel_body(double* __restrict__ W, double* __restrict__ V, int m, int NB, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots) { jacobi_block_body<4, 16>(W, V, m, NB, round, tol, d_small2, rot_slots); }
__device__ __forceinline__ void jacobi_block2_kernel_body(double* __restrict__ W, double* __restrict__ V, int m, int NB, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots) { jacobi_block_body<2, 32>(W, V, m, NB, round, tol, d_small2, rot_slots); }
__global__ void __launch_bounds__(512) jacobi_block8_kernel(double* __restrict__ W, double* __restrict__ V, int m, int NB, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots) { jacobi_block8_kernel_body(W, V, m, NB, round, tol, d_small2, rot_slots); }
__global__ void __launch_bounds__(256) jacobi_block4_kernel(double* __restrict__ W, double* __restrict__ V, int m, int NB, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots) { jacobi_block4_kernel_body(W, V, m, NB, round, tol, d_small2, rot_slots); }
__global__ void __launch_bounds__(128) jacobi_block2_kernel(double* __restrict__ W, double* __restrict__ V, int m, int NB, int round, double tol, const double* __restrict__ d_small2, int* __restrict__ rot_slots) { jacobi_block2_kernel_body(W, V, m, NB, round, tol, d_small2, rot_slots); }
NK_BATCHED_TWIN(jacobi_block8_kernel, (512), double*, double*, int, int, int, double, const double*, int*)
NK_BATCHED_TWIN(jacobi_block4_kernel, (256), double*, double*, int, int, int, double, const double*, int*)
NK_BATCHED_TWIN(jacobi_block2_kernel, (128), double*, double*, int, int, int, double, const double*, int*)
static bool g_jb_attr_set = false;
static int jacobi_block_attrs() {
  if (g_jb_attr_set) return NK_OK;
  const void* fns[6] = {reinterpret_cast<const void*>(jacobi_block8_kernel), reinterpret_cast<const void*>(jacobi_block4_kernel),
                        reinterpret_cast<const void*>(jacobi_block2_kernel), reinterpret_cast<const void*>(jacobi_block8_kernel_batched),
                        reinterpret_cast<const void*>(jacobi_block4_kernel_batched), reinterpret_cast<const void*>(jacobi_block2_kernel_batched)};
  for (const void* f : fns) NK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, JB_LDS_BYTES));
  g_jb_attr_set = true;
  return NK_OK;
}

// total[0] = sum of the per-pair rotation counters of a sweep; the counters are cleared for the next sweep.  One workgroup.
__device__ __forceinline__ void rot_total_kernel_body(int* __restrict__ slots, int n, int* __restrict__ total) {
  __shared__ int part[4];
  int s = 0;
  for (int j = threadIdx.x; j < n; j += 256) { s += slots[j]; slots[j] = 0; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) total[0] = part[0] + part[1] + part[2] + part[3];
}
__global__ void __launch_bounds__(256) rot_total_kernel(int* __restrict__ slots, int n, int* __restrict__ total) { rot_total_kernel_body(slots, n, total); }
NK_BATCHED_TWIN(rot_total_kernel, (256), int*, int, int*)

// sig2[j] = |row j of W|^2, one wave per row
__device__ __forceinline__ void row_sumsq_kernel_body(const double* __restrict__ W, int m, double* __restrict__ sig2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  double s = 0.0;
  for (int k = threadIdx.x & 63; k < m; k += 64) {
    const double x = W[(int64_t)row * m + k];
    s = fma(x, x, s);
  }
  s = wave_sum_p(s);
  if ((threadIdx.x & 63) == 0) sig2[row] = s;
}
__global__ void __launch_bounds__(256) row_sumsq_kernel(const double* __restrict__ W, int m, double* __restrict__ sig2) { row_sumsq_kernel_body(W, m, sig2); }
NK_BATCHED_TWIN(row_sumsq_kernel, (256), const double*, int, double*)

// d_small2[0] = (rcond * max_j |row j of W|)^2: the running estimate of the cut-off (max column norm <= sigma_max, and it
// converges to sigma_max as the columns become orthogonal).  One workgroup.
__device__ __forceinline__ void dead_threshold_kernel_body(const double* __restrict__ sig2, int m, double rcond, double* __restrict__ d_small2) {
  __shared__ double shmax[4];
  double mx = 0.0;
  for (int j = threadIdx.x; j < m; j += 256) mx = fmax(mx, sig2[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0) shmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) d_small2[0] = rcond * rcond * fmax(fmax(shmax[0], shmax[1]), fmax(shmax[2], shmax[3]));
}
__global__ void __launch_bounds__(256) dead_threshold_kernel(const double* __restrict__ sig2, int m, double rcond, double* __restrict__ d_small2) { dead_threshold_kernel_body(sig2, m, rcond, d_small2); }
NK_BATCHED_TWIN(dead_threshold_kernel, (256), const double*, int, double, double*)

// scale[j] = 1 / sigma_j^2 for the singular values that are kept, else 0; out[0] = rank, out[1] = sigma_max,
// out[2] = smallest retained sigma, out[3] = smallest sigma.  One workgroup.
// Cut-off: rcond * sigma_max (gelsd's rule).  One refinement: singular values up to window * sigma_max
// (window = 8 m eps) are within the rounding noise of the decomposition itself; if ALL singular values in that window
// form a cluster that a factor >= 1000 separates from the rest of the spectrum, the cluster is the image of an exact
// null space (its members would otherwise land on either side of the cut-off by chance -- and a kept one multiplies the
// right-hand side by 1e15) and is dropped as a whole.  A spectrum that decays continuously through the window, as those
// of the ill-conditioned kernel systems do, has no such gap and gets gelsd's rule unchanged.
__device__ __forceinline__ void pinv_scale_kernel_body(const double* __restrict__ sig2, int m, double rcond, double window, double* __restrict__ scale, double* __restrict__ out) {
  __shared__ double sh[4];
  __shared__ int shcnt[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  auto block_max = [&](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  };
  auto block_min = [&](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
  };
  double mx = 0.0;
  for (int j = threadIdx.x; j < m; j += 256) mx = fmax(mx, sig2[j]);
  const double smax = sqrt(block_max(mx));
  const double wtop = window * smax;
  double hi = 1e300, cmax = 0.0;
  for (int j = threadIdx.x; j < m; j += 256) {
    const double s = sqrt(sig2[j]);
    if (s > wtop) hi = fmin(hi, s); else cmax = fmax(cmax, s);
  }
  hi = block_min(hi);
  cmax = block_max(cmax);
  double cut = rcond * smax;
  if (cmax > cut && hi < 1e300 && hi >= 1000.0 * cmax) cut = wtop;  // an isolated rounding-level cluster: drop it whole
  int cnt = 0;
  double mn = 1e300, keepmin = 1e300;
  for (int j = threadIdx.x; j < m; j += 256) {
    const double s = sqrt(sig2[j]);
    mn = fmin(mn, s);
    if (s > cut) {
      scale[j] = 1.0 / sig2[j];
      keepmin = fmin(keepmin, s);
      ++cnt;
    } else {
      scale[j] = 0.0;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  __syncthreads();
  if (lane == 0) shcnt[w] = cnt;
  mn = block_min(mn);
  keepmin = block_min(keepmin);
  if (threadIdx.x == 0) {
    out[0] = (double)(shcnt[0] + shcnt[1] + shcnt[2] + shcnt[3]);
    out[1] = smax;
    out[2] = keepmin;
    out[3] = mn;
  }
}
__global__ void __launch_bounds__(256) pinv_scale_kernel(const double* __restrict__ sig2, int m, double rcond, double window, double* __restrict__ scale, double* __restrict__ out) { pinv_scale_kernel_body(sig2, m, rcond, window, scale, out); }
NK_BATCHED_TWIN(pinv_scale_kernel, (256), const double*, int, double, double, double*, double*)

__device__ __forceinline__ void scale_cols_kernel_body(double* __restrict__ T, int64_t ldt, int rows, int cols, const double* __restrict__ scale) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e - r * cols;
    T[r * ldt + c] *= scale[c];
  }
}
__global__ void __launch_bounds__(256) scale_cols_kernel(double* __restrict__ T, int64_t ldt, int rows, int cols, const double* __restrict__ scale) { scale_cols_kernel_body(T, ldt, rows, cols, scale); }
NK_BATCHED_TWIN(scale_cols_kernel, (256), double*, int64_t, int, int, const double*)

__device__ __forceinline__ void sumsq_all_kernel_body(const double* __restrict__ P, int64_t ldp, int m, double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  const int64_t total = (int64_t)m * m;
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const int64_t r = e / m, c = e - r * m;
    const double v = P[r * ldp + c];
    s = fma(v, v, s);
  }
  s = wave_sum_p(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) sumsq_all_kernel(const double* __restrict__ P, int64_t ldp, int m, double* __restrict__ out) { sumsq_all_kernel_body(P, ldp, m, out); }
NK_BATCHED_TWIN(sumsq_all_kernel, (256), const double*, int64_t, int, double*)

// E_out (rows x m) = E (rows x m) * pinv(P) with gelsd's singular-value cut-off `rcond` (relative to the largest one).
// P: symmetric m x m (device).  Synchronises the current stream (one host round trip per sweep).
int pinv_right_divide(nk_ctx* ctx, const double* P, int64_t ldp, int m, const double* E, int64_t lde, int rows,
                      double* E_out, int64_t ldeo, double rcond, PinvInfo* info) {
  const ArenaMark mk = arena_mark(ctx);
  const size_t mm = (size_t)m * m;
  double *W = nullptr, *V = nullptr, *sig2 = nullptr, *scale = nullptr, *T = nullptr, *d_out = nullptr;
  int* d_rot = nullptr;
  NK_TRY(arena_alloc_t(ctx, mm, &W));
  NK_TRY(arena_alloc_t(ctx, mm, &V));
  NK_TRY(arena_alloc_t(ctx, (size_t)m, &sig2));
  NK_TRY(arena_alloc_t(ctx, (size_t)m, &scale));
  NK_TRY(arena_alloc_t(ctx, (size_t)(rows > 0 ? rows : 1) * m, &T));
  NK_TRY(arena_alloc_t(ctx, (size_t)8, &d_out));
  // d_rot[0] = rotations of the last sweep, d_rot[2 ...] = one counter per pair of a round (ints)
  NK_TRY(arena_alloc_t(ctx, (size_t)m + 8, &d_rot));
  NK_HIP(hipMemsetAsync(d_rot, 0, ((size_t)m + 8) * sizeof(int), ctx->stream));
  NK_TRY(launch_copy2d(ctx, P, ldp, W, m, m, m));
  NK_TRY(launch_fill(ctx, V, m, m, m, 0.0));
  NK_TRY(launch_add_diag(ctx, V, m, m, 1.0));
  hipLaunchKernelGGL(sumsq_all_kernel, dim3(1), dim3(256), 0, ctx->stream, P, ldp, m, d_out);
  NK_HIP(hipGetLastError());
  NK_HIP(hipMemcpyAsync(ctx->h_scalars, d_out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  const double fro = std::sqrt(ctx->h_scalars[0]);
  if (!std::isfinite(fro)) {
    set_error("pseudo-inverse fallback: the system matrix is not finite");
    arena_release(ctx, mk);
    return NK_ERR_NOT_SPD;
  }
  const double tol = 2.220446049250313e-16 * (double)(m > 64 ? m : 64);  // orthogonality target, above the rounding level
                                                                         // of an m-term dot product
  const int N = m + (m & 1);
  const int max_sweeps = 150;  // cloth systems: 20-35; a spectrum spread evenly over 17 decades at m = 1000: ~80
  const bool trace = getenv("NYSKOOP_PINV_TRACE") != nullptr;
  // columns this far below the cut-off take no further part (they are dropped, and leaving them unconverged perturbs P by
  // less than 1e-3 eps sigma_max); NYSKOOP_PINV_DEAD overrides the factor for experiments
  const double dead_rel = getenv("NYSKOOP_PINV_DEAD") ? atof(getenv("NYSKOOP_PINV_DEAD")) : 1e-3 * rcond;
  int sweeps = 0, last_rot = -1;
  // A sweep as ONE launch pays inside a lock-step group (several units per launch: 72 MB move per round, the launch is
  // bandwidth bound and the kernel boundaries of 505 launches per sweep add their L2 write-backs: 1.10 -> 0.74 s for the 16
  // rank-deficient units of the cloth grid); a single solve is faster with a launch per round (5 us per round against
  // 8.5 us with the software barrier: 140 against 235 ms at m = 506, tools/jacobi_bench.py).  NYSKOOP_PINV_SWEEP_LAUNCH=0/1
  // forces either.
  const char* sweep_env = getenv("NYSKOOP_PINV_SWEEP_LAUNCH");
  bool one_launch = sweep_env ? sweep_env[0] != '0' : ctx_recording(ctx);
  int* d_sync = d_rot + 2 + (N / 2 + 2);
  // block Jacobi (default for m <= 2048; NYSKOOP_PINV_BLOCK=0 selects the scalar rounds)
  int block_b = 0;
  if (!(getenv("NYSKOOP_PINV_BLOCK") && getenv("NYSKOOP_PINV_BLOCK")[0] == '0') && m >= 4)
    block_b = m <= 512 ? 8 : (m <= 1024 ? 4 : (m <= 2048 ? 2 : 0));
  int NB = 0;
  if (block_b > 0) {
    NB = (m + block_b - 1) / block_b;
    NB += NB & 1;
    NK_TRY(jacobi_block_attrs());
  }
  if (m > 1) {
    for (; sweeps < max_sweeps; ++sweeps) {
      // refresh the dead-column threshold from the current column norms
      hipLaunchKernelGGL(row_sumsq_kernel, dim3((m + 3) / 4), dim3(256), 0, ctx->stream, W, m, sig2);
      hipLaunchKernelGGL(dead_threshold_kernel, dim3(1), dim3(256), 0, ctx->stream, sig2, m, dead_rel, d_out + 4);
      if (block_b > 0) {
        for (int r = 0; r < NB - 1; ++r) {
          if (block_b == 8)
            hipLaunchKernelGGL(jacobi_block8_kernel, dim3(NB / 2), dim3(512), JB_LDS_BYTES, ctx->stream, W, V, m, NB, r, tol, d_out + 4, d_rot + 2);
          else if (block_b == 4)
            hipLaunchKernelGGL(jacobi_block4_kernel, dim3(NB / 2), dim3(256), JB_LDS_BYTES, ctx->stream, W, V, m, NB, r, tol, d_out + 4, d_rot + 2);
          else
            hipLaunchKernelGGL(jacobi_block2_kernel, dim3(NB / 2), dim3(128), JB_LDS_BYTES, ctx->stream, W, V, m, NB, r, tol, d_out + 4, d_rot + 2);
        }
      } else if (m <= 64 * JAC_KPT && one_launch) {
        NK_HIP(hipMemsetAsync(d_sync, 0, 2 * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(jacobi_sweep_kernel, dim3((N / 2 + 15) / 16), dim3(1024), 0, ctx->stream, W, V, m, N, tol, d_out + 4,
                           d_rot + 2, d_sync);
        NK_HIP(hipMemcpyAsync(ctx->h_info + 9, d_sync + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      } else if (m <= 64 * JAC_KPT) {
        for (int r = 0; r < N - 1; ++r)
          hipLaunchKernelGGL(jacobi_round_wave_kernel, dim3((N / 2 + 3) / 4), dim3(256), 0, ctx->stream, W, V, m, N, r, tol,
                             d_out + 4, d_rot + 2);
      } else {
        for (int r = 0; r < N - 1; ++r)
          hipLaunchKernelGGL(jacobi_round_kernel, dim3(N / 2), dim3(256), 0, ctx->stream, W, V, m, N, r, tol, d_out + 4,
                             d_rot + 2);
      }
      hipLaunchKernelGGL(rot_total_kernel, dim3(1), dim3(256), 0, ctx->stream, d_rot + 2, block_b > 0 ? NB / 2 : N / 2, d_rot);
      NK_HIP(hipGetLastError());
      NK_HIP(hipMemcpyAsync(ctx->h_info + 8, d_rot, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      NK_HIP(hipStreamSynchronize(ctx->stream));
      last_rot = ctx->h_info[8];
      if (trace) fprintf(stderr, "[nk pinv] m=%d sweep %d: %d rotations\n", m, sweeps, last_rot);
      if (block_b == 0 && one_launch && m <= 64 * JAC_KPT && ctx->h_info[9] != 0) {
        // a workgroup gave up waiting for the others (they were not all resident): finish with one launch per round
        ctx->h_info[9] = 0;
        one_launch = false;
        count_event(CNT_JACOBI_GIVEUP);
        if (trace) fprintf(stderr, "[nk pinv] single-launch sweep gave up waiting: one launch per round from here\n");
        continue;
      }
      if (last_rot == 0) { ++sweeps; break; }
    }
  }
  hipLaunchKernelGGL(row_sumsq_kernel, dim3((m + 3) / 4), dim3(256), 0, ctx->stream, W, m, sig2);
  hipLaunchKernelGGL(pinv_scale_kernel, dim3(1), dim3(256), 0, ctx->stream, sig2, m, rcond,
                     8.0 * (double)m * 2.220446049250313e-16, scale, d_out);
  NK_HIP(hipGetLastError());
  if (rows > 0) {
    NK_TRY(launch_gemm(ctx, false, true, rows, m, m, 1.0, E, lde, V, m, 0.0, T, m));  // T = E V   (V^T stored: rows v_j)
    hipLaunchKernelGGL(scale_cols_kernel, dim3(256), dim3(256), 0, ctx->stream, T, (int64_t)m, rows, m, scale);
    NK_HIP(hipGetLastError());
    NK_TRY(launch_gemm(ctx, false, false, rows, m, m, 1.0, T, m, W, m, 0.0, E_out, ldeo));  // (T diag) W, rows of W = u_j
  }
  NK_HIP(hipMemcpyAsync(ctx->h_scalars, d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  if (info) {
    info->rank = (int)ctx->h_scalars[0];
    info->sigma_max = ctx->h_scalars[1];
    info->sigma_min_kept = ctx->h_scalars[2];
    info->sigma_min = ctx->h_scalars[3];
    info->sweeps = sweeps;
    info->converged = last_rot == 0 || m <= 1;
  }
  arena_release(ctx, mk);
  return NK_OK;
}

}  // namespace nk
