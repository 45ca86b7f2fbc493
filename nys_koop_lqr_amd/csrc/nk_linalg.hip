// Dense symmetric-positive-definite machinery of the O(m^3) stage, built on the fp64 MFMA GEMM engine:
//   * matrix square root / inverse square root through the polar factor of the Cholesky factor (scaled Newton-Schulz,
//     GEMM only; the coupled Newton-Schulz iteration is kept as the fallback for numerically singular input),
//     replacing scipy.linalg.sqrtm + solve(assume_a='her') (regressors.py:140,152,153,163,175,177);
//   * blocked Cholesky with the right-hand sides riding along as extra rows + a single-launch backward substitution
//     (nk_trsm.hip), replacing scipy.linalg.lstsq on the (numerically full-rank) regularised normal matrices
//     (regressors.py:155,165).  A non-positive pivot is reported as NK_ERR_NOT_SPD; there is no silent rank truncation.
#include "nk_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace nk {

// ---------------------------------------------------------------------------------------------------------------
// elementwise helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void add_diag_kernel_body(double* A, int64_t lda, int n, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[(int64_t)i * lda + i] += v;
}
__global__ void __launch_bounds__(256) add_diag_kernel(double* A, int64_t lda, int n, double v) { add_diag_kernel_body(A, lda, n, v); }
NK_BATCHED_TWIN(add_diag_kernel, (256), double*, int64_t, int, double)
__device__ __forceinline__ void copy2d_kernel_body(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e - r * cols;
    dst[r * ldd + c] = src[r * lds + c];
  }
}
__global__ void __launch_bounds__(256) copy2d_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd, int64_t rows, int64_t cols) { copy2d_kernel_body(src, lds, dst, ldd, rows, cols); }
NK_BATCHED_TWIN(copy2d_kernel, (256), const double*, int64_t, double*, int64_t, int64_t, int64_t)
__device__ __forceinline__ void axpby2d_kernel_body(double a, const double* __restrict__ X, int64_t ldx, double b, double* __restrict__ Y, int64_t ldy, int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e - r * cols;
    const double y = (b == 0.0) ? 0.0 : b * Y[r * ldy + c];
    Y[r * ldy + c] = a * X[r * ldx + c] + y;
  }
}
__global__ void __launch_bounds__(256) axpby2d_kernel(double a, const double* __restrict__ X, int64_t ldx, double b, double* __restrict__ Y, int64_t ldy, int64_t rows, int64_t cols) { axpby2d_kernel_body(a, X, ldx, b, Y, ldy, rows, cols); }
NK_BATCHED_TWIN(axpby2d_kernel, (256), double, const double*, int64_t, double, double*, int64_t, int64_t, int64_t)
__device__ __forceinline__ bool launch_skipped(const double* state, int step) {
  if (state == nullptr) return false;
  const double f = state[0];
  return f != 0.0 && f <= (double)step;
}
__device__ __forceinline__ void scale_add_identity_kernel_body(double a, const double* __restrict__ X, int64_t ldx, double c, double* __restrict__ Y, int64_t ldy, int n, const double* skip_state, int skip_step) {
  if (launch_skipped(skip_state, skip_step)) return;
  const int64_t total = (int64_t)n * n;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / n, col = e - r * n;
    Y[r * ldy + col] = a * X[r * ldx + col] + (r == col ? c : 0.0);
  }
}
__global__ void __launch_bounds__(256) scale_add_identity_kernel(double a, const double* __restrict__ X, int64_t ldx, double c, double* __restrict__ Y, int64_t ldy, int n, const double* skip_state, int skip_step) { scale_add_identity_kernel_body(a, X, ldx, c, Y, ldy, n, skip_state, skip_step); }
NK_BATCHED_TWIN(scale_add_identity_kernel, (256), double, const double*, int64_t, double, double*, int64_t, int, const double*, int)
__device__ __forceinline__ void fill_kernel_body(double* A, int64_t lda, int64_t rows, int64_t cols, double v) {
  const int64_t total = rows * cols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e - r * cols;
    A[r * lda + c] = v;
  }
}
__global__ void __launch_bounds__(256) fill_kernel(double* A, int64_t lda, int64_t rows, int64_t cols, double v) { fill_kernel_body(A, lda, rows, cols, v); }
NK_BATCHED_TWIN(fill_kernel, (256), double*, int64_t, int64_t, int64_t, double)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// per-block partial sums of (M - I)^2; finished by a second single-block pass (deterministic order)
__device__ __forceinline__ void frob_mi_partial_kernel_body(const double* __restrict__ M, int64_t ldm, int n, double* __restrict__ partial, const double* skip_state, int skip_step) {
  if (launch_skipped(skip_state, skip_step)) return;  // the stale partials give the old residual: harmless
  __shared__ double sh[4];
  const int64_t total = (int64_t)n * n;
  double s = 0.0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / n, c = e - r * n;
    const double v = M[r * ldm + c] - (r == c ? 1.0 : 0.0);
    s = fma(v, v, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) frob_mi_partial_kernel(const double* __restrict__ M, int64_t ldm, int n, double* __restrict__ partial, const double* skip_state, int skip_step) { frob_mi_partial_kernel_body(M, ldm, n, partial, skip_state, skip_step); }
NK_BATCHED_TWIN(frob_mi_partial_kernel, (256), const double*, int64_t, int, double*, const double*, int)
// per-block partial [sum of squares, trace]; finished by sum_partials_kernel on each half
__device__ __forceinline__ void sumsq_trace_partial_kernel_body(const double* __restrict__ M, int64_t ldm, int n, double* __restrict__ partial, int nblocks) {
  __shared__ double sh[8];
  const int64_t total = (int64_t)n * n;
  double s = 0.0, t = 0.0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / n, c = e - r * n;
    const double v = M[r * ldm + c];
    s = fma(v, v, s);
    if (r == c) t += v;
  }
  s = wave_sum(s);
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; sh[4 + (threadIdx.x >> 6)] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    partial[nblocks + blockIdx.x] = sh[4] + sh[5] + sh[6] + sh[7];
  }
}
__global__ void __launch_bounds__(256) sumsq_trace_partial_kernel(const double* __restrict__ M, int64_t ldm, int n, double* __restrict__ partial, int nblocks) { sumsq_trace_partial_kernel_body(M, ldm, n, partial, nblocks); }
NK_BATCHED_TWIN(sumsq_trace_partial_kernel, (256), const double*, int64_t, int, double*, int)
__device__ __forceinline__ void sum_partials_kernel_body(const double* __restrict__ partial, int count, double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += blockDim.x) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) sum_partials_kernel(const double* __restrict__ partial, int count, double* __restrict__ out) { sum_partials_kernel_body(partial, count, out); }
NK_BATCHED_TWIN(sum_partials_kernel, (256), const double*, int, double*)
// one wave per row: |row| sums, then max over rows via a second pass
__device__ __forceinline__ void abs_rowsum_kernel_body(const double* __restrict__ M, int64_t ldm, int n, double* __restrict__ rowsum) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  double s = 0.0;
  for (int c = threadIdx.x & 63; c < n; c += 64) s += fabs(M[(int64_t)row * ldm + c]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) rowsum[row] = s;
}
__global__ void __launch_bounds__(256) abs_rowsum_kernel(const double* __restrict__ M, int64_t ldm, int n, double* __restrict__ rowsum) { abs_rowsum_kernel_body(M, ldm, n, rowsum); }
NK_BATCHED_TWIN(abs_rowsum_kernel, (256), const double*, int64_t, int, double*)
__device__ __forceinline__ void max_kernel_body(const double* __restrict__ v, int count, double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += blockDim.x) s = fmax(s, v[i]);
  s = wave_max(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__global__ void __launch_bounds__(256) max_kernel(const double* __restrict__ v, int count, double* __restrict__ out) { max_kernel_body(v, count, out); }
NK_BATCHED_TWIN(max_kernel, (256), const double*, int, double*)

// Column sums of squared differences (the RMSE scorer): each workgroup reduces a slab of rows for 64 columns with
// coalesced row reads; wavefront reduction across the 4 waves through LDS; per-slab partials are summed in order.
__device__ __forceinline__ void colsum_sqdiff_partial_kernel_body(const double* __restrict__ P, int64_t ldp, const double* __restrict__ Y, int64_t ldy, int64_t rows, int cols, int rows_per_block, double* __restrict__ partial) {
  __shared__ double sh[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  double s = 0.0;
  if (col < cols)
    for (int64_t r = r0 + w; r < r1; r += 4) {
      const double v = P[r * ldp + col] - Y[r * ldy + col];
      s = fma(v, v, s);
    }
  sh[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && col < cols)
    partial[(int64_t)blockIdx.y * cols + col] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
__global__ void __launch_bounds__(256) colsum_sqdiff_partial_kernel(const double* __restrict__ P, int64_t ldp, const double* __restrict__ Y, int64_t ldy, int64_t rows, int cols, int rows_per_block, double* __restrict__ partial) { colsum_sqdiff_partial_kernel_body(P, ldp, Y, ldy, rows, cols, rows_per_block, partial); }
NK_BATCHED_TWIN(colsum_sqdiff_partial_kernel, (256), const double*, int64_t, const double*, int64_t, int64_t, int, int, double*)
__device__ __forceinline__ void colsum_finish_kernel_body(const double* __restrict__ partial, int nslabs, int cols, double* __restrict__ colsum) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  double s = 0.0;
  for (int k = 0; k < nslabs; ++k) s += partial[(int64_t)k * cols + col];
  colsum[col] = s;
}
__global__ void __launch_bounds__(256) colsum_finish_kernel(const double* __restrict__ partial, int nslabs, int cols, double* __restrict__ colsum) { colsum_finish_kernel_body(partial, nslabs, cols, colsum); }
NK_BATCHED_TWIN(colsum_finish_kernel, (256), const double*, int, int, double*)

// Res (nr x mq) = R - X P with every dot product accumulated in DOUBLED precision (Ogita / Rump / Oishi's Dot2: the exact
// product by an fma, the exact sum by Knuth's TwoSum, the error terms summed aside): the residual of the refinement step of
// the regularised solves.  A residual formed in plain fp64 carries a rounding error of eps |X| |P| -- as large as the
// residual itself, and the "correction" solved from it moves the solution AWAY from the true one (measured, also with
// LAPACK's factor in NumPy: the cloth fixture's A goes from 4e-5 to 2e-3 off the reference); with the doubled-precision
// residual each step contracts the forward error by cond(P) x the factor's backward error.  10 flop per term on the vector
// ALU: 0.1 ms at m = 200, ~4 ms at m = 2000 -- paid only by fits whose pivots say they need it.
__device__ __forceinline__ void resid_dd_kernel_body(const double* __restrict__ X, int64_t ldx, const double* __restrict__ P,
                                                     int64_t ldp, const double* __restrict__ R, int64_t ldr,
                                                     double* __restrict__ Res, int64_t ldres, int nr, int mq) {
#pragma clang fp contract(off)  // the error-free transformations below need the product and the sum rounded separately
  constexpr int T = 64, BK = 16;
  __shared__ double Xs[T][BK + 1];
  __shared__ double Ps[BK][T + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.y * T, j0 = blockIdx.x * T;
  double s[4][4], e[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      s[a][b] = (i < nr && j < mq) ? R[(int64_t)i * ldr + j] : 0.0;
      e[a][b] = 0.0;
    }
  for (int k0 = 0; k0 < mq; k0 += BK) {
#pragma unroll
    for (int t = tid; t < T * BK; t += 256) {
      const int ii = t / BK, kk = t % BK;   // X tile: k fastest (rows of X are contiguous in k)
      const int i = i0 + ii, k = k0 + kk;
      Xs[ii][kk] = (i < nr && k < mq) ? -X[(int64_t)i * ldx + k] : 0.0;
      const int kq = t / T, jj = t % T;     // P tile: j fastest
      const int kp = k0 + kq, j = j0 + jj;
      Ps[kq][jj] = (kp < mq && j < mq) ? P[(int64_t)kp * ldp + j] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < BK; ++kk) {
      double xv[4], pv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xv[a] = Xs[ty + 16 * a][kk];
#pragma unroll
      for (int b = 0; b < 4; ++b) pv[b] = Ps[kk][tx + 16 * b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double h = xv[a] * pv[b];
          const double r = __builtin_fma(xv[a], pv[b], -h);  // x y = h + r exactly
          const double t = s[a][b] + h;                       // s + h = t + q exactly (TwoSum)
          const double z = t - s[a][b];
          const double q = (s[a][b] - (t - z)) + (h - z);
          s[a][b] = t;
          e[a][b] += q + r;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      if (i < nr && j < mq) Res[(int64_t)i * ldres + j] = s[a][b] + e[a][b];
    }
}
__global__ void __launch_bounds__(256) resid_dd_kernel(const double* __restrict__ X, int64_t ldx, const double* __restrict__ P, int64_t ldp, const double* __restrict__ R, int64_t ldr, double* __restrict__ Res, int64_t ldres, int nr, int mq) { resid_dd_kernel_body(X, ldx, P, ldp, R, ldr, Res, ldres, nr, mq); }
NK_BATCHED_TWIN(resid_dd_kernel, (256), const double*, int64_t, const double*, int64_t, const double*, int64_t, double*, int64_t, int, int)

// Guard of the refinement steps: a step is applied only while the corrections contract.  Per-block partial sums of squares of
// the correction dX and of the solution X (rows x cols each) ...
__device__ __forceinline__ void refine_norms_partial_kernel_body(const double* __restrict__ dX, int64_t ldd, const double* __restrict__ X, int64_t ldx, int64_t rows, int64_t cols, double* __restrict__ partial, int nblocks) {
  __shared__ double sh[8];
  const int64_t total = rows * cols;
  double s = 0.0, t = 0.0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e - r * cols;
    const double v = dX[r * ldd + c], w = X[r * ldx + c];
    s = fma(v, v, s);
    t = fma(w, w, t);
  }
  s = wave_sum(s);
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; sh[4 + (threadIdx.x >> 6)] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    partial[nblocks + blockIdx.x] = sh[4] + sh[5] + sh[6] + sh[7];
  }
}
__global__ void __launch_bounds__(256) refine_norms_partial_kernel(const double* __restrict__ dX, int64_t ldd, const double* __restrict__ X, int64_t ldx, int64_t rows, int64_t cols, double* __restrict__ partial, int nblocks) { refine_norms_partial_kernel_body(dX, ldd, X, ldx, rows, cols, partial, nblocks); }
NK_BATCHED_TWIN(refine_norms_partial_kernel, (256), const double*, int64_t, const double*, int64_t, int64_t, int64_t, double*, int)
// ... and the verdict (one block).  state = [alive, |dX|^2 of the last accepted step, accepted steps, |dX_0| / |X|].  Step 0 is
// accepted when |dX_0| <= |X| / 4 -- the ratio estimates cond x (backward error of the factor), the contraction per step; a
// numerically singular system that happened to factor (gelsd would truncate it) gives >= 1 here and is left alone -- and a
// later step when its correction is at most half the previous one.  Once a step is rejected all later ones are.
__device__ __forceinline__ void refine_gate_kernel_body(const double* __restrict__ partial, int nblocks, int step, double* __restrict__ state) {
  __shared__ double sh[8];
  double s = 0.0, t = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { s += partial[i]; t += partial[nblocks + i]; }
  s = wave_sum(s);
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; sh[4 + (threadIdx.x >> 6)] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double dx2 = sh[0] + sh[1] + sh[2] + sh[3], x2 = sh[4] + sh[5] + sh[6] + sh[7];
    const bool alive = step == 0 ? true : state[0] != 0.0;
    const double ref2 = step == 0 ? 0.0625 * x2 : 0.25 * state[1];
    const bool ok = alive && dx2 <= ref2;  // false for NaN
    state[0] = ok ? 1.0 : 0.0;
    if (ok) state[1] = dx2;
    state[2] = (step == 0 ? 0.0 : state[2]) + (ok ? 1.0 : 0.0);
    if (step == 0) state[3] = x2 > 0.0 ? sqrt(dx2 / x2) : 0.0;
  }
}
__global__ void __launch_bounds__(256) refine_gate_kernel(const double* __restrict__ partial, int nblocks, int step, double* __restrict__ state) { refine_gate_kernel_body(partial, nblocks, step, state); }
NK_BATCHED_TWIN(refine_gate_kernel, (256), const double*, int, int, double*)
__device__ __forceinline__ void guarded_add_kernel_body(const double* __restrict__ dX, int64_t ldd, double* __restrict__ X, int64_t ldx, int64_t rows, int64_t cols, const double* __restrict__ state) {
  if (state[0] == 0.0) return;
  const int64_t total = rows * cols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e - r * cols;
    X[r * ldx + c] += dX[r * ldd + c];
  }
}
__global__ void __launch_bounds__(256) guarded_add_kernel(const double* __restrict__ dX, int64_t ldd, double* __restrict__ X, int64_t ldx, int64_t rows, int64_t cols, const double* __restrict__ state) { guarded_add_kernel_body(dX, ldd, X, ldx, rows, cols, state); }
NK_BATCHED_TWIN(guarded_add_kernel, (256), const double*, int64_t, double*, int64_t, int64_t, int64_t, const double*)

static inline int grid_for(int64_t total, int num_cu) {
  int64_t b = (total + 255) / 256;
  const int64_t cap = (int64_t)num_cu * 8;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int launch_add_diag(nk_ctx* ctx, double* A, int64_t lda, int n, double v) {
  if (n <= 0) return NK_OK;
  hipLaunchKernelGGL(add_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A, lda, n, v);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int launch_copy2d(nk_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return NK_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * cols, ctx->num_cu)), dim3(256), 0, ctx->stream, src, lds, dst,
                     ldd, rows, cols);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int launch_axpby2d(nk_ctx* ctx, double a, const double* X, int64_t ldx, double b, double* Y, int64_t ldy, int64_t rows,
                   int64_t cols) {
  if (rows <= 0 || cols <= 0) return NK_OK;
  hipLaunchKernelGGL(axpby2d_kernel, dim3(grid_for(rows * cols, ctx->num_cu)), dim3(256), 0, ctx->stream, a, X, ldx, b,
                     Y, ldy, rows, cols);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int launch_resid_dd(nk_ctx* ctx, const double* X, int64_t ldx, const double* P, int64_t ldp, const double* R, int64_t ldr,
                    double* Res, int64_t ldres, int nr, int mq) {
  if (nr <= 0 || mq <= 0) return NK_OK;
  hipLaunchKernelGGL(resid_dd_kernel, dim3((mq + 63) / 64, (nr + 63) / 64), dim3(256), 0, ctx->stream, X, ldx, P, ldp, R,
                     ldr, Res, ldres, nr, mq);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int launch_refine_apply(nk_ctx* ctx, const double* dX, int64_t ldd, double* X, int64_t ldx, int64_t rows, int64_t cols, int step,
                        double* state, double* partial /* 2 * refine_partial_blocks() doubles */) {
  if (rows <= 0 || cols <= 0) return NK_OK;
  const int blocks = std::min(grid_for(rows * cols, ctx->num_cu), refine_partial_blocks());
  hipLaunchKernelGGL(refine_norms_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dX, ldd, (const double*)X, ldx, rows,
                     cols, partial, blocks);
  hipLaunchKernelGGL(refine_gate_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)partial, blocks, step, state);
  hipLaunchKernelGGL(guarded_add_kernel, dim3(grid_for(rows * cols, ctx->num_cu)), dim3(256), 0, ctx->stream, dX, ldd, X, ldx,
                     rows, cols, (const double*)state);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int launch_scale_add_identity(nk_ctx* ctx, double a, const double* X, int64_t ldx, double c, double* Y, int64_t ldy,
                              int n, const TnSkip* skip) {
  if (n <= 0) return NK_OK;
  hipLaunchKernelGGL(scale_add_identity_kernel, dim3(grid_for((int64_t)n * n, ctx->num_cu)), dim3(256), 0, ctx->stream,
                     a, X, ldx, c, Y, ldy, n, skip ? skip->state : nullptr, skip ? skip->step : 0);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int launch_fill(nk_ctx* ctx, double* A, int64_t lda, int64_t rows, int64_t cols, double v) {
  if (rows <= 0 || cols <= 0) return NK_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(rows * cols, ctx->num_cu)), dim3(256), 0, ctx->stream, A, lda, rows,
                     cols, v);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
int launch_frob_minus_identity(nk_ctx* ctx, const double* M, int64_t ldm, int n, double* d_out, const TnSkip* skip) {
  const ArenaMark mk = arena_mark(ctx);
  const int blocks = grid_for((int64_t)n * n, ctx->num_cu);
  double* partial = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)blocks, &partial));
  hipLaunchKernelGGL(frob_mi_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, M, ldm, n, partial,
                     skip ? skip->state : nullptr, skip ? skip->step : 0);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, blocks, d_out);
  NK_HIP(hipGetLastError());
  arena_release(ctx, mk);
  return NK_OK;
}
int launch_max_abs_rowsum(nk_ctx* ctx, const double* M, int64_t ldm, int n, double* d_out) {
  const ArenaMark mk = arena_mark(ctx);
  double* rs = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)n, &rs));
  hipLaunchKernelGGL(abs_rowsum_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, M, ldm, n, rs);
  hipLaunchKernelGGL(max_kernel, dim3(1), dim3(256), 0, ctx->stream, rs, n, d_out);
  NK_HIP(hipGetLastError());
  arena_release(ctx, mk);
  return NK_OK;
}
int launch_colsum_sqdiff(nk_ctx* ctx, const double* P, int64_t ldp, const double* Y, int64_t ldy, int64_t rows,
                         int cols, double* d_colsum) {
  const ArenaMark mk = arena_mark(ctx);
  const int rpb = 256;
  const int nslabs = (int)((rows + rpb - 1) / rpb);
  double* partial = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)nslabs * cols, &partial));
  hipLaunchKernelGGL(colsum_sqdiff_partial_kernel, dim3((cols + 63) / 64, nslabs), dim3(256), 0, ctx->stream, P, ldp, Y,
                     ldy, rows, cols, rpb, partial);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, ctx->stream, partial, nslabs, cols,
                     d_colsum);
  NK_HIP(hipGetLastError());
  arena_release(ctx, mk);
  return NK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// matrix square root: coupled Newton-Schulz  Y <- Y T, Z <- T Z, T = (3I - ZY)/2,  Y0 = P/c, Z0 = I
//   Y -> (P/c)^{1/2}, Z -> (P/c)^{-1/2}.  Only M = ZY is symmetrised (mirrored upper tiles); symmetrising Y and Z as
//   well was observed to destabilise the iteration, the plain products are stable.
// ---------------------------------------------------------------------------------------------------------------
static int read_scalar(nk_ctx* ctx, const double* d_ptr, double* out) {
  NK_HIP(hipMemcpyAsync(ctx->h_scalars, d_ptr, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  *out = ctx->h_scalars[0];
  return NK_OK;
}

int sqrtm_spd_coupled(nk_ctx* ctx, const double* P, int64_t ldp, int m, double* S, double* Sinv, int* iters,
                      double* resid) {
  const ArenaMark mk = arena_mark(ctx);
  const size_t mm = (size_t)m * m;
  double *Y = nullptr, *Z = nullptr, *Yn = nullptr, *Zn = nullptr, *M = nullptr, *T = nullptr;
  double *Yt = nullptr, *Zt = nullptr, *Ytn = nullptr, *Ztn = nullptr;  // transposes, written by the GEMM epilogues
  NK_TRY(arena_alloc_t(ctx, mm, &Y));
  NK_TRY(arena_alloc_t(ctx, mm, &Z));
  NK_TRY(arena_alloc_t(ctx, mm, &Yn));
  NK_TRY(arena_alloc_t(ctx, mm, &Zn));
  NK_TRY(arena_alloc_t(ctx, mm, &M));
  NK_TRY(arena_alloc_t(ctx, mm, &T));
  NK_TRY(arena_alloc_t(ctx, mm, &Yt));
  NK_TRY(arena_alloc_t(ctx, mm, &Zt));
  NK_TRY(arena_alloc_t(ctx, mm, &Ytn));
  NK_TRY(arena_alloc_t(ctx, mm, &Ztn));
  double c = 0.0;
  NK_TRY(launch_max_abs_rowsum(ctx, P, ldp, m, ctx->d_scalars));
  NK_TRY(read_scalar(ctx, ctx->d_scalars, &c));
  if (!(c > 0.0) || !std::isfinite(c)) {
    set_error("sqrtm: matrix norm is %g", c);
    arena_release(ctx, mk);
    return NK_ERR_NOT_SPD;
  }
  NK_TRY(launch_axpby2d(ctx, 1.0 / c, P, ldp, 0.0, Y, m, m, m));
  NK_TRY(launch_fill(ctx, Z, m, m, m, 0.0));
  NK_TRY(launch_add_diag(ctx, Z, m, m, 1.0));
  NK_TRY(launch_transpose(ctx, Y, m, Yt, m, m, m));  // Y_0 = P / c (P symmetric only up to the caller's rounding)
  NK_TRY(launch_copy2d(ctx, Z, m, Zt, m, m, m));     // Z_0 = I
  // spectrum interval [a, b] of M_0 = Y_0: b = 1 (c = ||P||_inf bounds the largest eigenvalue); a = mean of the
  // eigenvalues other than the dominant one, from trace and Frobenius norm -- an OVER-estimate of the smallest
  // eigenvalue, which is the safe side: the scaled steps stay inside (0, 3) for every eigenvalue <= b, eigenvalues
  // below a still grow by the same factor, and the scaling fades to 1 as a -> 1 (plain Newton-Schulz finish).
  double a_lo = 1.0, b_hi = 1.0;
  {
    const int blocks = grid_for((int64_t)m * m, ctx->num_cu);
    double* partial = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)2 * blocks, &partial));
    hipLaunchKernelGGL(sumsq_trace_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, Y, (int64_t)m, m, partial,
                       blocks);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, blocks, ctx->d_scalars + 1);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, partial + blocks, blocks,
                       ctx->d_scalars + 2);
    NK_HIP(hipGetLastError());
    NK_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    const double fro = std::sqrt(ctx->h_scalars[1]), tr = ctx->h_scalars[2];
    const double lam1 = fro < 1.0 ? fro : 1.0;
    if (m > 1 && tr > lam1) a_lo = (tr - lam1) / (m - 1);
    else a_lo = tr / m * 1e-2;
    if (!(a_lo > 0.0) || !std::isfinite(a_lo)) a_lo = 1e-12;
    if (a_lo > 1.0) a_lo = 1.0;
  }
  auto p3 = [](double x) { return x * (3.0 - x) * (3.0 - x) * 0.25; };
  GemmOpts sym;
  sym.tri = TRI_UPPER_MIRROR;
  const int maxit = 100;
  double r = 1e300, r_prev = 1e300;
  int it = 0;
  bool ok = false;
  for (; it < maxit; ++it) {
    // every product is issued as P^T Q with P stored contraction-major (fast LDS-DMA engine); Z and Y are only
    // symmetric up to rounding and must NOT be replaced by their transposes (that variant diverges), so true
    // transposes are kept alongside (written by the epilogue of the launch that produces Y and Z)
    if (it == 0) {
      // Z_0 = I: M_0 = Y_0, taken as (Y_0 + Y_0^T)/2 so that T_0 is exactly symmetric (no GEMM)
      NK_TRY(launch_copy2d(ctx, Y, m, M, m, m, m));
      NK_TRY(launch_axpby2d(ctx, 0.5, Yt, m, 0.5, M, m, m, m));
    } else {
      NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, Zt, m, Y, m, 0.0, M, m, sym));
    }
    // a_lo over-estimates the smallest eigenvalue of M: while it is below 1/2 the iteration cannot have converged
    // (||M - I||_F / sqrt(m) >= (1 - lambda_min) / sqrt(m)), so the residual reduction and its host round trip are
    // skipped during the growth phase
    if (a_lo >= 0.5 || it + 1 == maxit) {
      NK_TRY(launch_frob_minus_identity(ctx, M, m, m, ctx->d_scalars));
      double r2 = 0.0;
      NK_TRY(read_scalar(ctx, ctx->d_scalars, &r2));
      r_prev = r;
      r = std::sqrt(r2 / m);
      if (!std::isfinite(r)) break;
      // quadratic convergence: once the previous residual was below 1e-7 this iterate sits on the rounding floor
      if (r < 5e-14 || r_prev < 1e-7) {
        ok = true;
        break;
      }
    }
    // scaled step: T = s (3I - s^2 M)/2 with s^2 = 3/(a + sqrt(ab) + b), which equalises p(s^2 a) = p(s^2 b) for
    // p(x) = x (3-x)^2 / 4, the map the step applies to the eigenvalues of M; s -> 1 as a -> b = 1
    const double s2 = 3.0 / (a_lo + std::sqrt(a_lo * b_hi) + b_hi);
    const double sc = std::sqrt(s2);
    {
      const double xa = s2 * a_lo, xb = s2 * b_hi;
      const double lo = p3(xa) < p3(xb) ? p3(xa) : p3(xb);
      b_hi = (xa <= 1.0 && xb >= 1.0) ? 1.0 : (p3(xa) > p3(xb) ? p3(xa) : p3(xb));
      a_lo = lo < b_hi ? lo : b_hi;
    }
    NK_TRY(launch_scale_add_identity(ctx, -0.5 * s2 * sc, M, m, 1.5 * sc, T, m, m));
    {
      // Y T and T Z (T is exactly symmetric) share K = m: one fused launch, 2 x 256 tiles = two workgroups per CU
      TnProblem pr[2];
      pr[0].A = Yt; pr[0].B = T; pr[0].C = Yn; pr[0].lda = pr[0].ldb = pr[0].ldc = m; pr[0].M = pr[0].N = m;
      pr[0].Ct = Ytn; pr[0].ldct = m;
      pr[1].A = T; pr[1].B = Z; pr[1].C = Zn; pr[1].lda = pr[1].ldb = pr[1].ldc = m; pr[1].M = pr[1].N = m;
      pr[1].Ct = Ztn; pr[1].ldct = m;
      if (it == 0) {
        // Z_0 = I: Z_1 = T_0 (symmetric), only Y_0 T_0 needs a GEMM
        if (tn_fast_ok(pr[0]) && m >= 128) {
          NK_TRY(launch_gemm_tn_multi(ctx, pr, 1, m, 0));
        } else {
          NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, Yt, m, T, m, 0.0, Yn, m));
          NK_TRY(launch_transpose(ctx, Yn, m, Ytn, m, m, m));
        }
        NK_TRY(launch_copy2d(ctx, T, m, Zn, m, m, m));
        NK_TRY(launch_copy2d(ctx, T, m, Ztn, m, m, m));
      } else if (tn_fast_ok(pr[0]) && tn_fast_ok(pr[1]) && m >= 128) {
        NK_TRY(launch_gemm_tn_multi(ctx, pr, 2, m, 0));
      } else {
        NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, Yt, m, T, m, 0.0, Yn, m));
        NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, T, m, Z, m, 0.0, Zn, m));
        NK_TRY(launch_transpose(ctx, Yn, m, Ytn, m, m, m));
        NK_TRY(launch_transpose(ctx, Zn, m, Ztn, m, m, m));
      }
    }
    double* t = Y; Y = Yn; Yn = t;
    t = Z; Z = Zn; Zn = t;
    t = Yt; Yt = Ytn; Ytn = t;
    t = Zt; Zt = Ztn; Ztn = t;
  }
  if (iters) *iters = it;
  if (resid) *resid = r;
  NK_TRY(x_align());
  if (!ok) {
    set_error("sqrtm: Newton-Schulz did not converge (residual %g after %d iterations)", r, it);
    arena_release(ctx, mk);
    return NK_ERR_NO_CONVERGENCE;
  }
  const double sc = std::sqrt(c);
  NK_TRY(launch_axpby2d(ctx, sc, Y, m, 0.0, S, m, m, m));
  NK_TRY(launch_axpby2d(ctx, 1.0 / sc, Z, m, 0.0, Sinv, m, m, m));
  arena_release(ctx, mk);
  return NK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// blocked Cholesky (lower) with inverted diagonal blocks
// ---------------------------------------------------------------------------------------------------------------
// potrf_diag_kernel lives in nk_potrf.hip (fully unrolled, slow to compile)
int launch_potrf_diag_pair(nk_ctx* ctx, double* const* Ajj, const int64_t* lda, const int* nb, double* const* Linv,
                           int nsys, int blk, double* const* plog);

// failure flags and [min, max] pivot slots of the current stream's two systems: flags <- 0, min <- +inf, max <- 0
__device__ __forceinline__ void reset_pivots_kernel_body(int* info, unsigned long long* piv) {
  if (threadIdx.x < 4) piv[threadIdx.x] = (threadIdx.x & 1) ? 0ull : 0x7FF0000000000000ull;
  if (threadIdx.x < 2) info[threadIdx.x] = 0;
}
__global__ void __launch_bounds__(256) reset_pivots_kernel(int* info, unsigned long long* piv) { reset_pivots_kernel_body(info, piv); }
NK_BATCHED_TWIN(reset_pivots_kernel, (256), int*, unsigned long long*)
static int reset_pivots(nk_ctx* ctx) {
  hipLaunchKernelGGL(reset_pivots_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_info + info_base(ctx),
                     ctx->d_piv + 2 * info_base(ctx));
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// order, workspace and pivot log of diagonal block jb + 1 of every system (order 0: the system has no such block)
static void next_block(const CholSys* sys, int nsys, int jb, int* nbn, double** Lin, double** Pln) {
  constexpr int NB = CHOL_NB;
  const int j1 = (jb + 1) * NB;
  for (int q = 0; q < nsys; ++q) {
    const CholSys& y = sys[q];
    if (j1 >= y.m) continue;
    nbn[q] = y.m - j1 < NB ? y.m - j1 : NB;
    Lin[q] = y.Linv + (size_t)(jb + 1) * CHOL_WS;
    Pln[q] = y.pivlog ? y.pivlog + j1 : nullptr;
  }
}

int cholesky_lower_pair_async(nk_ctx* ctx, const CholSys* sys, int nsys) {
  constexpr int NB = CHOL_NB;
  NK_REQUIRE(nsys >= 1 && nsys <= 2, "cholesky_lower_pair: 1..2 systems");
  NK_TRY(reset_pivots(ctx));
  int nblk = 0;
  for (int q = 0; q < nsys; ++q) nblk = std::max(nblk, (sys[q].m + NB - 1) / NB);
  const bool fuse = chol_fuse_enabled();
  bool diag_done = false;  // the diagonal block of this step was factored by the previous step's fused launch
  for (int jb = 0; jb < nblk; ++jb) {
    const int j0 = jb * NB;
    double* Ajj[2] = {nullptr, nullptr};
    double* Li[2] = {nullptr, nullptr};
    double* Pl[2] = {nullptr, nullptr};
    int64_t lda[2] = {0, 0};
    int nbj[2] = {0, 0};
    GemmCall panel[2], trail[2];
    for (int q = 0; q < nsys; ++q) {
      const CholSys& y = sys[q];
      if (j0 >= y.m) continue;
      nbj[q] = y.m - j0 < NB ? y.m - j0 : NB;
      Ajj[q] = y.P + (int64_t)j0 * y.ldp + j0;
      lda[q] = y.ldp;
      Li[q] = y.Linv + (size_t)jb * CHOL_WS;
      Pl[q] = y.pivlog ? y.pivlog + j0 : nullptr;
      const int rem = y.m - j0 - nbj[q];
      if (rem > 0) {
        double* pnl = y.P + (int64_t)(j0 + nbj[q]) * y.ldp + j0;
        // panel <- panel * Linv_jj^T   (in place: one n-tile, every workgroup reads exactly the rows it writes)
        panel[q].M = rem; panel[q].N = nbj[q]; panel[q].K = nbj[q];
        panel[q].A = pnl; panel[q].lda = y.ldp; panel[q].B = Li[q]; panel[q].ldb = NB;
        panel[q].C = pnl; panel[q].ldc = y.ldp;
        // trailing <- trailing - panel * panel^T  (lower tiles)
        trail[q].M = rem; trail[q].N = rem; trail[q].K = nbj[q]; trail[q].alpha = -1.0; trail[q].beta = 1.0;
        trail[q].A = pnl; trail[q].lda = y.ldp; trail[q].B = pnl; trail[q].ldb = y.ldp;
        trail[q].C = y.P + (int64_t)(j0 + nbj[q]) * y.ldp + (j0 + nbj[q]); trail[q].ldc = y.ldp;
        trail[q].opts.tri = TRI_LOWER;
      }
    }
    if (!diag_done) NK_TRY(launch_potrf_diag_pair(ctx, Ajj, lda, nbj, Li, nsys, jb, Pl));
    diag_done = false;
    {
      int rc_panel = NK_OK;  // 64 x 64 panel product: specialised kernel (nk_trail.hip), generic engine otherwise
      if (!launch_chol_panel_pair(ctx, panel, nsys, &rc_panel)) NK_TRY(launch_gemm_pair(ctx, false, true, panel, nsys));
      NK_TRY(rc_panel);
    }
    // trailing update -- in one launch with the factorisation of the next diagonal block where the shapes allow
    if (fuse && jb + 1 < nblk) {
      int rc_f = NK_OK;
      int nbn[2] = {0, 0};
      double* Lin[2] = {nullptr, nullptr};
      double* Pln[2] = {nullptr, nullptr};
      next_block(sys, nsys, jb, nbn, Lin, Pln);
      if (launch_chol_trail_potrf_pair(ctx, trail, nsys, nbn, Lin, jb + 1, Pln, &rc_f)) {
        NK_TRY(rc_f);
        diag_done = true;
        continue;
      }
    }
    {
      int rc_trail = NK_OK;  // K = 64 rank update: specialised kernel (nk_trail.hip), generic engine otherwise
      if (!launch_chol_trail_pair(ctx, trail, nsys, &rc_trail)) NK_TRY(launch_gemm_pair(ctx, false, true, trail, nsys));
      NK_TRY(rc_trail);
    }
  }
  return NK_OK;
}

// Host-side verdict of the factorisations queued by cholesky_lower_pair_async (synchronises the current stream).
int cholesky_check_pair(nk_ctx* ctx, const CholSys* sys, int nsys) {
  int failed[2] = {0, 0};
  NK_TRY(cholesky_fail_flags(ctx, sys, nsys, failed));
  for (int q = 0; q < nsys; ++q)
    if (failed[q] != 0) {
      if (failed[q] > 0)
        set_error("Cholesky: non-positive pivot at index %d of %d (system %d is numerically rank deficient; the "
                  "reference's lstsq would truncate here)", failed[q] - 1, sys[q].m, q);
      else
        set_error("Cholesky: system %d (order %d) has an isolated cluster of rounding-level pivots: an exact null space (the "
                  "reference's lstsq truncates it)", q, sys[q].m);
      return NK_ERR_NOT_SPD;
    }
  return NK_OK;
}

// Verdict of the (paired) factorisation queued last on the current stream: failed[q] > 0 when system q met a
// non-positive pivot (index + 1), -1 when its pivots show an EXACT null space: a cluster of pivots at the rounding level of
// the factorisation (<= 8 m eps d_max) that a factor >= 1000 separates from all other pivots -- duplicated landmarks,
// a rank-deficient Gram matrix -- where rounding merely happened to leave the pivots positive.  (Needs sys[q].pivlog; without it
// only d_min <= eps d_max counts.)  A spectrum that decays CONTINUOUSLY to that level -- the ill-conditioned kernel systems of
// a hyper-parameter grid: genuine pivots of 1e-13 d_max across the whole gamma = 1e-7 row of the cloth grid -- has no such
// gap and is solved at full rank: there the reference's own rank decision is rounding noise and the
// SVD path would cost 100 x more for an answer no closer to it.  Synchronises the current stream.
int cholesky_fail_flags(nk_ctx* ctx, const CholSys* sys, int nsys, int* failed, double* piv_ratio) {
  const int ib = info_base(ctx);
  std::vector<double> plog[2];
  NK_HIP(hipMemcpyAsync(ctx->h_info + ib, ctx->d_info + ib, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipMemcpyAsync(ctx->h_piv + 2 * ib, ctx->d_piv + 2 * ib, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                        ctx->stream));
  for (int q = 0; q < nsys; ++q)
    if (sys[q].pivlog) {
      plog[q].resize((size_t)sys[q].m);
      NK_HIP(hipMemcpyAsync(plog[q].data(), sys[q].pivlog, sizeof(double) * sys[q].m, hipMemcpyDeviceToHost, ctx->stream));
    }
  NK_HIP(hipStreamSynchronize(ctx->stream));
  const double eps = 2.220446049250313e-16;
  for (int q = 0; q < nsys; ++q) {
    failed[q] = ctx->h_info[ib + q];
    if (piv_ratio) piv_ratio[q] = 0.0;
    if (failed[q] != 0) continue;
    double dmin, dmax;
    memcpy(&dmin, &ctx->h_piv[2 * (ib + q)], 8);
    memcpy(&dmax, &ctx->h_piv[2 * (ib + q) + 1], 8);
    if (!(dmax > 0.0)) continue;
    if (piv_ratio) piv_ratio[q] = dmin / dmax;
    if (dmin <= eps * dmax) { failed[q] = -1; continue; }
    if (plog[q].empty()) continue;
    std::sort(plog[q].begin(), plog[q].end());
    const double window = 8.0 * (double)sys[q].m * eps * dmax;
    size_t k = 0;
    while (k < plog[q].size() && plog[q][k] <= window) ++k;
    if (k > 0 && k < plog[q].size() && plog[q][k] >= 1000.0 * plog[q][k - 1]) failed[q] = -1;
  }
  return NK_OK;
}

int cholesky_lower_pair(nk_ctx* ctx, const CholSys* sys, int nsys) {
  NK_TRY(cholesky_lower_pair_async(ctx, sys, nsys));
  return cholesky_check_pair(ctx, sys, nsys);
}

// Factorisation of [P; R^T] with the right-hand sides riding along as extra rows, then the backward substitution on
// those rows.  Per block step: potrf (both systems), panel, trailing; the backward pass is a single launch in which
// every workgroup carries a band of rows through the whole substitution.  The separate forward substitution
// (2 launches per block) disappears.
//
// LOOK-AHEAD.  The chain is latency bound: per block step a one-wave diagonal factorisation (~42 us), the panel product
// (~12 us) and the rank-64 trailing update (~33 us), each waiting for the one before.  Only the NEXT block column of the
// trailing matrix is needed to go on, so the update is split: the next 64 columns are updated on the chain's own stream
// (one tile column), the rest of the trailing matrix on a second stream of the same priority, beside the next step's
// diagonal block and panel.  Dependencies: rest(j) needs panel(j) [event P] and rest(j-1) [stream order]; the narrow update
// of step j needs rest(j-1) [event R] -- the column it touches received its older updates there.  Members of a lock-step
// group record everything into one sequence (events are no-ops there), which is a valid order of the same graph.
int cholesky_aug_pair_async(nk_ctx* ctx, const CholSys* sys, int nsys, hipEvent_t pause, int pause_step) {
  constexpr int NB = CHOL_NB;
  NK_REQUIRE(nsys >= 1 && nsys <= 2, "cholesky_aug_pair: 1..2 systems");
  NK_TRY(reset_pivots(ctx));
  int nblk = 0;
  for (int q = 0; q < nsys; ++q) nblk = std::max(nblk, (sys[q].m + NB - 1) / NB);
  // Measured inside the headline fit (bench.py, same box): with the look-ahead the fit is 1.4 ms SLOWER -- the chain shares
  // the chip with the square-root iteration's GEMMs, its kernels crawl for lack of issue slots rather than for lack of
  // parallelism, and a second stream of them takes more from the GEMMs than the shorter dependency chain gives back.
  // Off by default; NYSKOOP_CHOL_LOOKAHEAD=1 turns it on for chains on the main stream (a stand-alone nk_solve_spd on an
  // idle chip gains from it).
  static const bool la_env = getenv("NYSKOOP_CHOL_LOOKAHEAD") && getenv("NYSKOOP_CHOL_LOOKAHEAD")[0] == '1';
  // which look-ahead stream pairs with the current one (none for callers on other streams)
  const int la = ctx->stream == ctx->stream_main ? 1 : -1;
  const bool lookahead = la_env && la >= 0 && nblk >= 4 && !ctx_recording(ctx);
  hipStream_t s_chain = ctx->stream;
  hipStream_t s_rest = lookahead ? ctx->stream_la[la] : ctx->stream;
  hipEvent_t* ev = lookahead ? ctx->ev_la[la] : nullptr;  // [0..1] panel done (parity of the step), [2..3] rest done
  bool rest_pending = false;
  hipEvent_t last_rest = nullptr;
  if (lookahead) {  // the look-ahead stream starts behind whatever the chain's stream has queued so far
    NK_HIP(hipEventRecord(ev[1], s_chain));
    NK_HIP(hipStreamWaitEvent(s_rest, ev[1], 0));
  }
  const bool fuse = chol_fuse_enabled();
  bool diag_done = false;  // the diagonal block of this step was factored by the previous step's fused launch
  for (int jb = 0; jb < nblk; ++jb) {
    if (pause != nullptr && jb == pause_step) NK_HIP(hipStreamWaitEvent(ctx->stream, pause, 0));
    const int j0 = jb * NB;
    double* Ajj[2] = {nullptr, nullptr};
    double* Li[2] = {nullptr, nullptr};
    double* Pl[2] = {nullptr, nullptr};
    int64_t lda[2] = {0, 0};
    int nbj[2] = {0, 0};
    GemmCall panel[2], trail[2], next[2], rest[2];
    bool any_rest = false;
    for (int q = 0; q < nsys; ++q) {
      const CholSys& y = sys[q];
      if (j0 >= y.m) continue;
      nbj[q] = y.m - j0 < NB ? y.m - j0 : NB;
      Ajj[q] = y.P + (int64_t)j0 * y.ldp + j0;
      lda[q] = y.ldp;
      Li[q] = y.Linv + (size_t)jb * CHOL_WS;
      Pl[q] = y.pivlog ? y.pivlog + j0 : nullptr;
      const int rem = y.m - j0 - nbj[q];      // rows of the square part below the diagonal block
      const int rows = rem + y.extra;         // ... plus the right-hand-side rows
      if (rows > 0) {
        double* pnl = y.P + (int64_t)(j0 + nbj[q]) * y.ldp + j0;
        panel[q].M = rows; panel[q].N = nbj[q]; panel[q].K = nbj[q];
        panel[q].A = pnl; panel[q].lda = y.ldp; panel[q].B = Li[q]; panel[q].ldb = NB;
        panel[q].C = pnl; panel[q].ldc = y.ldp;
        if (rem > 0) {
          trail[q].M = rows; trail[q].N = rem; trail[q].K = nbj[q]; trail[q].alpha = -1.0; trail[q].beta = 1.0;
          trail[q].A = pnl; trail[q].lda = y.ldp; trail[q].B = pnl; trail[q].ldb = y.ldp;
          trail[q].C = y.P + (int64_t)(j0 + nbj[q]) * y.ldp + (j0 + nbj[q]); trail[q].ldc = y.ldp;
          trail[q].opts.tri = TRI_LOWER;  // lower tiles of the square part, full tiles for the extra rows
          // split for the look-ahead: the next block column (all rows) | everything to the right of it
          const int nbn = rem < NB ? rem : NB;
          next[q] = trail[q];
          next[q].N = nbn;
          if (rem > nbn) {
            rest[q] = trail[q];
            rest[q].M = rows - nbn; rest[q].N = rem - nbn;
            rest[q].A = rest[q].B = pnl + (int64_t)nbn * y.ldp;
            rest[q].C = trail[q].C + (int64_t)nbn * y.ldp + nbn;
            any_rest = true;
          }
        }
      }
    }
    if (!diag_done) NK_TRY(launch_potrf_diag_pair(ctx, Ajj, lda, nbj, Li, nsys, jb, Pl));
    diag_done = false;
    {
      int rc_panel = NK_OK;  // 64 x 64 panel product: specialised kernel (nk_trail.hip), generic engine otherwise
      if (!launch_chol_panel_pair(ctx, panel, nsys, &rc_panel)) NK_TRY(launch_gemm_pair(ctx, false, true, panel, nsys));
      NK_TRY(rc_panel);
    }
    // trailing update -- in one launch with the factorisation of the next diagonal block where the shapes allow
    if (fuse && !lookahead && jb + 1 < nblk) {
      int rc_f = NK_OK;
      int nbn[2] = {0, 0};
      double* Lin[2] = {nullptr, nullptr};
      double* Pln[2] = {nullptr, nullptr};
      next_block(sys, nsys, jb, nbn, Lin, Pln);
      if (launch_chol_trail_potrf_pair(ctx, trail, nsys, nbn, Lin, jb + 1, Pln, &rc_f)) {
        NK_TRY(rc_f);
        diag_done = true;
        continue;
      }
    }
    auto update = [&](const GemmCall* calls) -> int {  // K = 64 rank update: specialised kernel, generic engine otherwise
      int rc_trail = NK_OK;
      if (!launch_chol_trail_pair(ctx, calls, nsys, &rc_trail)) NK_TRY(launch_gemm_pair(ctx, false, true, calls, nsys));
      return rc_trail;
    };
    if (!lookahead) {
      NK_TRY(update(trail));
    } else {
      hipEvent_t evP = ev[jb & 1], evR = ev[2 + (jb & 1)], evR_prev = ev[2 + ((jb + 1) & 1)];
      NK_HIP(hipEventRecord(evP, s_chain));                               // panel(jb) is complete
      if (rest_pending) NK_HIP(hipStreamWaitEvent(s_chain, evR_prev, 0));  // the next column has its older updates
      NK_TRY(update(next));
      rest_pending = false;
      if (any_rest) {
        ctx->stream = s_rest;
        NK_HIP(hipStreamWaitEvent(s_rest, evP, 0));
        const int rc = update(rest);
        if (rc == NK_OK) NK_HIP(hipEventRecord(evR, s_rest));
        last_rest = evR;
        ctx->stream = s_chain;
        NK_TRY(rc);
        rest_pending = true;
      }
    }
  }
  if (lookahead && last_rest) NK_HIP(hipStreamWaitEvent(s_chain, last_rest, 0));  // (the chain also ends behind the last rest)
  // backward on the extra rows E (extra x m, now holding (L^-1 R)^T):  E <- E L^-1, one launch (nk_trsm.hip)
  NK_TRY(launch_trsm_right_lower_pair(ctx, sys, nsys));
  return NK_OK;
}

int cholesky_solve_pair(nk_ctx* ctx, const CholSys* sys, int nsys) {
  constexpr int NB = CHOL_NB;
  NK_REQUIRE(nsys >= 1 && nsys <= 2, "cholesky_solve_pair: 1..2 systems");
  int nblk = 0;
  for (int q = 0; q < nsys; ++q) nblk = std::max(nblk, (sys[q].m + NB - 1) / NB);
  const ArenaMark mk = arena_mark(ctx);
  double* tmp[2] = {nullptr, nullptr};
  for (int q = 0; q < nsys; ++q) NK_TRY(arena_alloc_t(ctx, (size_t)NB * sys[q].ldr, &tmp[q]));
  // One diagonal-block solve  R_j <- op(L_jj)^-1 R_j : the product with the explicitly inverted block, then one correction
  // step from the data (the product alone is not backward stable, see chol_panel_kernel):
  //   X = Linv R_j ;  X += Linv (R_j - L_jj X)
  auto diag_solve = [&](int jb, bool trans) -> int {
    const int j0 = jb * NB;
    GemmCall g1[2], g2[2], g3[2];
    for (int q = 0; q < nsys; ++q) {
      const CholSys& y = sys[q];
      if (j0 >= y.m) continue;
      const int nbj = y.m - j0 < NB ? y.m - j0 : NB;
      const double* Li = y.Linv + (size_t)jb * CHOL_WS;
      const double* Ld = Li + NB * NB;
      double* Rj = y.R + (int64_t)j0 * y.ldr;
      NK_TRY(launch_copy2d(ctx, Rj, y.ldr, tmp[q], y.ldr, nbj, y.nrhs));
      g1[q].M = nbj; g1[q].N = y.nrhs; g1[q].K = nbj; g1[q].A = Li; g1[q].lda = NB; g1[q].B = tmp[q]; g1[q].ldb = y.ldr;
      g1[q].C = Rj; g1[q].ldc = y.ldr;
      g2[q] = g1[q]; g2[q].A = Ld; g2[q].B = Rj; g2[q].C = tmp[q]; g2[q].alpha = -1.0; g2[q].beta = 1.0;
      g3[q] = g1[q]; g3[q].beta = 1.0;
    }
    NK_TRY(launch_gemm_pair(ctx, trans, false, g1, nsys));
    NK_TRY(launch_gemm_pair(ctx, trans, false, g2, nsys));
    NK_TRY(launch_gemm_pair(ctx, trans, false, g3, nsys));
    return NK_OK;
  };
  // forward: L T = R
  for (int jb = 0; jb < nblk; ++jb) {
    const int j0 = jb * NB;
    GemmCall upd[2];
    for (int q = 0; q < nsys; ++q) {
      const CholSys& y = sys[q];
      if (j0 >= y.m) continue;
      const int nbj = y.m - j0 < NB ? y.m - j0 : NB;
      double* Rj = y.R + (int64_t)j0 * y.ldr;
      const int rem = y.m - j0 - nbj;
      if (rem > 0) {
        upd[q].M = rem; upd[q].N = y.nrhs; upd[q].K = nbj; upd[q].alpha = -1.0; upd[q].beta = 1.0;
        upd[q].A = y.P + (int64_t)(j0 + nbj) * y.ldp + j0; upd[q].lda = y.ldp; upd[q].B = Rj; upd[q].ldb = y.ldr;
        upd[q].C = y.R + (int64_t)(j0 + nbj) * y.ldr; upd[q].ldc = y.ldr;
      }
    }
    NK_TRY(diag_solve(jb, false));
    NK_TRY(launch_gemm_pair(ctx, false, false, upd, nsys));
  }
  // backward: L^T X = T
  for (int jb = nblk - 1; jb >= 0; --jb) {
    const int j0 = jb * NB;
    GemmCall upd[2];
    for (int q = 0; q < nsys; ++q) {
      const CholSys& y = sys[q];
      if (j0 >= y.m) continue;
      const int nbj = y.m - j0 < NB ? y.m - j0 : NB;
      double* Rj = y.R + (int64_t)j0 * y.ldr;
      if (j0 > 0) {
        upd[q].M = j0; upd[q].N = y.nrhs; upd[q].K = nbj; upd[q].alpha = -1.0; upd[q].beta = 1.0;
        upd[q].A = y.P + (int64_t)j0 * y.ldp; upd[q].lda = y.ldp; upd[q].B = Rj; upd[q].ldb = y.ldr;
        upd[q].C = y.R; upd[q].ldc = y.ldr;
      }
    }
    NK_TRY(diag_solve(jb, true));
    NK_TRY(launch_gemm_pair(ctx, true, false, upd, nsys));
  }
  arena_release(ctx, mk);
  return NK_OK;
}

int cholesky_lower(nk_ctx* ctx, double* P, int64_t ldp, int m, double* Linv) {
  CholSys y;
  y.P = P; y.ldp = ldp; y.m = m; y.Linv = Linv;
  return cholesky_lower_pair(ctx, &y, 1);
}

int cholesky_solve(nk_ctx* ctx, const double* L, int64_t ldl, int m, const double* Linv, double* R, int64_t ldr,
                   int nrhs) {
  CholSys y;
  y.P = const_cast<double*>(L); y.ldp = ldl; y.m = m; y.Linv = const_cast<double*>(Linv); y.R = R; y.ldr = ldr;
  y.nrhs = nrhs;
  return cholesky_solve_pair(ctx, &y, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// matrix square root through the polar decomposition of the Cholesky factor:  P = L L^T,  L^T = Q H  with Q orthogonal
// and H = (L L^T)^{1/2} = P^{1/2}, hence  S = Q^T L^T  and  S^-1 = L^-T Q.   Q is the limit of the scaled Newton-Schulz
// iteration  X <- X T,  T = s (3 I - s^2 X^T X) / 2,  X_0 = L^T / sqrt(||P||_inf): the eigenvalues of M = X^T X follow
// exactly the map of the coupled iteration above (M_0 = P / c in both), so the step count is the same, but a step costs
// one symmetric product (half the tiles) and ONE full product instead of two -- 3 m^3 flop instead of 5 m^3 -- and the
// full product is a single 256-tile launch at m = 2000 that leaves every CU one workgroup slot for the factorisation
// chain running beside it.  The price is one more latency-bound blocked Cholesky (with the identity riding along as
// extra rows, which leaves L^-T), queued by sqrtm_prepare long before the iteration is needed.
// ---------------------------------------------------------------------------------------------------------------
// Xt = L * s on and below the diagonal, zero above (the factorisation leaves the old upper triangle in place);
// X = Xt^T;  s = 1 / sqrt(d_c[0])
__device__ __forceinline__ void tri_scale_both_kernel_body(const double* __restrict__ L, int64_t ldl, int m, const double* __restrict__ d_c, double* __restrict__ Xt, double* __restrict__ X) {
  __shared__ double tile[32][33];
  const double s = 1.0 / sqrt(d_c[0]);
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = by + r, j = bx + tx;
    double v = 0.0;
    if (i < m && j <= i) v = L[(int64_t)i * ldl + j] * s;
    if (i < m && j < m) Xt[(int64_t)i * m + j] = v;
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bx + r, j = by + tx;
    if (i < m && j < m) X[(int64_t)i * m + j] = tile[tx][r];
  }
}
__global__ void __launch_bounds__(256) tri_scale_both_kernel(const double* __restrict__ L, int64_t ldl, int m, const double* __restrict__ d_c, double* __restrict__ Xt, double* __restrict__ X) { tri_scale_both_kernel_body(L, ldl, m, d_c, Xt, X); }
NK_BATCHED_TWIN(tri_scale_both_kernel, (256), const double*, int64_t, int, const double*, double*, double*)

int sqrtm_prepare(nk_ctx* ctx, const double* P, int64_t ldp, int m, SqrtPlan* plan) {
  plan->P = P; plan->ldp = ldp; plan->m = m;
  plan->mark = arena_mark(ctx);
  const size_t mm = (size_t)m * m;
  const int nblk = (m + CHOL_NB - 1) / CHOL_NB;
  NK_TRY(arena_alloc_t(ctx, 2 * mm, &plan->W));
  NK_TRY(arena_alloc_t(ctx, (size_t)nblk * CHOL_WS, &plan->Linv));
  NK_TRY(arena_alloc_t(ctx, mm, &plan->X0));
  NK_TRY(arena_alloc_t(ctx, mm, &plan->X0t));
  NK_TRY(arena_alloc_t(ctx, (size_t)8, &plan->d_sc));
  NK_TRY(launch_max_abs_rowsum(ctx, P, ldp, m, plan->d_sc));
  {
    const ArenaMark mk = arena_mark(ctx);
    const int blocks = grid_for((int64_t)m * m, ctx->num_cu);
    double* partial = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)2 * blocks, &partial));
    hipLaunchKernelGGL(sumsq_trace_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P, ldp, m, partial, blocks);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, blocks, plan->d_sc + 1);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, partial + blocks, blocks, plan->d_sc + 2);
    NK_HIP(hipGetLastError());
    arena_release(ctx, mk);
  }
  // the three scalars of the scaling schedule are on the host long before the iteration is queued
  NK_HIP(hipMemcpyAsync(ctx->h_scalars + 12, plan->d_sc, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipEventRecord(ctx->ev[11], ctx->stream));
  double* E = plan->W + mm;
  NK_TRY(launch_copy2d(ctx, P, ldp, plan->W, m, m, m));
  NK_TRY(launch_fill(ctx, E, m, m, m, 0.0));
  NK_TRY(launch_add_diag(ctx, E, m, m, 1.0));
  CholSys y;
  y.P = plan->W; y.ldp = m; y.m = m; y.extra = m; y.backward = false; y.Linv = plan->Linv;
  NK_TRY(cholesky_aug_pair_async(ctx, &y, 1, plan->pause_event, plan->pause_step));  // W <- [L ; L^-T]
  {
    // ||L^-1||_F^2 (the extra rows hold L^-T): 1 / it bounds the smallest eigenvalue of P from below
    const ArenaMark mk = arena_mark(ctx);
    const int blocks = grid_for((int64_t)m * m, ctx->num_cu);
    double* partial = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)2 * blocks, &partial));
    hipLaunchKernelGGL(sumsq_trace_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, E, (int64_t)m, m, partial, blocks);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, blocks, plan->d_sc + 3);
    NK_HIP(hipGetLastError());
    arena_release(ctx, mk);
  }
  const int tb = (m + 31) / 32;
  hipLaunchKernelGGL(tri_scale_both_kernel, dim3(tb, tb), dim3(256), 0, ctx->stream, plan->W, (int64_t)m, m, plan->d_sc,
                     plan->X0t, plan->X0);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// convergence bookkeeping of the queued iteration: fixed-order sum of the per-block partials of sum (M - I)^2, then
// state[0] = step + 1 of the first step whose residual is below 1e-7 (0: not yet), state[1] = that residual,
// state[2] = last residual seen
__device__ __forceinline__ void ns_flag_kernel_body(const double* __restrict__ partial, int count, int m, int step, double* __restrict__ state) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += blockDim.x) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double r = sqrt((sh[0] + sh[1] + sh[2] + sh[3]) / m);
    state[2] = r;
    if (state[0] == 0.0 && r < 1e-7) {
      state[0] = (double)(step + 1);
      state[1] = r;
    }
  }
}
__global__ void __launch_bounds__(256) ns_flag_kernel(const double* __restrict__ partial, int count, int m, int step, double* __restrict__ state) { ns_flag_kernel_body(partial, count, m, step, state); }
NK_BATCHED_TWIN(ns_flag_kernel, (256), const double*, int, int, int, double*)

// S = Q^T L^T = sqrt(c) Q^T X_0 ;  S^-1 = L^-T Q = (L^-1)^T Q with L^-1 = (extra rows)^T   (Q = the converged iterate)
// Q_even / select: the iterate after an even number of steps and the device word holding the step count (queued form)
static int sqrtm_polar_products(nk_ctx* ctx, SqrtPlan* plan, const double* Q, double* scratch, double c, double* S,
                                double* Sinv, const double* Q_even = nullptr, const double* select = nullptr) {
  const int m = plan->m;
  const size_t mm = (size_t)m * m;
  double* Linv_full = scratch;
  NK_TRY(launch_transpose(ctx, plan->W + mm, m, Linv_full, m, m, m));
  TnProblem pr[2];
  pr[0].A = Q; pr[0].B = plan->X0; pr[0].C = S; pr[0].lda = pr[0].ldb = pr[0].ldc = m; pr[0].M = pr[0].N = m;
  pr[0].alpha = std::sqrt(c);
  pr[0].ktrim = KTRIM_B_UPPER;  // X_0 = L^T / sqrt(c) is upper triangular
  pr[1].ktrim = KTRIM_A_LOWER;  // L^-1 is lower triangular
  pr[1].A = Linv_full; pr[1].B = Q; pr[1].C = Sinv; pr[1].lda = pr[1].ldb = pr[1].ldc = m; pr[1].M = pr[1].N = m;
  pr[0].A_even = Q_even;
  pr[1].B_even = Q_even;
  TnSkip sel;
  sel.select = select;
  if (tn_fast_ok(pr[0]) && tn_fast_ok(pr[1]) && m >= 128) {
    NK_TRY(launch_gemm_tn_multi(ctx, pr, 2, m, 0, nullptr, true, select ? &sel : nullptr));
  } else {
    NK_TRY(launch_gemm(ctx, true, false, m, m, m, pr[0].alpha, Q, m, plan->X0, m, 0.0, S, m));
    NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, Linv_full, m, Q, m, 0.0, Sinv, m));
  }
  return NK_OK;
}

int sqrtm_finish(nk_ctx* ctx, SqrtPlan* plan, double* S, double* Sinv) {
  const int m = plan->m;
  const size_t mm = (size_t)m * m;
  const int ib = info_base(ctx);
  plan->deferred = false;
  plan->rc = NK_OK; plan->iters = 0; plan->resid = 0.0;
  // With a caller-supplied eigenvalue bound (large aligned matrices) nothing of the factorisation is needed to queue the
  // iteration: the host only waits for the three schedule scalars, copied right at the start of sqrtm_prepare.
  plan->early = plan->lambda_min_hint > 0.0 && m >= 1024 && m % 2 == 0;
  double c, sumsq, trace, linv2 = 0.0;
  if (plan->early) {
    NK_HIP(hipEventSynchronize(ctx->ev[11]));
    c = ctx->h_scalars[12]; sumsq = ctx->h_scalars[13]; trace = ctx->h_scalars[14];
    if (!(c > 0.0) || !std::isfinite(c)) plan->early = false;
  }
  if (!plan->early) {
    NK_HIP(hipMemcpyAsync(ctx->h_scalars, plan->d_sc, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipMemcpyAsync(ctx->h_info + ib, ctx->d_info + ib, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));  // the one host round trip of the square root
    c = ctx->h_scalars[0]; sumsq = ctx->h_scalars[1]; trace = ctx->h_scalars[2]; linv2 = ctx->h_scalars[3];
  }
  if (!plan->early &&
      (ctx->h_info[ib] != 0 || !(c > 0.0) || !std::isfinite(c) || !(linv2 > 0.0) || !std::isfinite(linv2))) {
    // not numerically positive definite for the Cholesky route (e.g. a rank-deficient kernel matrix with a jitter below
    // the rounding level): the coupled iteration needs no factorisation
    arena_release(ctx, plan->mark);
    plan->rc = sqrtm_spd_coupled(ctx, plan->P, plan->ldp, m, S, Sinv, &plan->iters, &plan->resid);
    return plan->rc;
  }
  double *Xa = nullptr, *Xta = nullptr, *Xb = nullptr, *Xtb = nullptr, *M = nullptr, *T = nullptr;
  NK_TRY(arena_alloc_t(ctx, mm, &Xa));
  NK_TRY(arena_alloc_t(ctx, mm, &Xta));
  NK_TRY(arena_alloc_t(ctx, mm, &Xb));
  NK_TRY(arena_alloc_t(ctx, mm, &Xtb));
  NK_TRY(arena_alloc_t(ctx, mm, &M));
  NK_TRY(arena_alloc_t(ctx, mm, &T));
  // spectrum interval [a, b] of M_0 = P / c, as in the coupled iteration: b = 1, a = mean of the eigenvalues other than
  // the dominant one (an over-estimate of the smallest eigenvalue: the safe side)
  double a_lo = 1.0, b_hi = 1.0;
  {
    const double fro = std::sqrt(sumsq) / c, tr = trace / c;
    const double lam1 = fro < 1.0 ? fro : 1.0;
    if (m > 1 && tr > lam1) a_lo = (tr - lam1) / (m - 1);
    else a_lo = tr / m * 1e-2;
    if (!(a_lo > 0.0) || !std::isfinite(a_lo)) a_lo = 1e-12;
    if (a_lo > 1.0) a_lo = 1.0;
  }
  auto p3 = [](double x) { return x * (3.0 - x) * (3.0 - x) * 0.25; };
  // one step of the interval recurrence under the scaling s2
  auto advance = [&](double s2, double& a, double& b) {
    const double xa = s2 * a, xb = s2 * b;
    const double lo = p3(xa) < p3(xb) ? p3(xa) : p3(xb);
    b = (xa <= 1.0 && xb >= 1.0) ? 1.0 : (p3(xa) > p3(xb) ? p3(xa) : p3(xb));
    a = lo < b ? lo : b;
  };
  GemmOpts sym;
  sym.tri = TRI_UPPER_MIRROR;
  const double* X = plan->X0;
  const double* Xt = plan->X0t;
  double *Xn = Xa, *Xtn = Xta;
  TnProblem probe;
  probe.A = Xta; probe.B = T; probe.C = Xa; probe.lda = probe.ldb = probe.ldc = m; probe.M = probe.N = m;
  const bool fast = m >= 128 && tn_fast_ok(probe);
  // Queue the whole iteration (no host round trips) for large matrices; small ones are launch bound, and the spare steps
  // the rigorous step budget adds (about five at m = 500, six launches each) would cost more than the few
  // synchronisations of the host-checked loop below.
  const bool queued = fast && m >= 1024;
  if (plan->early && !queued) {  // cannot happen for the shapes `early` is set for; keep the contract simple
    set_error("sqrtm: internal: early queueing needs the LDS-DMA path");
    return NK_ERR_BAD_ARG;
  }

  if (queued) {
    // ---- the whole iteration is queued without host round trips.  The step count is data dependent, so (a) a rigorous
    // lower bound of the smallest eigenvalue, lambda_min(P) >= 1 / ||L^-1||_F^2, run through the scaling schedule
    // gives the latest step kmax at which the iteration can converge, and (b) the launches of steps after the one that
    // actually converged return at once on the device (TnSkip), and the two final products pick the buffer that holds
    // the converged iterate by the parity of the step count (also on the device).
    double s2s[128];
    bool checks[128];
    int kmax = 0;
    {
      // true interval: [lower bound, 1]; the bound is the caller's (jitter) or 1 / ||L^-1||_F^2
      double ta = plan->early ? 0.9 * plan->lambda_min_hint / c : 0.5 / (c * linv2), tb = 1.0;
      if (ta > a_lo) ta = a_lo;
      // Interval the scaling schedule is built for.  Any lower end is safe (the scaled step keeps every eigenvalue <= b
      // inside (0, 3)); it only decides how long the steps stay aggressively scaled.  The mean-of-the-bulk over-estimate
      // stops scaling after ~4 steps and leaves the smallest eigenvalues to the unscaled 2.25x growth, the rigorous
      // bound keeps scaling for steps nobody needs: a weighted geometric mean (0.8 / 0.2) is used.  Steps at C4, m = 2000,
      // lengthscales 5 / 10 / 20 / 40 / 80: 7 / 9 / 10 / 13 / 16 against 6 / 10 / 13 / 15 / 18 with the over-estimate alone
      // and 8 / 12 / 12 / 13 / - with equal weights.
      const double a_sched = std::pow(a_lo, 0.8) * std::pow(ta, 0.2);
      double a = a_sched, b = b_hi;
      int kconv = -1;
      for (int k = 0; k < 100; ++k) {
        if (kconv < 0 && ta >= 1.0 - 1e-9 && tb <= 1.0 + 1e-9) kconv = k;  // M_k is within the 1e-7 residual bar
        checks[k] = a >= 0.5;
        const double s2 = 3.0 / (a + std::sqrt(a * b) + b);
        s2s[k] = s2;
        advance(s2, a, b);
        advance(s2, ta, tb);
        if (kconv >= 0 && k >= kconv + 1) { kmax = k + 1; break; }  // one spare step beyond the predicted last one
      }
      if (kmax == 0) kmax = 100;
    }
    double* state = plan->d_sc + 5;
    NK_HIP(hipMemsetAsync(state, 0, 3 * sizeof(double), ctx->stream));
    double* rpart = nullptr;
    {
      const size_t mt = (size_t)(m + 127) / 128;
      const size_t need = std::max((size_t)grid_for((int64_t)m * m, ctx->num_cu), mt * (mt + 1) / 2 * 8);
      NK_TRY(arena_alloc_t(ctx, need, &rpart));
    }
    for (int k = 0; k < kmax; ++k) {
      TnSkip skip;
      skip.state = state; skip.step = k;
      int npart = 0;  // residual partials written by the reduce kernel of the M product (k > 0)
      if (k == 0) {
        NK_TRY(launch_transpose(ctx, plan->P, plan->ldp, M, m, m, m));
        NK_TRY(launch_axpby2d(ctx, 0.5 / c, plan->P, plan->ldp, 0.5 / c, M, m, m, m));
      } else {
        // M = X^T X and, from the same epilogue, T = s (3 I - s^2 M) / 2 (its coefficients come from the schedule)
        const double s2k = s2s[k], sck = std::sqrt(s2k);
        TnProblem pm;
        pm.A = X; pm.B = X; pm.C = M; pm.lda = pm.ldb = pm.ldc = m; pm.M = pm.N = m; pm.tri = TRI_UPPER_MIRROR;
        pm.Caff = T; pm.aff_a = -0.5 * s2k * sck; pm.aff_c = 1.5 * sck;
        TnSkip skip_m = skip;
        skip_m.resid_partials = rpart; skip_m.resid_count = &npart;
        NK_TRY(launch_gemm_tn_multi(ctx, &pm, 1, m, 0, nullptr, true, &skip_m));
      }
      if (checks[k]) {
        if (npart == 0) {  // M_0 (no product) or a single-slice product: separate pass over M for the partials
          npart = grid_for((int64_t)m * m, ctx->num_cu);
          hipLaunchKernelGGL(frob_mi_partial_kernel, dim3(npart), dim3(256), 0, ctx->stream, M, (int64_t)m, m, rpart,
                             skip.state, skip.step);
        }
        // sum of the partials in index order and the convergence flag (one launch)
        hipLaunchKernelGGL(ns_flag_kernel, dim3(1), dim3(256), 0, ctx->stream, rpart, npart, m, k, state);
        NK_HIP(hipGetLastError());
      }
      if (k == 0) {
        const double s2 = s2s[0], sc = std::sqrt(s2);
        NK_TRY(launch_scale_add_identity(ctx, -0.5 * s2 * sc, M, m, 1.5 * sc, T, m, m, &skip));
      }
      TnProblem pr;
      pr.A = Xt; pr.B = T; pr.C = Xn; pr.lda = pr.ldb = pr.ldc = m; pr.M = pr.N = m; pr.Ct = Xtn; pr.ldct = m;
      // one K slice: at most one workgroup slot per CU is taken, the other stays free for the factorisation chain on the
      // main stream (a two-slice launch would take every slot for its whole duration)
      NK_TRY(launch_gemm_tn_multi(ctx, &pr, 1, m, 1, nullptr, true, &skip));
      X = Xn; Xt = Xtn;
      Xn = (Xn == Xa) ? Xb : Xa;
      Xtn = (Xtn == Xta) ? Xtb : Xta;
    }
    // X_j lives in Xa for odd j and in Xb for even j; the step count state[0] = j picks the operand on the device
    NK_TRY(sqrtm_polar_products(ctx, plan, Xa, T, c, S, Sinv, Xb, state));
    NK_HIP(hipMemcpyAsync(ctx->h_scalars + 8, state, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (plan->early)  // the verdict of the factorisation travels with the verdict of the iteration
      NK_HIP(hipMemcpyAsync(ctx->h_info + ib, ctx->d_info + ib, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    plan->deferred = true;
    plan->kmax = kmax;
    arena_release(ctx, plan->mark);
    return NK_OK;
  }

  // ---- small / unaligned matrices: convergence read by the host (one step behind the queue)
  const int maxit = 100;
  double r = 1e300;
  int it = 0;
  bool ok = false;
  for (; it < maxit; ++it) {
    if (it == 0) {
      // M_0 = X_0^T X_0 = L L^T / c = P / c, taken as the average of P and P^T (no GEMM)
      NK_TRY(launch_transpose(ctx, plan->P, plan->ldp, M, m, m, m));
      NK_TRY(launch_axpby2d(ctx, 0.5 / c, plan->P, plan->ldp, 0.5 / c, M, m, m, m));
    } else {
      NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, X, m, X, m, 0.0, M, m, sym));  // M = X^T X
    }
    const bool check = a_lo >= 0.5 || it + 2 >= maxit;
    if (check) {
      NK_TRY(launch_frob_minus_identity(ctx, M, m, m, ctx->d_scalars));
      NK_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      NK_HIP(hipEventRecord(ctx->ev[10], ctx->stream));
    }
    const double s2 = 3.0 / (a_lo + std::sqrt(a_lo * b_hi) + b_hi);
    const double sc = std::sqrt(s2);
    advance(s2, a_lo, b_hi);
    NK_TRY(launch_scale_add_identity(ctx, -0.5 * s2 * sc, M, m, 1.5 * sc, T, m, m));
    if (fast) {  // X T with the transposed copy from the epilogue
      TnProblem pr;
      pr.A = Xt; pr.B = T; pr.C = Xn; pr.lda = pr.ldb = pr.ldc = m; pr.M = pr.N = m; pr.Ct = Xtn; pr.ldct = m;
      NK_TRY(launch_gemm_tn_multi(ctx, &pr, 1, m, 0));
    } else {
      NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, Xt, m, T, m, 0.0, Xn, m));
      NK_TRY(launch_transpose(ctx, Xn, m, Xtn, m, m, m));
    }
    X = Xn; Xt = Xtn;
    Xn = (Xn == Xa) ? Xb : Xa;
    Xtn = (Xtn == Xta) ? Xtb : Xta;
    if (check) {
      NK_HIP(hipEventSynchronize(ctx->ev[10]));
      r = std::sqrt(ctx->h_scalars[0] / m);
      if (!std::isfinite(r)) break;
      if (r < 1e-7) {
        ok = true;
        ++it;
        break;
      }
    }
  }
  plan->iters = it;
  plan->resid = r;
  NK_TRY(x_align());  // lock-step groups: the iteration count differs from unit to unit; re-align the launch sequences here
  if (!ok) {
    set_error("sqrtm: Newton-Schulz did not converge (residual %g after %d iterations)", r, it);
    arena_release(ctx, plan->mark);
    plan->rc = NK_ERR_NO_CONVERGENCE;
    return plan->rc;
  }
  NK_TRY(sqrtm_polar_products(ctx, plan, X, T, c, S, Sinv));
  arena_release(ctx, plan->mark);
  return NK_OK;
}

int sqrtm_verdict(nk_ctx* ctx, SqrtPlan* plan, int* iters, double* resid) {
  if (plan->deferred) {
    const double flag = ctx->h_scalars[8];
    plan->deferred = false;
    if (plan->early && (ctx->h_info[2] != 0 || flag == 0.0 || !std::isfinite(ctx->h_scalars[10]))) {
      plan->iters = plan->kmax;
      plan->resid = ctx->h_scalars[10];
      plan->rc = NK_SQRT_RETRY;
    } else if (flag == 0.0 || !std::isfinite(ctx->h_scalars[10])) {
      plan->iters = plan->kmax;
      plan->resid = ctx->h_scalars[10];
      plan->rc = NK_ERR_NO_CONVERGENCE;
      set_error("sqrtm: Newton-Schulz did not converge (residual %g after %d iterations)", plan->resid, plan->kmax);
    } else {
      plan->iters = (int)flag;
      plan->resid = ctx->h_scalars[9];
      plan->rc = NK_OK;
    }
  }
  if (iters) *iters = plan->iters;
  if (resid) *resid = plan->resid;
  return plan->rc;
}

int sqrtm_spd(nk_ctx* ctx, const double* P, int64_t ldp, int m, double* S, double* Sinv, int* iters, double* resid) {
  SqrtPlan plan;
  NK_TRY(sqrtm_prepare(ctx, P, ldp, m, &plan));
  NK_TRY(sqrtm_finish(ctx, &plan, S, Sinv));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return sqrtm_verdict(ctx, &plan, iters, resid);
}

}  // namespace nk
