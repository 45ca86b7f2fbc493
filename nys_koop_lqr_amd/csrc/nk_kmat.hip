// Kernel-matrix builds on gfx950: out[i][j] = k(A[i,:], B[j,:]).
//
// Replaces `kern.kernel(A, B)` of the reference (regressors.py:22,26,30, called at :139,141-144,174,176), i.e.
// sklearn RBF / Matern(nu=2.5) / DotProduct __call__ -> scipy cdist.  Like cdist, squared distances are
// accumulated from DIRECT differences sum_k ((a_k - b_k)/l_k)^2 in fp64 (no ||a||^2+||b||^2-2ab expansion), so
// coincident points give exactly 0 and k = 1, and K(Z, Z) is bitwise symmetric.
//
// One 256-thread workgroup produces a 128 x 128 tile; each thread owns an 8 x 8 register tile (64 fp64
// accumulators).  The two row panels are staged through LDS in slices of 16 dimensions, pre-multiplied by
// 1/lengthscale, stored [k][row] so that the per-k operand fetch is 4+4 ds_read_b128 (broadcast across the
// 16 lanes that share a row group).  The inner loop is v_add_f64 + v_fma_f64 per (i, j, k): fp64 VALU bound at
// large d (d = 384: 3*d/8 = 144 flop per output byte); at d <= 2 the kernel is bound by the HBM write of `out`.
#include "nk_common.h"

namespace nk {

constexpr int KT = 128;      // tile edge
constexpr int KBK = 16;      // dimensions per LDS slice
constexpr int KSTRIDE = 128; // doubles per LDS row

template <int KTYPE>
__device__ __forceinline__ double kmat_epilogue(double acc, double sigma0sq) {
#pragma clang fp contract(off)  // the same operations in every kernel that inlines this (bits do not depend on the shape)
  if (KTYPE == NK_KERNEL_RBF) {
    return exp_nonpos(-0.5 * acc);
  } else if (KTYPE == NK_KERNEL_MATERN52) {
    const double t = sqrt(acc) * 2.23606797749978969641;  // sqrt(5) * r
    return (1.0 + t + t * t / 3.0) * exp_nonpos(-t);
  } else {
    return acc + sigma0sq;
  }
}

template <int KTYPE>
__device__ __forceinline__ void kmat_body(const double* __restrict__ A, int64_t lda, int nA, const double* __restrict__ B, int64_t ldb, int nB, int d, const double* __restrict__ winv, double sigma0sq, double* __restrict__ out, int64_t ldo, int vecA, int vecB, int vecO) {
  __shared__ __attribute__((aligned(16))) double As[KBK * KSTRIDE];
  __shared__ __attribute__((aligned(16))) double Bs[KBK * KSTRIDE];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.y * KT, j0 = blockIdx.x * KT;

  double acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;

  // staging assignment: row = tid & 127, k half = tid >> 7 (8 consecutive dimensions per thread)
  const int srow = tid & 127, skh = tid >> 7;
  const bool a_ok = i0 + srow < nA, b_ok = j0 + srow < nB;
  const double* a_src = A + (int64_t)(i0 + srow) * lda;
  const double* b_src = B + (int64_t)(j0 + srow) * ldb;

  for (int k0 = 0; k0 < d; k0 += KBK) {
    const int kb = k0 + skh * 8;
    double va[8], vb[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = (kb + q < d) ? winv[kb + q] : 0.0;
    if (a_ok && vecA && kb + 7 < d) {
      const double2* s2 = reinterpret_cast<const double2*>(a_src + kb);
#pragma unroll
      for (int q = 0; q < 4; ++q) { double2 v = s2[q]; va[2 * q] = v.x; va[2 * q + 1] = v.y; }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) va[q] = (a_ok && kb + q < d) ? a_src[kb + q] : 0.0;
    }
    if (b_ok && vecB && kb + 7 < d) {
      const double2* s2 = reinterpret_cast<const double2*>(b_src + kb);
#pragma unroll
      for (int q = 0; q < 4; ++q) { double2 v = s2[q]; vb[2 * q] = v.x; vb[2 * q + 1] = v.y; }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) vb[q] = (b_ok && kb + q < d) ? b_src[kb + q] : 0.0;
    }
    __syncthreads();  // previous slice fully consumed
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      As[(skh * 8 + q) * KSTRIDE + srow] = va[q] * w[q];
      Bs[(skh * 8 + q) * KSTRIDE + srow] = vb[q] * w[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KBK; ++k) {
      double a[8], b[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double2 v = *reinterpret_cast<const double2*>(&As[k * KSTRIDE + ty * 2 + 32 * r]);
        a[2 * r] = v.x; a[2 * r + 1] = v.y;
        const double2 u = *reinterpret_cast<const double2*>(&Bs[k * KSTRIDE + tx * 2 + 32 * r]);
        b[2 * r] = u.x; b[2 * r + 1] = u.y;
      }
#pragma unroll
      for (int ia = 0; ia < 8; ++ia)
#pragma unroll
        for (int ib = 0; ib < 8; ++ib) {
          if (KTYPE == NK_KERNEL_LINEAR) {
            acc[ia][ib] = fma(a[ia], b[ib], acc[ia][ib]);
          } else {
            const double df = a[ia] - b[ib];
            acc[ia][ib] = fma(df, df, acc[ia][ib]);
          }
        }
    }
  }

#pragma unroll
  for (int ia = 0; ia < 8; ++ia) {
    const int i = i0 + ty * 2 + 32 * (ia >> 1) + (ia & 1);
    if (i >= nA) continue;
    double* orow = out + (int64_t)i * ldo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + tx * 2 + 32 * c;
      const double v0 = kmat_epilogue<KTYPE>(acc[ia][2 * c], sigma0sq);
      const double v1 = kmat_epilogue<KTYPE>(acc[ia][2 * c + 1], sigma0sq);
      if (vecO && j + 1 < nB) {
        *reinterpret_cast<double2*>(orow + j) = make_double2(v0, v1);
      } else {
        if (j < nB) orow[j] = v0;
        if (j + 1 < nB) orow[j + 1] = v1;
      }
    }
  }
}
template <int KTYPE>
__global__ void __launch_bounds__(256, 2)
kmat_kernel(const double* __restrict__ A, int64_t lda, int nA, const double* __restrict__ B, int64_t ldb, int nB, int d, const double* __restrict__ winv, double sigma0sq, double* __restrict__ out, int64_t ldo, int vecA, int vecB, int vecO) { kmat_body<KTYPE>(A, lda, nA, B, ldb, nB, d, winv, sigma0sq, out, ldo, vecA, vecB, vecO); }
template <int KTYPE>
__global__ void __launch_bounds__(256, 2) kmat_kernel_batched(const nk::ArgPack<const double*, int64_t, int, const double*, int64_t, int, int, const double*, double, double*, int64_t, int, int, int>* table) {
  const nk::ArgPack<const double*, int64_t, int, const double*, int64_t, int, int, const double*, double, double*, int64_t, int, int, int> p = table[blockIdx.z];
  nk::pack_apply([](auto... a) { kmat_body<KTYPE>(a...); }, p);
}
#define NK_KMAT_TWIN(K)                                                                                              \
  static nk::TwinReg kmat_twin_reg_##K(                                                                              \
      reinterpret_cast<const void*>(static_cast<void (*)(const double*, int64_t, int, const double*, int64_t, int, int, const double*, double, double*, int64_t, int, int, int)>(kmat_kernel<K>)),                                  \
      reinterpret_cast<const void*>(kmat_kernel_batched<K>), sizeof(nk::ArgPack<const double*, int64_t, int, const double*, int64_t, int, int, const double*, double, double*, int64_t, int, int, int>), "kmat_kernel<" #K ">");
NK_KMAT_TWIN(0) NK_KMAT_TWIN(1) NK_KMAT_TWIN(2)


// ---------------------------------------------------------------------------------------------------------------
// Small state dimension (d <= 8: the Duffing / HJB configs of benchmark_lqr_classic.py and _hjb.py, d = 2 and 1).  The
// build is bound by the HBM WRITE of the output (8 B per entry against <= 2 d flop of distance work): the tiled kernel
// above spends its time in LDS staging and 8 x 8 register tiles that have nothing to amortise.  Here the output (row-major, rows of
// nB entries at stride ldo) is walked as one flat index space: a thread produces two consecutive entries per step and stores them as one 16-byte
// word, a wave writes 1 KiB contiguous, the grid strides over the array.  The landmark rows (pre-scaled by 1/lengthscale)
// sit in LDS, the sample row is read through the cache (64 lanes share at most two rows).  Same arithmetic per entry as
// the tiled kernel (differences of pre-scaled coordinates, sequential FMA over the dimensions), hence the same bits.
// ---------------------------------------------------------------------------------------------------------------
constexpr int KFLAT_DMAX = 8;
template <int KTYPE>
__device__ __forceinline__ void kmat_flat_body(const double* __restrict__ A, int64_t lda, int64_t nA, const double* __restrict__ B,
                                               int64_t ldb, int nB, int d, const double* __restrict__ winv, double sigma0sq,
                                               double* __restrict__ out, int64_t ldo, int64_t step_i, int step_j) {
#pragma clang fp contract(off)  // (a w) - (b w) with both products rounded, as in the tiled kernel: same bits
  extern __shared__ __attribute__((aligned(16))) double kf_lds[];
  double* Bs = kf_lds;          // nB x d, pre-scaled
  double* ws = Bs + (size_t)nB * d;
  for (int e = threadIdx.x; e < nB * d; e += 256) {
    const int j = e / d, k = e - j * d;
    Bs[e] = B[(int64_t)j * ldb + k] * winv[k];
  }
  if (threadIdx.x < d) ws[threadIdx.x] = winv[threadIdx.x];
  __syncthreads();
  const int64_t total = nA * (int64_t)nB;
  int64_t e = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
  int64_t i = e / nB;
  int j = (int)(e - i * nB);
  for (; e < total; e += 2 * (int64_t)gridDim.x * 256) {
    const double* arow = A + i * lda;
    double acc0 = 0.0, acc1 = 0.0;
    const double* b0 = Bs + (size_t)j * d;
#pragma unroll
    for (int k = 0; k < KFLAT_DMAX; ++k) {
      if (k < d) {
        const double a = arow[k] * ws[k];
        if (KTYPE == NK_KERNEL_LINEAR) {
          acc0 = fma(a, b0[k], acc0);
          acc1 = fma(a, b0[d + k], acc1);
        } else {
          const double d0 = a - b0[k], d1 = a - b0[d + k];
          acc0 = fma(d0, d0, acc0);
          acc1 = fma(d1, d1, acc1);
        }
      }
    }
    *reinterpret_cast<double2*>(out + i * ldo + j) = make_double2(kmat_epilogue<KTYPE>(acc0, sigma0sq), kmat_epilogue<KTYPE>(acc1, sigma0sq));
    i += step_i;
    j += step_j;
    if (j >= nB) { j -= nB; ++i; }
  }
}
template <int KTYPE>
__global__ void __launch_bounds__(256) kmat_flat_kernel(const double* __restrict__ A, int64_t lda, int64_t nA,
                                                        const double* __restrict__ B, int64_t ldb, int nB, int d,
                                                        const double* __restrict__ winv, double sigma0sq,
                                                        double* __restrict__ out, int64_t ldo, int64_t step_i,
                                                        int step_j) {
  kmat_flat_body<KTYPE>(A, lda, nA, B, ldb, nB, d, winv, sigma0sq, out, ldo, step_i, step_j);
}
template <int KTYPE>
__global__ void __launch_bounds__(256) kmat_flat_kernel_batched(
    const nk::ArgPack<const double*, int64_t, int64_t, const double*, int64_t, int, int, const double*, double, double*, int64_t, int64_t, int>* table) {
  const nk::ArgPack<const double*, int64_t, int64_t, const double*, int64_t, int, int, const double*, double, double*, int64_t, int64_t, int> p =
      table[blockIdx.z];
  nk::pack_apply([](auto... a) { kmat_flat_body<KTYPE>(a...); }, p);
}
#define NK_KFLAT_TWIN(K)                                                                                                  \
  static nk::TwinReg kflat_twin_reg_##K(                                                                                  \
      reinterpret_cast<const void*>(static_cast<void (*)(const double*, int64_t, int64_t, const double*, int64_t, int, int, \
                                                         const double*, double, double*, int64_t, int64_t, int)>(kmat_flat_kernel<K>)), \
      reinterpret_cast<const void*>(kmat_flat_kernel_batched<K>),                                                           \
      sizeof(nk::ArgPack<const double*, int64_t, int64_t, const double*, int64_t, int, int, const double*, double, double*, \
                         int64_t, int64_t, int>),                                                                                  \
      "kmat_flat_kernel<" #K ">");
NK_KFLAT_TWIN(0) NK_KFLAT_TWIN(1) NK_KFLAT_TWIN(2)

// ---------------------------------------------------------------------------------------------------------------
// A handful of entries (the lift or prediction of ONE state, regressors.py:171-178 inside a control loop; K(Z, Z) at
// tiny m): latency is all that matters and the tiled kernel would run in one or two workgroups whose threads mostly hold
// padding (71 us for a 1 x 100 row at d = 192).  One WAVE per entry: the 64 lanes fetch 64 consecutive dimensions of the
// two rows (coalesced) and form the scaled differences, then the sum over the dimensions is accumulated IN INDEX ORDER by
// a chain of FMAs whose operands are broadcast from lane q = 0, 1, ... by v_readlane (uniform, every lane runs the same
// chain).  Per entry that is, operation by operation, the arithmetic of the tiled kernel (products with 1/lengthscale
// rounded separately, difference, FMA over the dimensions in index order; padding adds exact zeros), hence the same
// bits whichever kernel a shape selects.
// ---------------------------------------------------------------------------------------------------------------
constexpr int64_t KSMALL_MAX_ENTRIES = 8192;
__device__ __forceinline__ double readlane_f64(double v, int q) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), q);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), q);
  return __hiloint2double(hi, lo);
}
template <int KTYPE>
__device__ __forceinline__ void kmat_small_body(const double* __restrict__ A, int64_t lda, int nA, const double* __restrict__ B,
                                                int64_t ldb, int nB, int d, const double* __restrict__ winv, double sigma0sq,
                                                double* __restrict__ out, int64_t ldo) {
#pragma clang fp contract(off)  // a w - b w must not become fma(a, w, -(b w)): K(Z, Z) is bitwise symmetric
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= (int64_t)nA * nB) return;  // wave-uniform
  const int i = (int)(e / nB), j = (int)(e - (int64_t)i * nB);
  const double* a = A + (int64_t)i * lda;
  const double* b = B + (int64_t)j * ldb;
  double acc = 0.0;
  for (int k0 = 0; k0 < d; k0 += 64) {
    const int k = k0 + lane;
    double x = 0.0, y = 0.0;
    if (k < d) {
      const double w = winv[k];
      x = a[k] * w;
      y = b[k] * w;
    }
    const int cnt = d - k0 < 64 ? d - k0 : 64;
    if (KTYPE == NK_KERNEL_LINEAR) {
      int q = 0;
      for (; q + 8 <= cnt; q += 8) {
        double sx[8], sy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { sx[u] = readlane_f64(x, q + u); sy[u] = readlane_f64(y, q + u); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(sx[u], sy[u], acc);
      }
      for (; q < cnt; ++q) acc = fma(readlane_f64(x, q), readlane_f64(y, q), acc);
    } else {
      const double df = x - y;
      int q = 0;
      for (; q + 8 <= cnt; q += 8) {  // the broadcasts of eight steps are independent of the chain: issued ahead of it
        double s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = readlane_f64(df, q + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(s[u], s[u], acc);
      }
      for (; q < cnt; ++q) {
        const double s = readlane_f64(df, q);
        acc = fma(s, s, acc);
      }
    }
  }
  if (lane == 0) out[(int64_t)i * ldo + j] = kmat_epilogue<KTYPE>(acc, sigma0sq);
}
template <int KTYPE>
__global__ void __launch_bounds__(256) kmat_small_kernel(const double* __restrict__ A, int64_t lda, int nA,
                                                        const double* __restrict__ B, int64_t ldb, int nB, int d,
                                                        const double* __restrict__ winv, double sigma0sq,
                                                        double* __restrict__ out, int64_t ldo) {
  kmat_small_body<KTYPE>(A, lda, nA, B, ldb, nB, d, winv, sigma0sq, out, ldo);
}
template <int KTYPE>
__global__ void __launch_bounds__(256) kmat_small_kernel_batched(
    const nk::ArgPack<const double*, int64_t, int, const double*, int64_t, int, int, const double*, double, double*, int64_t>* table) {
  const nk::ArgPack<const double*, int64_t, int, const double*, int64_t, int, int, const double*, double, double*, int64_t> p =
      table[blockIdx.z];
  nk::pack_apply([](auto... a) { kmat_small_body<KTYPE>(a...); }, p);
}
#define NK_KSMALL_TWIN(K)                                                                                                    \
  static nk::TwinReg ksmall_twin_reg_##K(                                                                                    \
      reinterpret_cast<const void*>(static_cast<void (*)(const double*, int64_t, int, const double*, int64_t, int, int,      \
                                                         const double*, double, double*, int64_t)>(kmat_small_kernel<K>)),   \
      reinterpret_cast<const void*>(kmat_small_kernel_batched<K>),                                                           \
      sizeof(nk::ArgPack<const double*, int64_t, int, const double*, int64_t, int, int, const double*, double, double*,      \
                         int64_t>),                                                                                          \
      "kmat_small_kernel<" #K ">");
NK_KSMALL_TWIN(0) NK_KSMALL_TWIN(1) NK_KSMALL_TWIN(2)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_kmat(nk_ctx* ctx, int ktype, const double* A, int64_t lda, int64_t nA, const double* B, int64_t ldb,
                int64_t nB, int d, const double* winv, double sigma0, double* out, int64_t ldo) {
  if (nA <= 0 || nB <= 0) return NK_OK;
  NK_REQUIRE(nA < (1LL << 31) && nB < (1LL << 31) && d > 0, "kernel matrix: dimension out of range");
  dim3 grid((unsigned)((nB + KT - 1) / KT), (unsigned)((nA + KT - 1) / KT));
  NK_REQUIRE(grid.y <= 65535u, "kernel matrix: too many row tiles");
  const int vecA = aligned16(A) && lda % 2 == 0, vecB = aligned16(B) && ldb % 2 == 0;
  const int vecO = aligned16(out) && ldo % 2 == 0;
  const double s2 = sigma0 * sigma0;
  if (d <= KFLAT_DMAX && ldo % 2 == 0 && nB % 2 == 0 && aligned16(out) && (size_t)(nB * d + d) * 8 <= 60 * 1024 &&
      nA * nB >= 4096 && ktype >= NK_KERNEL_RBF && ktype <= NK_KERNEL_LINEAR) {
    // write-bound regime: flat streaming kernel
    const int64_t pairs = nA * nB / 2;
    int64_t blocks = (pairs + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    const int64_t stride = 2 * blocks * 256;  // elements advanced per grid-stride step
    const int64_t step_i = stride / nB;
    const int step_j = (int)(stride - step_i * nB);
    const size_t lds = (size_t)(nB * d + d) * 8;
    const dim3 g((unsigned)blocks);
    if (ktype == NK_KERNEL_RBF)
      hipLaunchKernelGGL((kmat_flat_kernel<NK_KERNEL_RBF>), g, dim3(256), lds, ctx->stream, A, lda, nA, B, ldb, (int)nB, d,
                         winv, s2, out, ldo, step_i, step_j);
    else if (ktype == NK_KERNEL_MATERN52)
      hipLaunchKernelGGL((kmat_flat_kernel<NK_KERNEL_MATERN52>), g, dim3(256), lds, ctx->stream, A, lda, nA, B, ldb, (int)nB,
                         d, winv, s2, out, ldo, step_i, step_j);
    else
      hipLaunchKernelGGL((kmat_flat_kernel<NK_KERNEL_LINEAR>), g, dim3(256), lds, ctx->stream, A, lda, nA, B, ldb, (int)nB, d,
                         winv, s2, out, ldo, step_i, step_j);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  if (nA * nB <= KSMALL_MAX_ENTRIES && ktype >= NK_KERNEL_RBF && ktype <= NK_KERNEL_LINEAR) {
    const dim3 g((unsigned)((nA * nB + 3) / 4));  // one wave per entry
    if (ktype == NK_KERNEL_RBF)
      hipLaunchKernelGGL((kmat_small_kernel<NK_KERNEL_RBF>), g, dim3(256), 0, ctx->stream, A, lda, (int)nA, B, ldb, (int)nB, d,
                         winv, s2, out, ldo);
    else if (ktype == NK_KERNEL_MATERN52)
      hipLaunchKernelGGL((kmat_small_kernel<NK_KERNEL_MATERN52>), g, dim3(256), 0, ctx->stream, A, lda, (int)nA, B, ldb,
                         (int)nB, d, winv, s2, out, ldo);
    else
      hipLaunchKernelGGL((kmat_small_kernel<NK_KERNEL_LINEAR>), g, dim3(256), 0, ctx->stream, A, lda, (int)nA, B, ldb,
                         (int)nB, d, winv, s2, out, ldo);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  switch (ktype) {
    case NK_KERNEL_RBF:
      hipLaunchKernelGGL((kmat_kernel<NK_KERNEL_RBF>), grid, dim3(256), 0, ctx->stream, A, lda, (int)nA, B, ldb,
                         (int)nB, d, winv, s2, out, ldo, vecA, vecB, vecO);
      break;
    case NK_KERNEL_MATERN52:
      hipLaunchKernelGGL((kmat_kernel<NK_KERNEL_MATERN52>), grid, dim3(256), 0, ctx->stream, A, lda, (int)nA, B, ldb,
                         (int)nB, d, winv, s2, out, ldo, vecA, vecB, vecO);
      break;
    case NK_KERNEL_LINEAR:
      hipLaunchKernelGGL((kmat_kernel<NK_KERNEL_LINEAR>), grid, dim3(256), 0, ctx->stream, A, lda, (int)nA, B, ldb,
                         (int)nB, d, winv, s2, out, ldo, vecA, vecB, vecO);
      break;
    default:
      set_error("unknown kernel type %d", ktype);
      return NK_ERR_BAD_ARG;
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
