// Diagonal-block Cholesky of the blocked factorisation (nk_linalg.hip), in its own translation unit because the fully
// unrolled register-resident kernel takes minutes to compile.  Replaces the LAPACK potrf inside lstsq's role
// (regressors.py:155,165).
#include "nk_common.h"

namespace nk {

// ONE WAVE factorises a 64 x 64 diagonal block entirely in registers and inverts the factor.
//   Factor: lane i owns row i (64 fp64 = 128 VGPRs); right-looking column Cholesky, column k of the other rows is
//   fetched with v_readlane (no LDS, no barriers): sum_k (63-k) = 2016 readlane pairs + FMAs, fully unrolled.
//   Inverse: L goes to LDS (stride 65, conflict-free), lane c owns column c of X = L^-1 (forward substitution with
//   broadcast LDS reads of L).  Blocks shorter than 64 are padded with the identity.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

struct PotrfBatch {
  double* A[2];
  int64_t lda[2];
  int nb[2];
  double* Linv[2];
  int* info[2];
};

__global__ void __launch_bounds__(64) potrf_diag_kernel(PotrfBatch pb, int blk) {
  constexpr int NB = CHOL_NB;
  const int which = blockIdx.x;  // one wave per system of a paired factorisation
  const int nb = pb.nb[which];
  if (nb <= 0) return;
  // this wave is the critical path of the factorisation chain and may share its CU with GEMM waves of the side
  // stream: take the issue slots first
  __builtin_amdgcn_s_setprio(3);
  double* __restrict__ A = pb.A[which];
  const int64_t lda = pb.lda[which];
  double* __restrict__ Linv = pb.Linv[which];
  int* __restrict__ info = pb.info[which];
  static_assert(NB == 64, "one lane per row");
  __shared__ double Ls[NB * (NB + 1)];
  __shared__ double dinv[NB];
  const int lane = threadIdx.x;
  double R[NB];
  // whole rows are loaded (the upper triangle is never consumed below: column k is only read from lanes >= k), which
  // keeps the load free of per-element lane masks; rows / columns beyond nb are identity padding
  const bool row_ok = lane < nb;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double v = (j == lane) ? 1.0 : 0.0;
    if (row_ok && j < nb) v = A[(int64_t)lane * lda + j];
    R[j] = v;
  }
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    double dkk = readlane_f64(R[k], k);
    if (!(dkk > 0.0) || !isfinite(dkk)) {
      if (lane == 0 && k < nb) atomicCAS(info, 0, blk * NB + k + 1);
      dkk = 1.0;
    }
    const double s = sqrt(dkk);
    const double inv = 1.0 / s;
    R[k] = (lane == k) ? s : R[k] * inv;  // lanes > k: L[i][k]; lanes < k hold unused upper-triangle values
    // R[j] -= L[i][k] * L[j][k]: L[j][k] is lane j's R[k].  readlane -> SGPR pair -> scalar operand of the FMA, kept in
    // ONE asm statement per j: left to itself hipcc hoists all 63-k readlanes ahead of the FMAs and spills them
    // through v_writelane (3200 SGPR spills, 82 us per block).
    {
      const int rk_lo = __double2loint(R[k]), rk_hi = __double2hiint(R[k]);
#pragma unroll
      for (int j = k + 1; j < NB; ++j)
        asm volatile("v_readlane_b32 s96, %1, %3\n\tv_readlane_b32 s97, %2, %3\n\ts_nop 0\n\tv_fma_f64 %0, -%4, s[96:97], %0"
                     : "+v"(R[j])
                     : "v"(rk_lo), "v"(rk_hi), "i"(j), "v"(R[k])
                     : "s96", "s97");
    }
  }
  // L -> LDS and global (lower triangle)
  // rows go back whole as well: nothing reads the upper triangle of a factored diagonal block (the solves use Linv,
  // the trailing updates the panels below)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    Ls[lane * (NB + 1) + j] = R[j];
    if (row_ok && j < nb) A[(int64_t)lane * lda + j] = R[j];
  }
  __syncthreads();
  dinv[lane] = 1.0 / Ls[lane * (NB + 1) + lane];
  __syncthreads();
  // X = L^-1, column `lane`
  double X[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    double s0 = (i == lane) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four chains hide the FMA latency
#pragma unroll
    for (int j = 0; j < i; ++j) {
      const double l = Ls[i * (NB + 1) + j];
      if ((j & 3) == 0) s0 = fma(-l, X[j], s0);
      else if ((j & 3) == 1) s1 = fma(-l, X[j], s1);
      else if ((j & 3) == 2) s2 = fma(-l, X[j], s2);
      else s3 = fma(-l, X[j], s3);
    }
    X[i] = ((s0 + s1) + (s2 + s3)) * dinv[i];
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) Linv[i * NB + lane] = X[i];
}

// Ajj/lda/nb/Linv: per system (nb <= 0 skips a system); failures are flagged in ctx->d_info[system]
int launch_potrf_diag_pair(nk_ctx* ctx, double* const* Ajj, const int64_t* lda, const int* nb, double* const* Linv,
                           int nsys, int blk) {
  PotrfBatch pb;
  for (int q = 0; q < 2; ++q) {
    const bool on = q < nsys;
    pb.A[q] = on ? Ajj[q] : nullptr;
    pb.lda[q] = on ? lda[q] : 0;
    pb.nb[q] = on ? nb[q] : 0;
    pb.Linv[q] = on ? Linv[q] : nullptr;
    pb.info[q] = ctx->d_info + q;
  }
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(2), dim3(64), 0, ctx->stream, pb, blk);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
