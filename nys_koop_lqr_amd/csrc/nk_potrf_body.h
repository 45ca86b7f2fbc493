// Diagonal-block Cholesky of the blocked factorisation (nk_linalg.hip), in its own translation unit (fully unrolled).  Replaces the LAPACK potrf inside lstsq's role
// (regressors.py:155,165).
#pragma once
#include "nk_common.h"

namespace nk {

// ONE WAVE factorises a 64 x 64 diagonal block and inverts the factor.  The block lives in LDS; the work is arranged so
// that as little as possible of it is dependent scalar-style fp64 VALU code, because this wave usually shares its SIMD
// with MFMA waves of the square-root iteration (side stream) and every fp64 VALU FMA then queues behind a 64-cycle
// matrix instruction (measured: the earlier all-VALU kernel, 4032 dependent FMA issues, went from 52 us alone to 263 us
// beside the GEMM):
//   factor : four 16-column panels.  A panel is factorised with lane i owning row i (16 registers; column k of the other
//            rows fetched with v_readlane: 120 FMA issues per panel), then the trailing tiles get their rank-16 update
//            on the matrix pipe (v_mfma_f64_16x16x4, 10 tiles in total);
//   inverse: the four 16 x 16 diagonal blocks are inverted side by side (lane = block * 16 + column, forward
//            substitution, 120 FMA issues), the six blocks below them follow from  X_ij = -D_i sum_k L_ik X_kj  as
//            16 x 16 x 16 products on the matrix pipe.
// Blocks shorter than 64 are padded with the identity.
typedef double d4 __attribute__((ext_vector_type(4)));

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
  unsigned long long* piv[2];  // [min, max] pivot of the whole factorisation as bit patterns (positive doubles order like
                               // their bit patterns)
  double* plog[2];             // optional: every pivot of this block, in order (the numerical-rank verdict of
                               // cholesky_fail_flags looks for an isolated cluster of rounding-level pivots)
};

constexpr int PLD = CHOL_NB + 1;  // LDS row stride of the 64 x 64 images
constexpr int SLD = 17;           // LDS row stride of a 16 x 16 scratch tile

// acc += P Q for 16 x 16 operands in LDS (row strides ldp / ldq); NEG negates P.
// MFMA operand layout: a-lane l supplies P[l & 15][4 ks + (l >> 4)], b-lane l supplies Q[4 ks + (l >> 4)][l & 15];
// the accumulator lane holds rows (l >> 4) + 4 reg of column l & 15.
template <bool NEG>
__device__ __forceinline__ d4 mm16(const double* P, int ldp, const double* Q, int ldq, d4 acc, int l15, int l4) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    double a = P[l15 * ldp + 4 * ks + l4];
    const double b = Q[(4 * ks + l4) * ldq + l15];
    if (NEG) a = -a;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ void store_tile(double* T, int ldt, d4 acc, int l15, int l4) {
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) T[(l4 + 4 * reg) * ldt + l15] = acc[reg];
}

// `which`: the system of a paired factorisation this wave works on; `lane`: 0..63 within the wave (the wave may be one of
// several in its workgroup: chol_trail_potrf_kernel -- every __syncthreads below then only waits for the waves that are
// still alive, which is this one once the others have finished their tiles)
__device__ __forceinline__ void potrf_diag_kernel_body(const PotrfBatch& pb, int blk, int which, int lane) {
  constexpr int NB = CHOL_NB;
  static_assert(NB == 64, "one lane per row, four 16-column panels");
  const int nb = pb.nb[which];
  if (nb <= 0) return;
  // this wave is the critical path of the factorisation chain: take the issue slots first
  __builtin_amdgcn_s_setprio(3);
  double* __restrict__ A = pb.A[which];
  const int64_t lda = pb.lda[which];
  double* __restrict__ Linv = pb.Linv[which];
  int* __restrict__ info = pb.info[which];
  double* __restrict__ plog = pb.plog[which];
  double piv_min = 1.0e308, piv_max = 0.0;
  __shared__ double As[NB * PLD];
  __shared__ double Iv[NB * PLD];
  __shared__ double Sc[3 * 16 * SLD];
  __shared__ double dinv[NB];  // 1 / L_kk
  const int l15 = lane & 15, l4 = lane >> 4;
  // whole rows are loaded (the upper triangle is carried along but never consumed); rows / columns beyond nb are
  // identity padding
#pragma unroll 8
  for (int r = 0; r < NB; ++r) {
    double v = (r == lane) ? 1.0 : 0.0;
    if (r < nb && lane < nb) v = A[(int64_t)r * lda + lane];
    As[r * PLD + lane] = v;
  }
  __syncthreads();

  // ---- factor ---------------------------------------------------------------------------------------------------
#pragma unroll
  for (int pbk = 0; pbk < 4; ++pbk) {
    const int c0 = 16 * pbk;
    double R[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) R[jj] = As[lane * PLD + c0 + jj];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double dkk = readlane_f64(R[k], c0 + k);
      if (lane == 0 && plog != nullptr && c0 + k < nb) plog[c0 + k] = dkk;
      if (!(dkk > 0.0) || !isfinite(dkk)) {
        if (lane == 0 && c0 + k < nb) atomicCAS(info, 0, blk * NB + c0 + k + 1);
        dkk = 1.0;
      } else if (c0 + k < nb) {
        piv_min = fmin(piv_min, dkk);
        piv_max = fmax(piv_max, dkk);
      }
      // sqrt and reciprocal without the library routines (about 45 dependent fp64 VALU instructions per pivot between
      // them, more than the whole panel update): hardware rsq seed, two Newton steps, one Heron correction
      double inv = __builtin_amdgcn_rsq(dkk);
      inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
      inv = inv * fma(-0.5 * dkk * inv, inv, 1.5);
      double s = dkk * inv;
      s = fma(0.5 * inv, fma(-s, s, dkk), s);   // s = sqrt(dkk) to the last bit or two
      inv = fma(inv, fma(-s, inv, 1.0), inv);   // inv = 1 / s
      if (lane == c0 + k) dinv[c0 + k] = inv;
      R[k] = (lane == c0 + k) ? s : R[k] * inv;  // lanes > c0+k: L[i][c0+k]; lanes above hold unused upper-triangle values
      // R[jj] -= L[i][c0+k] * L[c0+jj][c0+k]; the second factor is lane c0+jj's R[k]: readlane -> SGPR pair -> scalar
      // operand of the FMA, kept in ONE asm statement (hipcc otherwise hoists the readlanes and spills the SGPRs)
      {
        const int rk_lo = __double2loint(R[k]), rk_hi = __double2hiint(R[k]);
#pragma unroll
        for (int jj = k + 1; jj < 16; ++jj)
          asm volatile("v_readlane_b32 s96, %1, %3\n\tv_readlane_b32 s97, %2, %3\n\ts_nop 0\n\tv_fma_f64 %0, -%4, s[96:97], %0"
                       : "+v"(R[jj])
                       : "v"(rk_lo), "v"(rk_hi), "i"(c0 + jj), "v"(R[k])
                       : "s96", "s97");
      }
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) As[lane * PLD + c0 + jj] = R[jj];
    __syncthreads();
    // rank-16 update of the trailing lower tiles on the matrix pipe
#pragma unroll
    for (int ti = pbk + 1; ti < 4; ++ti)
#pragma unroll
      for (int tj = pbk + 1; tj <= ti; ++tj) {
        d4 c;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) c[reg] = As[(16 * ti + l4 + 4 * reg) * PLD + 16 * tj + l15];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const double a = -As[(16 * ti + l15) * PLD + c0 + 4 * ks + l4];
          const double b = As[(16 * tj + l15) * PLD + c0 + 4 * ks + l4];
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) As[(16 * ti + l4 + 4 * reg) * PLD + 16 * tj + l15] = c[reg];
      }
    __syncthreads();
  }
  // L -> global (rows go back whole: nothing reads the upper triangle of a factored diagonal block)
#pragma unroll 8
  for (int r = 0; r < NB; ++r)
    if (r < nb && lane < nb) A[(int64_t)r * lda + lane] = As[r * PLD + lane];

  // ---- inverse ---------------------------------------------------------------------------------------------------
  {
    // diagonal blocks: lane = (block l4, column l15)
    const double* Lb = As + (16 * l4) * PLD + 16 * l4;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double s0 = (i == l15) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const double l = Lb[i * PLD + j];
        if ((j & 1) == 0) s0 = fma(-l, x[j], s0);
        else s1 = fma(-l, x[j], s1);
      }
      x[i] = (s0 + s1) * dinv[16 * l4 + i];
    }
    double* Db = Iv + (16 * l4) * PLD + 16 * l4;
#pragma unroll
    for (int i = 0; i < 16; ++i) Db[i * PLD + l15] = x[i];
  }
  __syncthreads();
  auto Lblk = [&](int i, int j) { return As + (16 * i) * PLD + 16 * j; };
  auto Xblk = [&](int i, int j) { return Iv + (16 * i) * PLD + 16 * j; };
  const d4 zero = d4{0.0, 0.0, 0.0, 0.0};
  // first block sub-diagonal: X_{i,i-1} = -D_i (L_{i,i-1} D_{i-1})
#pragma unroll
  for (int i = 1; i < 4; ++i)
    store_tile(Sc + (i - 1) * 16 * SLD, SLD, mm16<false>(Lblk(i, i - 1), PLD, Xblk(i - 1, i - 1), PLD, zero, l15, l4), l15, l4);
  __syncthreads();
#pragma unroll
  for (int i = 1; i < 4; ++i)
    store_tile(Xblk(i, i - 1), PLD, mm16<true>(Xblk(i, i), PLD, Sc + (i - 1) * 16 * SLD, SLD, zero, l15, l4), l15, l4);
  __syncthreads();
  // second: X_{i,i-2} = -D_i (L_{i,i-2} D_{i-2} + L_{i,i-1} X_{i-1,i-2})
#pragma unroll
  for (int i = 2; i < 4; ++i) {
    d4 t = mm16<false>(Lblk(i, i - 2), PLD, Xblk(i - 2, i - 2), PLD, zero, l15, l4);
    t = mm16<false>(Lblk(i, i - 1), PLD, Xblk(i - 1, i - 2), PLD, t, l15, l4);
    store_tile(Sc + (i - 2) * 16 * SLD, SLD, t, l15, l4);
  }
  __syncthreads();
#pragma unroll
  for (int i = 2; i < 4; ++i)
    store_tile(Xblk(i, i - 2), PLD, mm16<true>(Xblk(i, i), PLD, Sc + (i - 2) * 16 * SLD, SLD, zero, l15, l4), l15, l4);
  __syncthreads();
  // third: X_30 = -D_3 (L_30 D_0 + L_31 X_10 + L_32 X_20)
  {
    d4 t = mm16<false>(Lblk(3, 0), PLD, Xblk(0, 0), PLD, zero, l15, l4);
    t = mm16<false>(Lblk(3, 1), PLD, Xblk(1, 0), PLD, t, l15, l4);
    t = mm16<false>(Lblk(3, 2), PLD, Xblk(2, 0), PLD, t, l15, l4);
    store_tile(Sc, SLD, t, l15, l4);
  }
  __syncthreads();
  store_tile(Xblk(3, 0), PLD, mm16<true>(Xblk(3, 3), PLD, Sc, SLD, zero, l15, l4), l15, l4);
  __syncthreads();
  // dense 64 x 64 row-major inverse, exact zeros above the block diagonal
#pragma unroll 8
  for (int i = 0; i < NB; ++i) Linv[i * NB + lane] = (l4 <= (i >> 4)) ? Iv[i * PLD + lane] : 0.0;
  // ... followed by the factor block itself in the same dense form (CHOL_WS): the correction step of every product with the
  // inverse reads it (the copy in the matrix carries whatever the upper triangle held)
  // ... and by the verdict whether products with this inverse need the correction step at all: the error of such a product
  // grows with cond(L_jj) <= ||L_jj||_F ||L_jj^-1||_F.  Most diagonal blocks of a regularised kernel system are nearly
  // scalar (at the headline shape 62 of 64 have ||.||_F ||.||_F / 64 < 2.2; the first and the last, which holds the input
  // columns, 40 and 1000), so the correction runs where it is needed: above 8 x 64 (CHOL_FIX_KAPPA).
  double sl = 0.0, si = 0.0;
#pragma unroll 8
  for (int i = 0; i < NB; ++i) {
    const double vl = (lane <= i) ? As[i * PLD + lane] : 0.0;
    const double vi = (l4 <= (i >> 4)) ? Iv[i * PLD + lane] : 0.0;
    Linv[NB * NB + i * NB + lane] = vl;
    sl = fma(vl, vl, sl);
    si = fma(vi, vi, si);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sl += __shfl_xor(sl, off, 64);
    si += __shfl_xor(si, off, 64);
  }
  if (lane == 0) Linv[2 * NB * NB] = (sl * si > CHOL_FIX_KAPPA * CHOL_FIX_KAPPA) ? 1.0 : 0.0;  // NaN -> 0: a failed block
  if (lane == 0 && piv_max > 0.0) {
    atomicMin(pb.piv[which], (unsigned long long)__double_as_longlong(piv_min));
    atomicMax(pb.piv[which] + 1, (unsigned long long)__double_as_longlong(piv_max));
  }
}
}  // namespace nk
