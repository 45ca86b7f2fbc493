// Backward substitution of the augmented blocked Cholesky (nk_linalg.hip: cholesky_aug_pair_async) in ONE launch.
//
// After the factorisation the extra rows hold E = R^T L^-T; the solution of the regularised systems of
// regressors.py:155,165 is X^T = E L^-1, i.e. every ROW x of the result solves x L = e on its own.  The blocked
// right-looking form (per 64-column block: one small product with the inverted diagonal block, one rank-64 update of the
// columns to the left) needs 2 launches per block -- 64 dependent, latency-bound launches at m = 2000.  Because the rows
// are independent, one workgroup can instead carry a band of 16 rows through the whole substitution alone, in the
// left-looking order
//     for j = last block .. 0:   X_j = (E_j - sum_{i > j} X_i L[i, j]) L_jj^-1
// with the accumulation on the matrix pipe (v_mfma_f64_16x16x4): the band blocks X_i go through LDS (one coalesced
// cooperative load serves the four waves), the blocks of the factor straight into registers (the factor is read by
// every band from L2; 16 MB at m = 2000), the next block pair is fetched while the current one is multiplied, and
// nothing but the finished X_j is written.  125 + 24 workgroups at the C4 shape.
#include "nk_common.h"

namespace nk {

typedef double d4 __attribute__((ext_vector_type(4)));

struct TrsmSys {
  double* E;           // rows x m band matrix, overwritten by X
  int64_t lde;
  int rows;
  const double* L;     // m x m lower factor
  int64_t ldl;
  int m;
  const double* Dinv;  // inverted 64 x 64 diagonal blocks, dense row-major, identity padded
  int wg_begin;        // first workgroup of this system
  int fix;             // correction step of the diagonal-block solves: 0 never, 1 by the block's verdict word, 2 always
};
struct TrsmBatch {
  TrsmSys s[2];
  int nsys;
};

constexpr int TR_RT = 1;              // 16-row MFMA tiles per wave
constexpr int TR_ROWS = 16 * TR_RT;  // rows per workgroup; wave w owns columns 16w..16w+15 of the current block
constexpr int TR_PA = TR_ROWS / 4;    // doubles per thread of a cooperative band-block load (256 threads, 64 columns)
constexpr int TLD = 65;      // odd row stride: the 16 rows an operand fetch touches fall into distinct LDS banks

__device__ __forceinline__ void trsm_right_lower_kernel_body(const TrsmBatch& tb) {
  constexpr int NB = CHOL_NB;
  const int q = (tb.nsys > 1 && (int)blockIdx.x >= tb.s[1].wg_begin) ? 1 : 0;
  const TrsmSys s = tb.s[q];
  const int r0 = ((int)blockIdx.x - s.wg_begin) * TR_ROWS;
  if (r0 >= s.rows) return;
  __builtin_amdgcn_s_setprio(2);
  __shared__ double Xs[2][TR_ROWS * TLD];  // X_i band blocks (a-operand of the accumulation), double buffered
  __shared__ double T[TR_ROWS * TLD];      // accumulator on its way to becoming an a-operand
  __shared__ double O[TR_ROWS * TLD];      // X_j before its correction step, likewise
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nblk = (s.m + NB - 1) / NB;
  const int m = s.m, rows = s.rows;
  double* __restrict__ E = s.E;
  const double* __restrict__ L = s.L;
  const int64_t lde = s.lde, ldl = s.ldl;
  // cooperative band-block load: thread t moves TR_PA consecutive doubles of one row (rows past the end are clamped:
  // they only feed accumulator rows that are never stored)
  constexpr int TPR = 64 / TR_PA;  // threads per row
  const int xr = threadIdx.x / TPR, xc = (threadIdx.x % TPR) * TR_PA;
  const double* xsrc = E + (int64_t)min(r0 + xr, rows - 1) * lde;

  for (int j = nblk - 1; j >= 0; --j) {
    const int j0 = j * NB;
    const int colw = j0 + 16 * w + l15;  // this lane's output column (b-operand column and accumulator column)
    const bool col_ok = colw < m;
    const int colc = min(colw, m - 1);
    d4 acc[TR_RT];
#pragma unroll
    for (int rt = 0; rt < TR_RT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = r0 + 16 * rt + l4 + 4 * reg;
        acc[rt][reg] = (row < rows && col_ok) ? -E[(int64_t)row * lde + colw] : 0.0;  // acc = sum_i X_i L_ij - E_j
      }
    // Block pair (i, j): A = X_i (32 x 64, through LDS, shared by the four waves), B = L[i-block, j-block] (64 x 64,
    // straight into registers: each wave needs its own 16 columns).  The contraction index is dealt so that lane group
    // l4 owns k = 16 l4 .. 16 l4 + 15 of the block (any assignment works as long as both operands use the same one).
    // A k range that runs past m (last block only) is zeroed through the b-operand.
    // The blocks of the factor are cold (every block is used once per band and all bands advance in step, so each pair
    // starts with an L2 miss of 2-3 us, three times the 0.85 us the matrix pipe needs per pair): they are fetched THREE
    // pairs ahead into a ring of four register sets (no copies: the loop body is unrolled over the ring), and so are
    // the band blocks (32 MB of finished X at m = 2000: long evicted from the 4 MB L2 when they are needed again).
    double b0[16], b1[16], b2[16], b3[16], pa0[TR_PA], pa1[TR_PA], pa2[TR_PA], pa3[TR_PA];
    auto gload_b = [&](int i, double (&B)[16]) {
      const int kb = i * NB + 16 * l4;
      if (i * NB + NB <= m) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) B[ks] = L[(int64_t)(kb + ks) * ldl + colc];
      } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int k = kb + ks;
          const double vb = L[(int64_t)min(k, m - 1) * ldl + colc];
          B[ks] = k < m ? vb : 0.0;
        }
      }
    };
    auto gload_a = [&](int i, double (&PA)[TR_PA]) {
      const int i0 = i * NB;
      if (i0 + NB <= m) {
#pragma unroll
        for (int c = 0; c < TR_PA; ++c) PA[c] = xsrc[i0 + xc + c];
      } else {
#pragma unroll
        for (int c = 0; c < TR_PA; ++c) PA[c] = xsrc[min(i0 + xc + c, m - 1)];
      }
    };
    auto sstore = [&](int buf, const double (&PA)[TR_PA]) {
#pragma unroll
      for (int c = 0; c < TR_PA; ++c) Xs[buf][xr * TLD + xc + c] = PA[c];
    };
    // One pair.  Ring slot of pair i is (i - j - 1) mod 4 for both operands: multiply with Bcur (band block i is already
    // in LDS), refill the slot used one pair ago (Bfree) with factor block i + 3, refill this pair's band slot (Acur, its
    // block went to LDS one pair ago) with band block i + 4, and move band block i + 1 (Anext) to LDS at the end.
    auto step = [&](int i, const double (&Bcur)[16], double (&Bfree)[16], double (&Acur)[TR_PA],
                    const double (&Anext)[TR_PA]) {
      const int cur = (i - j - 1) & 1;
      if (i + 3 < nblk) gload_b(i + 3, Bfree);
      if (i + 4 < nblk) gload_a(i + 4, Acur);
      const double* x0 = Xs[cur] + l15 * TLD + 16 * l4;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int rt = 0; rt < TR_RT; ++rt)
          acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[rt * 16 * TLD + ks], Bcur[ks], acc[rt], 0, 0, 0);
      if (i + 1 < nblk) sstore(cur ^ 1, Anext);
      __syncthreads();
    };
    if (j + 1 < nblk) {
      gload_a(j + 1, pa0);
      gload_b(j + 1, b0);
      if (j + 2 < nblk) { gload_a(j + 2, pa1); gload_b(j + 2, b1); }
      if (j + 3 < nblk) { gload_a(j + 3, pa2); gload_b(j + 3, b2); }
      if (j + 4 < nblk) gload_a(j + 4, pa3);
      sstore(0, pa0);
    }
    __syncthreads();
    for (int i = j + 1; i < nblk; i += 4) {
      step(i, b0, b3, pa0, pa1);
      if (i + 1 < nblk) step(i + 1, b1, b0, pa1, pa2);
      if (i + 2 < nblk) step(i + 2, b2, b1, pa2, pa3);
      if (i + 3 < nblk) step(i + 3, b3, b2, pa3, pa0);
    }
    // X_j = -acc * L_jj^-1: the accumulator goes through LDS to become an a-operand
#pragma unroll
    for (int rt = 0; rt < TR_RT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) T[(16 * rt + l4 + 4 * reg) * TLD + 16 * w + l15] = col_ok ? acc[rt][reg] : 0.0;
    __syncthreads();
    const double* __restrict__ D = s.Dinv + (size_t)j * CHOL_WS;
    const double* __restrict__ Ld = D + NB * NB;  // the factor's diagonal block, dense (CHOL_WS)
    d4 out[TR_RT];
#pragma unroll
    for (int rt = 0; rt < TR_RT; ++rt) out[rt] = d4{0.0, 0.0, 0.0, 0.0};
    double bd[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) bd[ks] = -D[(4 * ks + l4) * NB + 16 * w + l15];  // X_j = -acc L_jj^-1
    // (two accumulator chains per product below: a dependent fp64 matrix instruction waits ~200 cycles for its predecessor)
    {
      d4 o2[TR_RT];
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt) o2[rt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2)
#pragma unroll
        for (int rt = 0; rt < TR_RT; ++rt) {
          out[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(16 * rt + l15) * TLD + 4 * ks + l4], bd[ks], out[rt], 0, 0, 0);
          o2[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(16 * rt + l15) * TLD + 4 * (ks + 1) + l4], bd[ks + 1], o2[rt], 0, 0, 0);
        }
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt) out[rt] += o2[rt];
    }
    // One correction step from the data (see chol_panel_kernel: the product with the explicit inverse alone is not backward
    // stable):  U = acc + X_j L_jj  (the negated residual of X_j L_jj = -acc),  X_j <- X_j - U L_jj^-1.  X_j goes through LDS
    // to become an a-operand; U takes the place of the accumulator image in T.
    if (s.fix == 2 || (s.fix == 1 && D[2 * NB * NB] != 0.0)) {  // (uniform: one word per block)
#pragma unroll
    for (int rt = 0; rt < TR_RT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) O[(16 * rt + l4 + 4 * reg) * TLD + 16 * w + l15] = out[rt][reg];
    __syncthreads();  // (also: every wave is done reading T)
    {
      d4 u[TR_RT];
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) u[rt][reg] = col_ok ? acc[rt][reg] : 0.0;
      d4 u2[TR_RT];
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt) u2[rt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2) {
        const double bl0 = Ld[(4 * ks + l4) * NB + 16 * w + l15], bl1 = Ld[(4 * (ks + 1) + l4) * NB + 16 * w + l15];
#pragma unroll
        for (int rt = 0; rt < TR_RT; ++rt) {
          u[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(O[(16 * rt + l15) * TLD + 4 * ks + l4], bl0, u[rt], 0, 0, 0);
          u2[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(O[(16 * rt + l15) * TLD + 4 * (ks + 1) + l4], bl1, u2[rt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt) u[rt] += u2[rt];
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) T[(16 * rt + l4 + 4 * reg) * TLD + 16 * w + l15] = u[rt][reg];
    }
    __syncthreads();
    {
      d4 o2[TR_RT];
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt) o2[rt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2)
#pragma unroll
        for (int rt = 0; rt < TR_RT; ++rt) {
          out[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(16 * rt + l15) * TLD + 4 * ks + l4], bd[ks], out[rt], 0, 0, 0);
          o2[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(T[(16 * rt + l15) * TLD + 4 * (ks + 1) + l4], bd[ks + 1], o2[rt], 0, 0, 0);
        }
#pragma unroll
      for (int rt = 0; rt < TR_RT; ++rt) out[rt] += o2[rt];
    }
    }
#pragma unroll
    for (int rt = 0; rt < TR_RT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = r0 + 16 * rt + l4 + 4 * reg;
        if (row < rows && col_ok) E[(int64_t)row * lde + colw] = out[rt][reg];
      }
    // the finished block is read back (as X_i) by the whole workgroup in the following steps
    __threadfence_block();
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) trsm_right_lower_kernel(TrsmBatch tb) { trsm_right_lower_kernel_body(tb); }
__global__ void __launch_bounds__(256) trsm_right_lower_kernel_batched(const nk::ArgPack<TrsmBatch>* table) {
  trsm_right_lower_kernel_body(table[blockIdx.z].v);
}
static nk::TwinReg trsm_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(TrsmBatch)>(trsm_right_lower_kernel)),
                                 reinterpret_cast<const void*>(trsm_right_lower_kernel_batched), sizeof(nk::ArgPack<TrsmBatch>),
                                 "trsm_right_lower_kernel");

// E_q <- E_q L_q^-1 for up to two systems (E: extra x m rows below the factor in the same array, see CholSys)
int launch_trsm_right_lower_pair(nk_ctx* ctx, const CholSys* sys, int nsys) {
  TrsmBatch tb;
  tb.nsys = 0;
  int wgs = 0;
  for (int q = 0; q < nsys && q < 2; ++q) {
    const CholSys& y = sys[q];
    if (y.extra <= 0 || !y.backward) continue;
    TrsmSys& t = tb.s[tb.nsys++];
    t.E = y.P + (int64_t)y.m * y.ldp; t.lde = y.ldp; t.rows = y.extra;
    t.L = y.P; t.ldl = y.ldp; t.m = y.m; t.Dinv = y.Linv; t.wg_begin = wgs; t.fix = chol_fix_enabled();
    wgs += (y.extra + TR_ROWS - 1) / TR_ROWS;
  }
  if (tb.nsys == 0) return NK_OK;
  if (tb.nsys == 1) tb.s[1] = tb.s[0];
  hipLaunchKernelGGL(trsm_right_lower_kernel, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, tb);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
