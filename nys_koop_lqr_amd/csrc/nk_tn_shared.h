// Shared between the fp64 TN engine (nk_gemm_tn.hip) and the fp32 one (nk_gemm_tn_f32.hip): the 128 x 128 tile walk and the
// deterministic split-K reduce (partial tiles are fp64 in both engines).
#pragma once
#include "nk_common.h"

namespace nk {

constexpr int TBM = 128;
constexpr int TN_MAXP = 4;
constexpr int RPARTS = 8;  // workgroups per tile in the reduce kernel

// tile (tm, tn) of index t within a problem.  Full problems whose tile grid is a multiple of 8 x 8 are walked in 8 x 8
// super-blocks: workgroups are dispatched in index order, so the ~64 tiles an XCD holds at a time then stream 8 + 8
// operand panels instead of 4 + 16, and more of the panel traffic is shared through that XCD's L2.
__device__ __forceinline__ void tn_tile_coords(int t, int tri, int tiles_n, int M, int& tm, int& tn) {
  if (tri == TRI_FULL) {
    const int tiles_m = (M + TBM - 1) / TBM;
    if ((tiles_m & 7) == 0 && (tiles_n & 7) == 0) {
      const int sb = t >> 6, w = t & 63;
      const int sbn = tiles_n >> 3;
      const int sbr = sb / sbn, sbc = sb - sbr * sbn;
      tm = sbr * 8 + (w >> 3);
      tn = sbc * 8 + (w & 7);
    } else {
      tm = t / tiles_n;
      tn = t - tm * tiles_n;
    }
  } else if ((tiles_n & 7) == 0) {
    // upper triangle in 8 x 8 super-blocks (I <= J, row-major over the super-blocks): 36 tiles in a diagonal super-block
    // (its own upper triangle, row-major), 64 in the others
    const int S = tiles_n >> 3;
    int I = 0, J = 0, rem = t;
    for (;;) {
      const int cnt = (I == J) ? 36 : 64;
      if (rem < cnt) break;
      rem -= cnt;
      if (++J == S) { ++I; J = I; }
    }
    int r, c;
    if (I == J) {
      r = 0;
      while (rem >= 8 - r) {
        rem -= 8 - r;
        ++r;
      }
      c = r + rem;
    } else {
      r = rem >> 3;
      c = rem & 7;
    }
    tm = I * 8 + r;
    tn = J * 8 + c;
  } else {  // upper triangle, row-major enumeration
    int row = 0, rem = t;
    while (rem >= tiles_n - row) {
      rem -= tiles_n - row;
      ++row;
    }
    tm = row;
    tn = row + rem;
  }
}

struct TnRed {
  double* C;
  double* Ct;
  double* Caff;  // see TnDev
  double aff_a, aff_c;
  int64_t ldc, ldct;
  int M, N, tiles_n, tri, tile_begin;
  double alpha, beta;
};
struct TnRedParams {
  TnRed p[TN_MAXP];
  int nprob, splitk;
  const double* slab;
  const double* skip_state;  // see TnParams
  int skip_step;
  // optional (single symmetric problem): every workgroup leaves sum (C - I)^2 over its band in resid_partials[block]
  // (mirrored tiles counted twice), for a convergence check of ||C - I||_F without a separate pass over C
  double* resid_partials;
};

int launch_tn_reduce(nk_ctx* ctx, const TnRedParams& R, int ntiles);  // sums the K slices of a slab in a fixed order
int tn_ensure_zero_page(nk_ctx* ctx);

}  // namespace nk
