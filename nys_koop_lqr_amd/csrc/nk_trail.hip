// Trailing update of one step of the blocked Cholesky (nk_linalg.hip: cholesky_aug_pair_async):
//     C[tm, tn] -= P[tm] P[tn]^T      P = the 64-column panel of this step (rows x 64, row-major, k contiguous),
// lower tiles of the square part plus full tiles for the extra (right-hand-side) rows, for up to two systems in one
// launch.  The generic engine (nk_gemm.hip) walks K in register-staged, barrier-separated steps of 16 -- with K = 64 it
// never reaches a steady state (50 us per launch alone, ~90 us beside the square-root GEMMs, 32 launches on the critical
// path of a fit).  Here a workgroup takes both 128 x 64 panel blocks of its tile straight from global memory into
// registers in two halves of K (each lane owns 8 contiguous doubles of its rows: the contraction index is dealt to the
// lane groups, any assignment works as long as both operands use the same one), multiplies on the matrix pipe
// (4 waves x 4 x 4 tiles of v_mfma_f64_16x16x4) and read-modify-writes the 128 x 128 tile of C.  No LDS, no barriers.
#include "nk_common.h"
#include "nk_potrf_body.h"

namespace nk {


struct TrailSys {
  const double* P;  // panel: rows x 64
  int64_t ldp;
  double* C;        // trailing matrix: rows x rem (lower tiles of the leading rem x rem part + all tiles below it)
  int64_t ldc;
  int rows, rem;
  int tiles_n;      // ceil(rem / 128)
  int nblocks;
};
struct TrailBatch {
  TrailSys s[2];
};

__device__ __forceinline__ void chol_trail_kernel_body(const TrailBatch& tb) {
  const TrailSys s = tb.s[blockIdx.y];
  if ((int)blockIdx.x >= s.nblocks) return;
  __builtin_amdgcn_s_setprio(2);  // latency-bound chain beside the GEMM-bound side stream
  // tile (tm, tn): tile row r of the lower set holds min(r + 1, tiles_n) tiles (rows past the square part are full)
  int tm = 0, tn = 0;
  {
    int rem = blockIdx.x;
    for (;;) {
      const int cnt = tm + 1 < s.tiles_n ? tm + 1 : s.tiles_n;
      if (rem < cnt) break;
      rem -= cnt;
      ++tm;
    }
    tn = rem;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int m0 = tm * 128 + wm * 64, n0 = tn * 128 + wn * 64;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  // operand rows of this lane (clamped: rows past the end only feed accumulator entries that are never stored)
  const double* pa[4];
  const double* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pa[i] = s.P + (int64_t)min(m0 + 16 * i + l15, s.rows - 1) * s.ldp + 8 * l4;
    pb[i] = s.P + (int64_t)min(n0 + 16 * i + l15, s.rem - 1) * s.ldp + 8 * l4;  // column c of C <-> panel row c
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // k in [32 h, 32 h + 32): lane group l4 owns k = 32 h + 8 l4 .. + 7
    double a[4][8], b[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        a[i][ks] = pa[i][32 * h + ks];
        b[i][ks] = pb[i][32 * h + ks];
      }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][ks], b[j][ks], acc[i][j], 0, 0, 0);
  }
  // C -= acc: lane holds rows (l4 + 4 reg) of column l15 of each 16 x 16 tile
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = m0 + 16 * i + l4 + 4 * reg;
        const int col = n0 + 16 * j + l15;
        if (row < s.rows && col < s.rem) {
          double* c = s.C + (int64_t)row * s.ldc + col;
          *c -= acc[i][j][reg];
        }
      }
}
__global__ void __launch_bounds__(256) chol_trail_kernel(TrailBatch tb) { chol_trail_kernel_body(tb); }
__global__ void __launch_bounds__(256) chol_trail_kernel_batched(const nk::ArgPack<TrailBatch>* table) { chol_trail_kernel_body(table[blockIdx.z].v); }
static nk::TwinReg chol_trail_kernel_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(TrailBatch)>(chol_trail_kernel)),
                                 reinterpret_cast<const void*>(chol_trail_kernel_batched), sizeof(nk::ArgPack<TrailBatch>), "chol_trail_kernel");

// trailing updates of up to two systems; calls[q] as prepared for the generic engine (A = B = panel, K = 64, alpha = -1,
// beta = 1, TRI_LOWER).  Returns false when a call does not have that shape (the caller then uses the generic engine).
bool launch_chol_trail_pair(nk_ctx* ctx, const GemmCall* calls, int ncalls, int* rc);

// ---------------------------------------------------------------------------------------------------------------
// Trailing update of step j AND the diagonal-block factorisation of step j + 1 in one launch.  The next diagonal block is the
// 64 x 64 sub-tile of wave 0 of workgroup 0: as soon as that wave has written it back, it factors and inverts it
// (potrf_diag_kernel_body, one wave) while the other workgroups are still on their tiles -- the ~30 us of the diagonal
// kernel leave the critical path wherever the trailing update takes as long, and every block step loses one of its three
// kernel boundaries.  Same arithmetic in the same order as the two separate launches: bit-identical results
// (NYSKOOP_CHOL_FUSE=0 issues them separately).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void chol_trail_potrf_kernel_body(const TrailBatch& tb, const PotrfBatch& pb, int blk) {
  chol_trail_kernel_body(tb);
  if (blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0 && pb.nb[blockIdx.y] > 0) {
    __threadfence();  // this wave's own stores of the block are visible to its loads (another lane layout) below
    potrf_diag_kernel_body(pb, blk, blockIdx.y, threadIdx.x & 63);
  }
}
__global__ void __launch_bounds__(256) chol_trail_potrf_kernel(TrailBatch tb, PotrfBatch pb, int blk) { chol_trail_potrf_kernel_body(tb, pb, blk); }
__global__ void __launch_bounds__(256) chol_trail_potrf_kernel_batched(const nk::ArgPack<TrailBatch, PotrfBatch, int>* table) {
  chol_trail_potrf_kernel_body(table[blockIdx.z].v, table[blockIdx.z].rest.v, table[blockIdx.z].rest.rest.v);
}
static nk::TwinReg chol_trail_potrf_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(TrailBatch, PotrfBatch, int)>(chol_trail_potrf_kernel)),
                                             reinterpret_cast<const void*>(chol_trail_potrf_kernel_batched),
                                             sizeof(nk::ArgPack<TrailBatch, PotrfBatch, int>), "chol_trail_potrf_kernel");

static bool fill_trail_batch(const GemmCall* calls, int ncalls, TrailBatch& tb, int& maxblocks) {
  maxblocks = 0;
  for (int q = 0; q < 2; ++q) {
    TrailSys& t = tb.s[q];
    t = TrailSys{};
    if (q >= ncalls || calls[q].M <= 0 || calls[q].N <= 0) continue;
    const GemmCall& c = calls[q];
    if (c.K != 64 || c.A != c.B || c.lda != c.ldb || c.alpha != -1.0 || c.beta != 1.0 || c.opts.tri != TRI_LOWER ||
        c.M < c.N)
      return false;
    t.P = c.A; t.ldp = c.lda; t.C = c.C; t.ldc = c.ldc; t.rows = (int)c.M; t.rem = (int)c.N;
    t.tiles_n = (t.rem + 127) / 128;
    const int tiles_m = (t.rows + 127) / 128;
    t.nblocks = t.tiles_n * (t.tiles_n + 1) / 2 + (tiles_m - t.tiles_n) * t.tiles_n;
    if (t.nblocks > maxblocks) maxblocks = t.nblocks;
  }
  return true;
}

bool launch_chol_trail_pair(nk_ctx* ctx, const GemmCall* calls, int ncalls, int* rc) {
  *rc = NK_OK;
  TrailBatch tb;
  int maxblocks = 0;
  if (!fill_trail_batch(calls, ncalls, tb, maxblocks)) return false;
  if (maxblocks == 0) return true;
  hipLaunchKernelGGL(chol_trail_kernel, dim3((unsigned)maxblocks, 2), dim3(256), 0, ctx->stream, tb);
  if (hipGetLastError() != hipSuccess) {
    set_error("chol_trail launch failed");
    *rc = NK_ERR_HIP;
  }
  return true;
}

// trailing updates of step `blk - 1` (calls[q] as for launch_chol_trail_pair) fused with the diagonal-block factorisation of
// step `blk` of the same systems (nb[q] = order of that block, 0: the system has none; its block is calls[q].C).  false: not
// that shape, nothing launched.
bool launch_chol_trail_potrf_pair(nk_ctx* ctx, const GemmCall* calls, int ncalls, const int* nb, double* const* Linv, int blk,
                                  double* const* plog, int* rc) {
  *rc = NK_OK;
  TrailBatch tb;
  int maxblocks = 0;
  if (!fill_trail_batch(calls, ncalls, tb, maxblocks) || maxblocks == 0) return false;
  PotrfBatch pb;
  for (int q = 0; q < 2; ++q) {
    const bool on = q < ncalls && nb[q] > 0 && tb.s[q].nblocks > 0;
    if (q < ncalls && nb[q] > 0 && tb.s[q].nblocks == 0) return false;  // a diagonal block without a trailing update before it
    pb.A[q] = on ? tb.s[q].C : nullptr;
    pb.lda[q] = on ? tb.s[q].ldc : 0;
    pb.nb[q] = on ? nb[q] : 0;
    pb.Linv[q] = on ? Linv[q] : nullptr;
    pb.info[q] = ctx->d_info + info_base(ctx) + q;
    pb.piv[q] = ctx->d_piv + 2 * (info_base(ctx) + q);
    pb.plog[q] = (on && plog) ? plog[q] : nullptr;
  }
  hipLaunchKernelGGL(chol_trail_potrf_kernel, dim3((unsigned)maxblocks, 2), dim3(256), 0, ctx->stream, tb, pb, blk);
  if (hipGetLastError() != hipSuccess) {
    set_error("chol_trail_potrf launch failed");
    *rc = NK_ERR_HIP;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Panel solve of the same step:  P <- P L_jj^-T  (rows x 64 against the 64 x 64 diagonal block; in place: a workgroup reads
// exactly the 64 rows it writes, and only after all of them have been read).  Register-only like the trailing update.
//
// The product with the explicitly inverted block, P1 = P Linv^T, is what makes this step a GEMM -- and on its own it is not
// backward stable: its error grows with cond(L_jj) (2e6 on the first block of config 2's system), and through it the whole
// blocked solve carried a backward error of 6 eps where LAPACK's substitution leaves 0.3 eps (tools/illcond_diag.py; the
// same numbers from a NumPy emulation of this algorithm).  One correction step from the DATA restores it,
//     P2 = P1 + (P - P1 L_jj^T) Linv^T,
// three 64-wide products instead of one.  All three run on the matrix pipe in the TRANSPOSED form (D = Linv P^T: output
// column = panel row), because in that form the accumulator of one product is, register for register, the b-operand of the
// next: lane (l15, l4) of an accumulator holds P1[r(l15)][4 t + l4], t = 4 j + reg, and with the contraction index dealt as
// k = 4 t + l4 that is exactly the b-operand of instruction t -- no LDS, no lane exchange.  The raw panel rows are loaded in
// the same dealing and serve both as b-operand of the first product and as the minuend of the residual.
// Columns beyond `nb` (a short last block; only the right-hand-side rows of an augmented system get there) are masked.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PANEL_ROWS = 64;  // rows per workgroup of the panel solve
struct PanelSys {
  double* P;
  int64_t ldp;
  const double* Linv;  // CHOL_WS doubles: 64 x 64 inverse, then the 64 x 64 factor block (dense, identity padded)
  int rows;
  int nb;              // valid columns of the block
  int fix;             // correction step: 0 never, 1 where the block's verdict word says so, 2 always
  int nblocks;
};
struct PanelBatch {
  PanelSys s[2];
};

__device__ __forceinline__ void chol_panel_kernel_body(const PanelBatch& pb) {
  const PanelSys s = pb.s[blockIdx.y];
  if ((int)blockIdx.x >= s.nblocks) return;
  __builtin_amdgcn_s_setprio(2);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  // 16 rows per wave, 64 per workgroup: the launch is latency bound (at most 63 workgroups on 256 CUs), so the rows are
  // spread as thin as the 16-row matrix instruction allows
  const int m0 = blockIdx.x * PANEL_ROWS + wave * 16;
  const int nb = s.nb;
  const double* __restrict__ Li = s.Linv;
  const double* __restrict__ Ld = s.Linv + CHOL_NB * CHOL_NB;
  // raw rows: a[t] = P[r][4 t + l4]
  double a[16];
  {
    const double* pr = s.P + (int64_t)min(m0 + l15, s.rows - 1) * s.ldp + l4;
    if (nb == CHOL_NB) {
#pragma unroll
      for (int t = 0; t < 16; ++t) a[t] = pr[4 * t];
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) a[t] = (4 * t + l4 < nb) ? pr[4 * t] : 0.0;
    }
  }
  // a-operands: rows of the inverse, k = 4 t + l4
  double li[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 16; ++t) li[j][t] = Li[(16 * j + l15) * CHOL_NB + 4 * t + l4];
  // 1.  P1 = P Linv^T
  // (t outer, j inner everywhere: four independent accumulator chains in flight -- a dependent fp64 matrix instruction waits
  // ~200 cycles for its predecessor, three times its issue interval)
  d4 p1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) p1[j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) p1[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(li[j][t], a[t], p1[j], 0, 0, 0);
  if (s.fix == 2 || (s.fix == 1 && s.Linv[2 * CHOL_NB * CHOL_NB] != 0.0)) {  // (uniform: one word per block)
    // 2.  Rsd = P - P1 L_jj^T   (a-operand: minus the rows of the factor block; the accumulator starts from the raw rows)
    d4 rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rs[j] = d4{a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]};
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        rs[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ld[(16 * j + l15) * CHOL_NB + 4 * t + l4], p1[t >> 2][t & 3], rs[j], 0, 0, 0);
    // 3.  P2 = P1 + Rsd Linv^T
    d4 p2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p2[j] = p1[j];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) p2[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(li[j][t], rs[t >> 2][t & 3], p2[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) p1[j] = p2[j];
  }
  // every wave of the workgroup must have its raw rows in registers before any wave overwrites (waves own disjoint
  // rows, so this is only needed against the clamped loads of the last, partial tile)
  __syncthreads();
  const int row = m0 + l15;
  if (row < s.rows) {
    double* pw = s.P + (int64_t)row * s.ldp + l4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        if (nb == CHOL_NB || 16 * j + 4 * reg + l4 < nb) pw[16 * j + 4 * reg] = p1[j][reg];
  }
}
__global__ void __launch_bounds__(256) chol_panel_kernel(PanelBatch pb) { chol_panel_kernel_body(pb); }
__global__ void __launch_bounds__(256) chol_panel_kernel_batched(const nk::ArgPack<PanelBatch>* table) { chol_panel_kernel_body(table[blockIdx.z].v); }
static nk::TwinReg chol_panel_kernel_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(PanelBatch)>(chol_panel_kernel)),
                                 reinterpret_cast<const void*>(chol_panel_kernel_batched), sizeof(nk::ArgPack<PanelBatch>), "chol_panel_kernel");

// panel solves of up to two systems; calls[q] as prepared for the generic engine (C = A in place, B = the block's CHOL_WS
// workspace with leading dimension 64, N = K = valid columns <= 64, alpha = 1, beta = 0).  false: not that shape.
bool launch_chol_panel_pair(nk_ctx* ctx, const GemmCall* calls, int ncalls, int* rc) {
  *rc = NK_OK;
  PanelBatch pb;
  int maxblocks = 0;
  for (int q = 0; q < 2; ++q) {
    PanelSys& t = pb.s[q];
    t = PanelSys{};
    if (q >= ncalls || calls[q].M <= 0 || calls[q].N <= 0) continue;
    const GemmCall& c = calls[q];
    if (c.K != c.N || c.N > 64 || c.A != c.C || c.lda != c.ldc || c.ldb != 64 || c.alpha != 1.0 || c.beta != 0.0) return false;
    t.P = c.C; t.ldp = c.ldc; t.Linv = c.B; t.rows = (int)c.M; t.nb = (int)c.N; t.fix = chol_fix_enabled();
    t.nblocks = (t.rows + PANEL_ROWS - 1) / PANEL_ROWS;
    if (t.nblocks > maxblocks) maxblocks = t.nblocks;
  }
  if (maxblocks == 0) return true;
  hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)maxblocks, 2), dim3(256), 0, ctx->stream, pb);
  if (hipGetLastError() != hipSuccess) {
    set_error("chol_panel launch failed");
    *rc = NK_ERR_HIP;
  }
  return true;
}

}  // namespace nk
