// Fast fp64 "TN" GEMM for gfx950: C = A^T B with both operands stored contraction-major (rows = k), i.e. the layout of
// every big product of the fit: the Gram contractions over the n samples of the sample-major feature matrix
// F = [K_nm_in | U | K_nm_out] (regressors.py:151,153,162,164) and, with explicit transposes, the O(m^3) products.
//
// Differences from the generic engine (nk_gemm.hip):
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): one wave instruction moves one 128-double k-row (1 KiB)
//     straight into the padded LDS image, so there is no VGPR staging, no ds_write and no per-element branching;
//     out-of-range columns are clamped to valid addresses (their products land in accumulator rows/columns that are
//     never stored), rows past the K range are redirected to a zero page;
//   * the DMA for k-step t+1 is issued right after the barrier that retires step t-1 and stays in flight during
//     the 64 MFMAs of step t (one barrier per step);
//   * up to 4 problems that share K are fused into ONE launch (the four Gram products of a fit), all 128x128 tiles
//     of all problems x splitk K-slices, slice = blockIdx % splitk (a multiple of 8 => one K range per XCD, panels
//     shared through that XCD's L2), which removes the per-launch tails;
//   * partial tiles go to a slab [tile][slice][128][128]; a second kernel sums the slices in a fixed order and
//     applies alpha/beta, bounds and the symmetric mirror (deterministic; no float atomics).
#include "nk_common.h"
#include "nk_tn_kstep.inc"
#include "nk_tn_shared.h"

#include <cstdlib>
#include <cstring>

namespace nk {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int TBK = 16;
constexpr int TSTRIDE = 144;                               // doubles per LDS row: conflict-free ds_read_b64 operand fetch
constexpr int TSTAGE = 2 * TBK * TSTRIDE;                  // doubles per pipeline stage (A rows then B rows)
constexpr int TN_LDS_BYTES = 2 * TSTAGE * 8;               // two stages

struct TnDev {
  const double* A;
  const double* B;
  int64_t lda, ldb;
  int M, N;
  int tiles_n;     // tiles along N
  int tri;
  int ktrim;       // KTRIM_*: skip the k range in which a triangular operand is zero
  int tile_begin;  // first global tile index of this problem
  double* C;       // direct epilogue (splitk == 1): C = alpha * acc + beta * C
  int64_t ldc;
  double alpha, beta;
  double* Ct;      // optional transposed copy of the result
  int64_t ldct;
  const double* A_even;  // operand selection (TnParams::select_state): used instead of A / B when the step count
  const double* B_even;  // in select_state[0] is even (nullptr: no alternative)
  double* Caff;          // optional second output: Caff = aff_a * C + aff_c * I (leading dimension ldc)
  double aff_a, aff_c;
};
struct TnParams {
  TnDev p[TN_MAXP];
  int nprob;
  int ntiles;
  int K, splitk, klen;
  int kmask;            // experiments only (NYSKOOP_TN_KMASK): operand rows are fetched from k & kmask; -1 = off
  int use_asm;          // 1: the hand-scheduled k steps (default); 0: the compiler-scheduled form of the same steps (NYSKOOP_TN_ASM=0, A/B runs)
  const double* zeros;  // >= 1 KiB of zeros
  double* slab;
  // conditional launch (device-side early exit of an iteration that has already converged, no host round trip): the
  // launch is skipped when skip_state[0] != 0 && skip_state[0] <= skip_step
  const double* skip_state;
  int skip_step;
  // operand selection: the result of a ping-pong iteration of data-dependent length lives in one of two buffers; the
  // parity of select_state[0] (the number of steps actually taken, written on the device) picks A/B or A_even/B_even
  const double* select_state;
  // fused kernel-matrix epilogue (EPI == 1): out[i][j] = k(sqa[i] + sqb[j] - 2 acc)
  const double* sqa;
  const double* sqb;
  double* out;
  int64_t ldo;
  int ktype;
  double sigma0sq;
};

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  // (the builtin returns a signed int: without the casts the low word would be sign-extended into the high one)
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}
// One LDS-DMA instruction: the wave's 64 lanes move 16 bytes each from gsrc (per lane) to lds_row + 16 * lane.
// Issued through inline assembly on purpose: the compiler models the builtin as a FLAT access that may touch LDS
// ("pending flat"), after which every LDS wait it inserts degrades to lgkmcnt(0) -- which would stall the MFMA
// stream on the operand prefetch issued just before.  Hidden from its model, the waits on ds_read stay exact; the
// price is that the vmcnt wait before the stage barrier is ours to place (dma_wait_all).
__device__ __forceinline__ void dma_row(const double* row_base, uint32_t lane_off_bytes, uint32_t lds_addr) {
  // row_base is wave-uniform (scalar register pair), the per-lane part is a 32-bit byte offset, lds_addr the LDS byte
  // address of the destination row (wave-uniform): in the steady state the whole address stream of a k-step is a
  // handful of SALU instructions, no VALU
  // (the batched twins read their parameters from a device table: make the uniformity explicit)
  const uint64_t rbu = uniform_u64((uint64_t)(uintptr_t)row_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(lane_off_bytes), "s"(rbu), "s"((uint32_t)__builtin_amdgcn_readfirstlane(lds_addr))
               : "m0");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int EPI>
__device__ __forceinline__ void tn_body(const TnParams& P) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int split = blockIdx.x % P.splitk;
  const int gt = blockIdx.x / P.splitk;
  int pi = 0;
#pragma unroll
  for (int q = 1; q < TN_MAXP; ++q)
    if (q < P.nprob && gt >= P.p[q].tile_begin) pi = q;
  const TnDev pr = P.p[pi];
  int tm, tn;
  if (EPI >= 1) {
    // kernel-matrix launch: blocks b, b+8, ... share an XCD (round-robin dispatch), so give every XCD group its own
    // landmark tile columns tn = xcd + 8c for all sample tiles tm: the landmark panels (2 x 393 KB at m = 2000,
    // d = 384) stay resident in that XCD's L2 while the sample panels stream through once per group
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int cols = (pr.tiles_n - xcd + 7) >> 3;
    if (cols <= 0) return;
    tm = i / cols;
    tn = xcd + 8 * (i - tm * cols);
    if (tm * TBM >= pr.M) return;
  } else {
    tn_tile_coords(gt - pr.tile_begin, pr.tri, pr.tiles_n, pr.M, tm, tn);
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (EPI == 0 && P.skip_state != nullptr) {
    const double f = P.skip_state[0];
    if (f != 0.0 && f <= (double)P.skip_step) return;
  }
  const double* opA = pr.A;
  const double* opB = pr.B;
  if (EPI == 0 && P.select_state != nullptr && (((int)P.select_state[0]) & 1) == 0) {
    if (pr.A_even) opA = pr.A_even;
    if (pr.B_even) opB = pr.B_even;
  }
  int kbeg = split * P.klen;
  int kend = min(P.K, kbeg + P.klen);
  if (pr.ktrim == KTRIM_B_UPPER) kend = min(kend, (tn + 1) * TBM);  // B[k][n] = 0 for k > n
  if (pr.ktrim == KTRIM_A_LOWER) kbeg = max(kbeg, tm * TBM);        // A[k][i] = 0 for i > k
  const int ktiles = kend > kbeg ? (kend - kbeg + TBK - 1) / TBK : 0;

  // per-lane source columns (2 doubles per lane), clamped into the valid, 16-byte aligned range, as byte offsets
  const uint32_t offa = (uint32_t)min(tm * TBM + lane * 2, (pr.M - 1) & ~1) * 8u;
  const uint32_t offb = (uint32_t)min(tn * TBM + lane * 2, (pr.N - 1) & ~1) * 8u;
  const uint32_t offz = (uint32_t)lane * 16u;

  // wave w moves k-rows w, w+4, w+8, w+12 of the A panel and of the B panel
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;  // low half of a generic LDS address = the LDS byte address
  auto issue = [&](int kt, int stage) {
    const uint32_t sa = lds0 + (uint32_t)(stage * TSTAGE) * 8u;
    const uint32_t sb = sa + (uint32_t)(TBK * TSTRIDE) * 8u;
    const int k0 = kbeg + kt * TBK;
    if (k0 + TBK <= kend) {  // all 16 rows inside the K range (every step but the last of a slice)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = wave + 4 * q;
        dma_row(opA + (int64_t)((k0 + r) & P.kmask) * pr.lda, offa, sa + (uint32_t)(r * TSTRIDE) * 8u);
        dma_row(opB + (int64_t)((k0 + r) & P.kmask) * pr.ldb, offb, sb + (uint32_t)(r * TSTRIDE) * 8u);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = wave + 4 * q;
        const int k = k0 + r;
        if (k < kend) {
          dma_row(opA + (int64_t)k * pr.lda, offa, sa + (uint32_t)(r * TSTRIDE) * 8u);
          dma_row(opB + (int64_t)k * pr.ldb, offb, sb + (uint32_t)(r * TSTRIDE) * 8u);
        } else {  // rows past the K range come from the zero page
          dma_row(P.zeros, offz, sa + (uint32_t)(r * TSTRIDE) * 8u);
          dma_row(P.zeros, offz, sb + (uint32_t)(r * TSTRIDE) * 8u);
        }
      }
    }
  };

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g4 = lane >> 4;

  // Operand fragments of k-sub-step ks (4 contraction rows) of a stage: 4 + 4 ds_read_b64 per lane.
  auto frag = [&](int stage, int ks, double (&a)[4], double (&b)[4]) {
    const double* a_base = smem + stage * TSTAGE + wm * 64 + r16 + (ks * 4 + g4) * TSTRIDE;
    const double* b_base = a_base + TBK * TSTRIDE + (wn - wm) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = a_base[i * 16];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = b_base[j * 16];
  };
  auto mma = [&](const double (&a)[4], const double (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  // Software pipeline.  Two waves share a SIMD's matrix pipe and the arbiter alternates between them, which keeps two
  // waves that run the same loop in phase: whatever one of them does outside its MFMA stream (LDS waits, DMA issue and
  // its address arithmetic), the other does at the same moment, and the pipe idles for all of it.  So (1) the fragments
  // of sub-step ks + 1 are fetched into a second register set while the 16 MFMAs of sub-step ks are issued, the
  // step's single barrier sits inside the LAST sub-step (by then this wave has read all of the current stage, and the
  // DMA of the next stage, issued three MFMA blocks earlier, has landed) and the first fragments of the next step are
  // fetched across it; (2) the steady-state step is ONE hand-scheduled assembly block (nk_tn_kstep.inc, generated by
  // tools/gen_tn_kstep.py) in which every non-matrix instruction sits in a gap between two MFMAs.  The C++ form of the
  // same step below (compiler-scheduled) runs the last step(s) of a K range, whose DMA needs row checks.
  double a0[4], b0[4], a1[4], b1[4];
  if (ktiles > 0) {
    issue(0, 0);
    dma_wait_all();
    __syncthreads();
    frag(0, 0, a0, b0);
  }
  // LDS read addresses of the fragments (byte addresses; stage 1 = stage 0 + one stage)
  const uint32_t ard0 = lds0 + (uint32_t)(wm * 64 + r16 + g4 * TSTRIDE) * 8u;
  const uint32_t brd0 = lds0 + (uint32_t)(TBK * TSTRIDE + wn * 64 + r16 + g4 * TSTRIDE) * 8u;
  constexpr uint32_t STAGE_B = (uint32_t)TSTAGE * 8u;
  const uint32_t stra = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pr.lda * 32));  // 4 rows, in bytes
  const uint32_t strb = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pr.ldb * 32));
  // Steady steps: kt = 0 .. nfull - 2 (step kt issues the DMA of step kt + 1, which must be a full step).  ONE assembly
  // block runs them two at a time (LDS stage 0, then 1), then an odd one left over (on stage 0: the pairs start at
  // kt = 0), then -- when the K range has no partial last step -- its final full step (no DMA, no barrier).  The C++
  // steps below only run the end of a K range that is not a multiple of 16 rows.  (One block, not one per case: with
  // several blocks of 40 register operands each the compiler's allocation across their joins spills.)
#define NK_TN_ACC_OPERANDS                                                                                             \
  [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c02] "+v"(acc[0][2]), [c03] "+v"(acc[0][3]), [c10] "+v"(acc[1][0]),   \
      [c11] "+v"(acc[1][1]), [c12] "+v"(acc[1][2]), [c13] "+v"(acc[1][3]), [c20] "+v"(acc[2][0]),                      \
      [c21] "+v"(acc[2][1]), [c22] "+v"(acc[2][2]), [c23] "+v"(acc[2][3]), [c30] "+v"(acc[3][0]),                      \
      [c31] "+v"(acc[3][1]), [c32] "+v"(acc[3][2]), [c33] "+v"(acc[3][3]), [a00] "+v"(a0[0]), [a01] "+v"(a0[1]),       \
      [a02] "+v"(a0[2]), [a03] "+v"(a0[3]), [b00] "+v"(b0[0]), [b01] "+v"(b0[1]), [b02] "+v"(b0[2]), [b03] "+v"(b0[3]), \
      [a10] "=&v"(a1[0]), [a11] "=&v"(a1[1]), [a12] "=&v"(a1[2]), [a13] "=&v"(a1[3]), [b10] "=&v"(b1[0]),              \
      [b11] "=&v"(b1[1]), [b12] "=&v"(b1[2]), [b13] "=&v"(b1[3])
#define NK_TN_IN_OPERANDS                                                                                             \
  [ard0] "v"(ard0), [brd0] "v"(brd0), [ard1] "v"(ard1), [brd1] "v"(brd1), [voa] "v"(offa), [vob] "v"(offb),           \
      [rowa] "s"(rowa), [rowb] "s"(rowb), [stra] "s"(stra), [strb] "s"(strb), [dst0] "s"(dst0), [dst1] "s"(dst1)
#define NK_TN_CLOBBERS "memory", "m0", "scc", "s92", "s93", "s94", "s95"
  const int nfull = (kend - kbeg) / TBK;
  int kt_start = 0;
  if (P.kmask == -1 && P.use_asm && nfull >= 1) {
    const uint32_t dst0 = (uint32_t)__builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(wave * TSTRIDE) * 8u);
    const uint32_t dst1 = dst0 + STAGE_B;
    const uint32_t ard1 = ard0 + STAGE_B, brd1 = brd0 + STAGE_B;
    const int steady = nfull - 1;
    uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(steady / 2));
    const uint32_t flags =
        (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((steady & 1) | (nfull == ktiles ? 2 : 0)));
    kt_start = 2 * (int)cnt + (int)(flags & 1) + (int)(flags >> 1);
    const uint64_t rowa = uniform_u64((uint64_t)(uintptr_t)(opA + (int64_t)(kbeg + TBK + wave) * pr.lda));
    const uint64_t rowb = uniform_u64((uint64_t)(uintptr_t)(opB + (int64_t)(kbeg + TBK + wave) * pr.ldb));
    if (kt_start > 0)
      asm volatile(NK_TN_KSTEPS_ASM
                   : NK_TN_ACC_OPERANDS, [cnt] "+s"(cnt)
                   : NK_TN_IN_OPERANDS, [flags] "s"(flags)
                   : NK_TN_CLOBBERS);
  }
  for (int kt = kt_start; kt < ktiles; ++kt) {
    const int st = kt & 1;
    const bool more = kt + 1 < ktiles;
    frag(st, 1, a1, b1);
    if (more) issue(kt + 1, st ^ 1);  // everybody left stage st^1 at the barrier of the previous step
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    frag(st, 2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    frag(st, 3, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    dma_wait_all();
    __syncthreads();  // (waits for this wave's LDS reads as well)
    if (more) frag(st ^ 1, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }

  if (EPI == 0 && P.splitk == 1) {
    // single K slice: finish in place (alpha/beta, bounds, symmetric mirror), no slab round trip
    const bool mirror = pr.tri == TRI_UPPER_MIRROR && tm != tn;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = tm * TBM + wm * 64 + i * 16 + g4 + 4 * reg;
          const int col = tn * TBM + wn * 64 + j * 16 + r16;
          if (row < pr.M && col < pr.N) {
            double v = pr.alpha * acc[i][j][reg];
            if (pr.beta != 0.0) v += pr.beta * pr.C[(int64_t)row * pr.ldc + col];
            pr.C[(int64_t)row * pr.ldc + col] = v;
            if (mirror) pr.C[(int64_t)col * pr.ldc + row] = v;
            if (pr.Ct) pr.Ct[(int64_t)col * pr.ldct + row] = v;
            if (pr.Caff) {
              pr.Caff[(int64_t)row * pr.ldc + col] = pr.aff_a * v + (row == col ? pr.aff_c : 0.0);
              if (mirror) pr.Caff[(int64_t)col * pr.ldc + row] = pr.aff_a * v;
            }
          }
        }
  } else if (EPI == 0) {
    // raw 128x128 partial tile -> slab[(tile*splitk + split)]
    double* out = P.slab + ((int64_t)gt * P.splitk + split) * (TBM * TBM);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = wm * 64 + i * 16 + g4 + 4 * reg;
          const int col = wn * 64 + j * 16 + r16;
          out[row * TBM + col] = acc[i][j][reg];
        }
  } else {
    // kernel-matrix epilogue (EPI = 1 RBF, 2 Matern-5/2, 3 linear; compile-time so that one formula is inlined):
    // squared distance from the Gram form, then the kernel function
    const bool interior = (tm + 1) * TBM <= pr.M && (tn + 1) * TBM <= pr.N;
    const double exp_ca = exp_nonpos_ca();
    double sb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = tn * TBM + wn * 64 + j * 16 + r16;
      sb[j] = (EPI != 3 && col < pr.N) ? P.sqb[col] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = tm * TBM + wm * 64 + i * 16 + g4 + 4 * reg;
        if (!interior && row >= pr.M) continue;
        const double sa = EPI != 3 ? P.sqa[row] : 0.0;
        double* orow = P.out + (int64_t)row * P.ldo + tn * TBM + wn * 64 + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double dot = acc[i][j][reg];
          double v;
          if (EPI == 3) {
            v = dot + P.sigma0sq;
          } else {
            const double D = fmax(sa + sb[j] - 2.0 * dot, 0.0);
            if (EPI == 1) {
              v = exp_nonpos(-0.5 * D, exp_ca);
            } else {
              const double t = sqrt(D) * 2.23606797749978969641;
              v = (1.0 + t + t * t / 3.0) * exp_nonpos(-t, exp_ca);
            }
          }
          if (interior || tn * TBM + wn * 64 + j * 16 + r16 < pr.N) orow[j * 16] = v;
        }
      }
  }
}

// Entry points.  The fused Gram launch of the fit (the dominant kernel of the whole path) gets its own symbol so that
// `rocprofv3 --kernel-trace --stats` reports it apart from the O(m^3) products that share the engine.
template <int EPI>
__global__ void __launch_bounds__(256, 2) gemm_tn_f64_kernel(TnParams P) {
  tn_body<EPI>(P);
}
__global__ void __launch_bounds__(256, 2) gram_fused_f64_kernel(TnParams P) {
  tn_body<0>(P);
}
// batched twins (nk_lockstep.h): the same body, parameters from table[blockIdx.z]
template <int EPI>
__global__ void __launch_bounds__(256, 2) gemm_tn_f64_kernel_batched(const nk::ArgPack<TnParams>* table) {
  tn_body<EPI>(table[blockIdx.z].v);  // by reference: a local copy of the (dynamically indexed) struct would live in scratch
}
__global__ void __launch_bounds__(256, 2) gram_fused_f64_kernel_batched(const nk::ArgPack<TnParams>* table) {
  tn_body<0>(table[blockIdx.z].v);
}
#define NK_TN_TWIN(E)                                                                                                   \
  static nk::TwinReg tn_twin_reg_##E(reinterpret_cast<const void*>(static_cast<void (*)(TnParams)>(gemm_tn_f64_kernel<E>)), \
                                     reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<E>),                        \
                                     sizeof(nk::ArgPack<TnParams>), "gemm_tn_f64_kernel<" #E ">");
NK_TN_TWIN(0) NK_TN_TWIN(1) NK_TN_TWIN(2) NK_TN_TWIN(3)
static nk::TwinReg gram_fused_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(TnParams)>(gram_fused_f64_kernel)),
                                       reinterpret_cast<const void*>(gram_fused_f64_kernel_batched),
                                       sizeof(nk::ArgPack<TnParams>), "gram_fused_f64_kernel");

// C = alpha * sum_s slab[tile][s] + beta * C (bounds, mirror, transposed copy).  RPARTS workgroups per tile, each summing
// a 16-row band of the slices in slice order (deterministic) with 16-byte loads; the mirrored / transposed copies go
// through LDS so that their stores are 64-byte runs instead of single strided doubles.
__device__ __forceinline__ void gemm_tn_reduce_kernel_body(const TnRedParams& P) {
  __shared__ double sh[16][130];
  if (P.skip_state != nullptr) {
    const double f = P.skip_state[0];
    if (f != 0.0 && f <= (double)P.skip_step) return;
  }
  const int gt = blockIdx.x / RPARTS, part = blockIdx.x % RPARTS;
  int pi = 0;
#pragma unroll
  for (int q = 1; q < TN_MAXP; ++q)
    if (q < P.nprob && gt >= P.p[q].tile_begin) pi = q;
  const TnRed pr = P.p[pi];
  int tm, tn;
  tn_tile_coords(gt - pr.tile_begin, pr.tri, pr.tiles_n, pr.M, tm, tn);
  typedef double d2 __attribute__((ext_vector_type(2)));
  const double* base = P.slab + (int64_t)gt * P.splitk * (TBM * TBM) + part * (16 * TBM) + threadIdx.x * 2;
  d2 s[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = d2{0.0, 0.0};
  for (int k = 0; k < P.splitk; ++k) {
    const double* bk = base + (int64_t)k * (TBM * TBM);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] += *reinterpret_cast<const d2*>(bk + 512 * i);
  }
  const bool mirror = pr.tri == TRI_UPPER_MIRROR && tm != tn;
  const bool transposed = mirror || pr.Ct != nullptr;
  const int row0 = tm * TBM + part * 16, col0 = tn * TBM;
  double rsum = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x * 2 + 512 * i;
    const int r = idx >> 7, c = idx & 127;
    const int row = row0 + r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = col0 + c + h;
      double v = 0.0;
      if (row < pr.M && col < pr.N) {
        v = pr.alpha * s[i][h];
        if (pr.beta != 0.0) v += pr.beta * pr.C[(int64_t)row * pr.ldc + col];
        pr.C[(int64_t)row * pr.ldc + col] = v;
        if (pr.Caff) pr.Caff[(int64_t)row * pr.ldc + col] = pr.aff_a * v + (row == col ? pr.aff_c : 0.0);
        const double e = v - (row == col ? 1.0 : 0.0);
        rsum = fma(e, e, rsum);
      }
      if (transposed) sh[r][c + h] = v;
    }
  }
  if (P.resid_partials != nullptr) {  // fixed-order workgroup sum (uniform branch)
    __shared__ double wsum[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rsum += __shfl_down(rsum, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = rsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double w = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
      P.resid_partials[blockIdx.x] = mirror ? 2.0 * w : w;
    }
  }
  if (!transposed) return;
  __syncthreads();
  // thread t: column c = t / 2, rows 8 * (t & 1) .. + 8 of this band
  const int c = threadIdx.x >> 1, rb = (threadIdx.x & 1) * 8;
  const int col = col0 + c;
  if (col >= pr.N) return;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = row0 + rb + q;
    if (row >= pr.M) break;
    const double v = sh[rb + q][c];
    if (mirror) pr.C[(int64_t)col * pr.ldc + row] = v;
    if (pr.Ct) pr.Ct[(int64_t)col * pr.ldct + row] = v;
    if (mirror && pr.Caff) pr.Caff[(int64_t)col * pr.ldc + row] = pr.aff_a * v;  // off-diagonal tile: no identity term
  }
}
__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel(TnRedParams P) { gemm_tn_reduce_kernel_body(P); }
__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel_batched(const nk::ArgPack<TnRedParams>* table) {
  gemm_tn_reduce_kernel_body(table[blockIdx.z].v);
}
static nk::TwinReg gemm_tn_reduce_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(TnRedParams)>(gemm_tn_reduce_kernel)),
                                           reinterpret_cast<const void*>(gemm_tn_reduce_kernel_batched),
                                           sizeof(nk::ArgPack<TnRedParams>), "gemm_tn_reduce_kernel");

int launch_tn_reduce(nk_ctx* ctx, const TnRedParams& R, int ntiles) {
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)ntiles * RPARTS), dim3(256), 0, ctx->stream, R);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// 32x32 tiled transpose through LDS
__device__ __forceinline__ void transpose_kernel_body(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd, int rows, int cols) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int row = by + r, col = bx + tx;
    if (row < rows && col < cols) tile[r][tx] = src[(int64_t)row * lds_ + col];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int orow = bx + r, ocol = by + tx;  // dst[col][row]
    if (orow < cols && ocol < rows) dst[(int64_t)orow * ldd + ocol] = tile[tx][r];
  }
}
__global__ void __launch_bounds__(256) transpose_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd, int rows, int cols) { transpose_kernel_body(src, lds_, dst, ldd, rows, cols); }
NK_BATCHED_TWIN(transpose_kernel, (256), const double*, int64_t, double*, int64_t, int, int)

int launch_transpose(nk_ctx* ctx, const double* src, int64_t lds_, double* dst, int64_t ldd, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return NK_OK;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, ctx->stream, src, lds_, dst, ldd, rows, cols);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool tn_fast_ok(const TnProblem& p) {
  return p.M >= 2 && p.N >= 2 && aligned16(p.A) && aligned16(p.B) && p.lda % 2 == 0 && p.ldb % 2 == 0 &&
         p.lda >= ((p.M + 1) & ~1) && p.ldb >= ((p.N + 1) & ~1);
}

static bool g_tn_attr_set = false;
static int tn_use_asm() {
  const char* e = getenv("NYSKOOP_TN_ASM");  // read per launch: A/B runs flip it inside one process
  return (e && e[0] == '0') ? 0 : 1;
}

int tn_ensure_zero_page(nk_ctx* ctx) {
  if (ctx->d_zeros) return NK_OK;
  NK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_zeros), 4096));
  NK_HIP(hipMemset(ctx->d_zeros, 0, 4096));  // blocking: both streams of the context read this page
  return NK_OK;
}

int launch_gemm_tn_multi(nk_ctx* ctx, const TnProblem* probs, int nprob, int64_t K, int splitk, float* ms_kernel,
                         bool sync_timing, const TnSkip* skip) {
  NK_REQUIRE(nprob >= 1 && nprob <= TN_MAXP, "gemm_tn_multi: 1..4 problems");
  NK_REQUIRE(K >= 0 && K < (1LL << 31), "gemm_tn_multi: K out of range");
  NK_TRY(tn_ensure_zero_page(ctx));
  TnParams P;
  TnRedParams R;
  int ntiles = 0;
  for (int q = 0; q < nprob; ++q) {
    const TnProblem& s = probs[q];
    NK_REQUIRE(tn_fast_ok(s), "gemm_tn_multi: problem %d violates the alignment contract", q);
    NK_REQUIRE(s.tri == TRI_FULL || (s.tri == TRI_UPPER_MIRROR && s.M == s.N), "gemm_tn_multi: bad tri mode");
    const int tmn = (s.M + TBM - 1) / TBM, tnn = (s.N + TBM - 1) / TBM;
    TnDev& d = P.p[q];
    d.A = s.A; d.B = s.B; d.lda = s.lda; d.ldb = s.ldb; d.M = s.M; d.N = s.N;
    d.tiles_n = tnn; d.tri = s.tri; d.ktrim = s.ktrim; d.tile_begin = ntiles;
    d.C = s.C; d.ldc = s.ldc; d.alpha = s.alpha; d.beta = s.beta; d.Ct = s.Ct; d.ldct = s.ldct;
    d.A_even = s.A_even; d.B_even = s.B_even;
    d.Caff = s.Caff; d.aff_a = s.aff_a; d.aff_c = s.aff_c;
    TnRed& r = R.p[q];
    r.Caff = s.Caff; r.aff_a = s.aff_a; r.aff_c = s.aff_c;
    r.Ct = s.Ct; r.ldct = s.ldct;
    r.C = s.C; r.ldc = s.ldc; r.M = s.M; r.N = s.N; r.tiles_n = tnn; r.tri = s.tri; r.tile_begin = ntiles;
    r.alpha = s.alpha; r.beta = s.beta;
    ntiles += s.tri == TRI_FULL ? tmn * tnn : tmn * (tmn + 1) / 2;
  }
  for (int q = nprob; q < TN_MAXP; ++q) {
    P.p[q] = P.p[0];
    P.p[q].tile_begin = 1 << 30;
    R.p[q] = R.p[0];
    R.p[q].tile_begin = 1 << 30;
  }
  const int ktiles_total = (int)((K + TBK - 1) / TBK);
  const bool want_resid = skip && skip->resid_partials != nullptr;
  if (splitk <= 0) {
    // Pick the number of K slices with a small cost model calibrated on MI355X (profiles/r01_*): a k-step costs
    // 1.22 units per workgroup when two workgroups share a CU and 2.35 when a workgroup is alone on its CU (one wave
    // per SIMD cannot hide the barrier + DMA latency); workgroups beyond 2/CU run as further rounds.  The slab
    // round trip (write + read of ntiles*splitk tiles) is charged at ~4 TB/s.  Multiples of 8 keep one K range per
    // XCD (panels shared through that XCD's L2) and get a small bonus.
    const int slots = 2 * ctx->num_cu;
    const int max_split = ktiles_total / 8 > 1 ? ktiles_total / 8 : 1;
    double best = 1e300;
    splitk = 1;
    for (int c = 1; c <= 64 && c <= max_split; ++c) {
      const int64_t wgs = (int64_t)ntiles * c;
      const double steps = (double)((ktiles_total + c - 1) / c);
      const int64_t full = wgs / slots, rem = wgs % slots;
      double t = full * 2.44 * steps;
      if (rem > 0) t += (rem <= ctx->num_cu ? 2.35 : 2.44) * steps;
      t *= 4096.0;                                                      // cycles of MFMA per k-step and wave
      if (c > 1) t += (double)wgs * (TBM * TBM * 8.0 * 2.0) / 4.0e12 * 2.4e9;  // slab write + read, in cycles
      t += 2000.0 * c / 8.0;                                            // mild preference for fewer slices
      if (c % 8 == 0) t *= 0.97;
      if (t < best) { best = t; splitk = c; }
    }
  }
  if (want_resid && splitk < 2) splitk = 2;  // the residual partials come out of the reduce kernel
  if (nprob >= 3 && getenv("NYSKOOP_TN_SPLITK")) splitk = atoi(getenv("NYSKOOP_TN_SPLITK"));  // experiments
  P.kmask = getenv("NYSKOOP_TN_KMASK") ? (int)strtol(getenv("NYSKOOP_TN_KMASK"), nullptr, 0) : -1;
  P.use_asm = tn_use_asm();
  P.nprob = nprob; P.ntiles = ntiles; P.K = (int)K; P.splitk = splitk;
  P.klen = ((ktiles_total + splitk - 1) / splitk) * TBK;
  if (P.klen == 0) P.klen = TBK;
  P.zeros = ctx->d_zeros;
  const ArenaMark mark = arena_mark(ctx);
  double* slab = nullptr;
  if (splitk > 1) NK_TRY(arena_alloc_t(ctx, (size_t)ntiles * splitk * TBM * TBM, &slab));
  P.slab = slab;
  R.nprob = nprob; R.splitk = splitk; R.slab = slab;
  P.skip_state = R.skip_state = skip ? skip->state : nullptr;
  P.skip_step = R.skip_step = skip ? skip->step : 0;
  P.select_state = skip ? skip->select : nullptr;
  R.resid_partials = (skip && splitk > 1 && nprob == 1) ? skip->resid_partials : nullptr;
  if (skip && skip->resid_count) *skip->resid_count = R.resid_partials ? ntiles * RPARTS : 0;
  if (!g_tn_attr_set) {
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_fused_f64_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_fused_f64_kernel_batched),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    g_tn_attr_set = true;
  }
  if (ms_kernel) NK_HIP(hipEventRecord(ctx->ev[14], ctx->stream));
  P.sqa = P.sqb = nullptr; P.out = nullptr; P.ldo = 0; P.ktype = 0; P.sigma0sq = 0.0;
  if (nprob >= 3)  // the fit's fused Gram launch
    hipLaunchKernelGGL(gram_fused_f64_kernel, dim3((unsigned)(ntiles * splitk)), dim3(256), TN_LDS_BYTES, ctx->stream, P);
  else
    hipLaunchKernelGGL(gemm_tn_f64_kernel<0>, dim3((unsigned)(ntiles * splitk)), dim3(256), TN_LDS_BYTES, ctx->stream, P);
  NK_HIP(hipGetLastError());
  if (ms_kernel) NK_HIP(hipEventRecord(ctx->ev[15], ctx->stream));
  if (splitk > 1) {
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)ntiles * RPARTS), dim3(256), 0, ctx->stream, R);
    NK_HIP(hipGetLastError());
  }
  if (ms_kernel && sync_timing) {  // otherwise the caller reads ev[14] -> ev[15] after its own synchronisation
    NK_HIP(hipEventSynchronize(ctx->ev[15]));
    NK_HIP(hipEventElapsedTime(ms_kernel, ctx->ev[14], ctx->ev[15]));
  }
  arena_release(ctx, mark);
  return NK_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Kernel matrix in Gram form on the MFMA engine (large d): out[i][j] = k(|a_i|^2 + |b_j|^2 - 2 a_i.b_j) with the rows
// centred, scaled by 1/lengthscale and transposed to contraction-major first.  Used for the two n x m blocks of the
// fit when d >= 32; K(Z, Z), small d and nk_kernel_matrix keep the direct-difference kernel (nk_kmat.hip).
// Error of the Gram form: |dD| <~ eps * (|a|^2 + |b|^2) absolute, i.e. a RELATIVE error of |dD|/2 on k (RBF), which the
// centring keeps at the 1e-15 level for the shapes of SURVEY 8d.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void prep_rows_kernel_body(const double* __restrict__ X, int64_t ldx, int rows, int d, const double* __restrict__ winv, const double* __restrict__ center, double* __restrict__ Xt, int64_t ldt) {
  __shared__ double tile[32][33];
  const int k0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int row = r0 + r, k = k0 + tx;
    if (row < rows && k < d) tile[r][tx] = (X[(int64_t)row * ldx + k] - center[k]) * winv[k];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, row = r0 + tx;
    if (k < d && row < rows) Xt[(int64_t)k * ldt + row] = tile[tx][r];
  }
}
__global__ void __launch_bounds__(256) prep_rows_kernel(const double* __restrict__ X, int64_t ldx, int rows, int d, const double* __restrict__ winv, const double* __restrict__ center, double* __restrict__ Xt, int64_t ldt) { prep_rows_kernel_body(X, ldx, rows, d, winv, center, Xt, ldt); }
NK_BATCHED_TWIN(prep_rows_kernel, (256), const double*, int64_t, int, int, const double*, const double*, double*, int64_t)
// squared norms of the columns of Xt (d x rows): 64 columns per workgroup, the d rows dealt to the four waves
// (coalesced 512-byte row segments, four loads in flight per lane), fixed-order sum of the four partials
__device__ __forceinline__ void colsq_kernel_body(const double* __restrict__ Xt, int64_t ldt, int rows, int d, double* __restrict__ sq) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i < rows) {
    int k = w;
    for (; k + 12 < d; k += 16) {
      const double v0 = Xt[(int64_t)k * ldt + i], v1 = Xt[(int64_t)(k + 4) * ldt + i];
      const double v2 = Xt[(int64_t)(k + 8) * ldt + i], v3 = Xt[(int64_t)(k + 12) * ldt + i];
      s0 = fma(v0, v0, s0); s1 = fma(v1, v1, s1); s2 = fma(v2, v2, s2); s3 = fma(v3, v3, s3);
    }
    for (; k < d; k += 4) {
      const double v = Xt[(int64_t)k * ldt + i];
      s0 = fma(v, v, s0);
    }
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && i < rows) sq[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
__global__ void __launch_bounds__(256) colsq_kernel(const double* __restrict__ Xt, int64_t ldt, int rows, int d, double* __restrict__ sq) { colsq_kernel_body(Xt, ldt, rows, d, sq); }
NK_BATCHED_TWIN(colsq_kernel, (256), const double*, int64_t, int, int, double*)
// column means in two deterministic passes: per-chunk partial sums (coalesced across columns), then a fixed-order sum
__device__ __forceinline__ void colsum_partial_kernel_body(const double* __restrict__ Z, int64_t ldz, int rows, int d, int rows_per_chunk, double* __restrict__ partial) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  double s = 0.0;
  for (int r = r0; r < r1; ++r) s += Z[(int64_t)r * ldz + k];
  partial[(int64_t)blockIdx.y * d + k] = s;
}
__global__ void __launch_bounds__(256) colsum_partial_kernel(const double* __restrict__ Z, int64_t ldz, int rows, int d, int rows_per_chunk, double* __restrict__ partial) { colsum_partial_kernel_body(Z, ldz, rows, d, rows_per_chunk, partial); }
NK_BATCHED_TWIN(colsum_partial_kernel, (256), const double*, int64_t, int, int, int, double*)
__device__ __forceinline__ void colmean_finish_kernel_body(const double* __restrict__ partial, int chunks, int rows, int d, double* __restrict__ mean) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d) return;
  double s = 0.0;
  for (int c = 0; c < chunks; ++c) s += partial[(int64_t)c * d + k];
  mean[k] = s / rows;
}
__global__ void __launch_bounds__(256) colmean_finish_kernel(const double* __restrict__ partial, int chunks, int rows, int d, double* __restrict__ mean) { colmean_finish_kernel_body(partial, chunks, rows, d, mean); }
NK_BATCHED_TWIN(colmean_finish_kernel, (256), const double*, int, int, int, double*)

int launch_colmean(nk_ctx* ctx, const double* Z, int64_t ldz, int rows, int d, double* mean) {
  const ArenaMark mk = arena_mark(ctx);
  const int rpc = 32;
  const int chunks = (rows + rpc - 1) / rpc;
  double* partial = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)chunks * d, &partial));
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((d + 255) / 256, chunks), dim3(256), 0, ctx->stream, Z, ldz, rows, d, rpc,
                     partial);
  hipLaunchKernelGGL(colmean_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, ctx->stream, partial, chunks, rows, d, mean);
  NK_HIP(hipGetLastError());
  arena_release(ctx, mk);
  return NK_OK;
}

int prep_rows(nk_ctx* ctx, const double* X, int64_t ldx, int64_t rows, int d, const double* winv, const double* center,
              double* Xt, int64_t ldt, double* sq) {
  dim3 grid((d + 31) / 32, (unsigned)((rows + 31) / 32));
  hipLaunchKernelGGL(prep_rows_kernel, grid, dim3(256), 0, ctx->stream, X, ldx, (int)rows, d, winv, center, Xt, ldt);
  hipLaunchKernelGGL(colsq_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, ctx->stream, Xt, ldt, (int)rows, d, sq);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

int launch_kmat_gram(nk_ctx* ctx, int ktype, const double* At, int64_t ldat, const double* sqa, int64_t nA,
                     const double* Bt, int64_t ldbt, const double* sqb, int64_t nB, int d, double sigma0, double* out,
                     int64_t ldo) {
  NK_TRY(tn_ensure_zero_page(ctx));
  TnProblem tp;
  tp.A = At; tp.B = Bt; tp.lda = ldat; tp.ldb = ldbt; tp.M = (int)nA; tp.N = (int)nB;
  NK_REQUIRE(tn_fast_ok(tp), "kmat_gram: operands violate the alignment contract");
  TnParams P;
  P.skip_state = nullptr; P.skip_step = 0; P.select_state = nullptr;
  const int tmn = (int)((nA + TBM - 1) / TBM), tnn = (int)((nB + TBM - 1) / TBM);
  TnDev& dv = P.p[0];
  dv.A = At; dv.B = Bt; dv.lda = ldat; dv.ldb = ldbt; dv.M = (int)nA; dv.N = (int)nB; dv.tiles_n = tnn; dv.tri = TRI_FULL;
  dv.tile_begin = 0; dv.ktrim = KTRIM_NONE; dv.A_even = dv.B_even = nullptr; dv.Caff = nullptr; dv.aff_a = dv.aff_c = 0.0;
  dv.C = nullptr; dv.ldc = 0; dv.alpha = 1.0; dv.beta = 0.0; dv.Ct = nullptr; dv.ldct = 0;
  for (int q = 1; q < TN_MAXP; ++q) { P.p[q] = P.p[0]; P.p[q].tile_begin = 1 << 30; }
  P.nprob = 1; P.ntiles = tmn * tnn; P.K = d; P.splitk = 1; P.kmask = -1; P.use_asm = tn_use_asm();
  P.klen = ((d + TBK - 1) / TBK) * TBK;
  P.zeros = ctx->d_zeros; P.slab = nullptr;
  P.sqa = sqa; P.sqb = sqb; P.out = out; P.ldo = ldo; P.ktype = ktype; P.sigma0sq = sigma0 * sigma0;
  if (!g_tn_attr_set) {
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_fused_f64_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel<3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f64_kernel_batched<3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_fused_f64_kernel_batched),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
    g_tn_attr_set = true;
  }
  const unsigned grid = 8u * (unsigned)tmn * (unsigned)((tnn + 7) / 8);
  if (ktype == NK_KERNEL_RBF)
    hipLaunchKernelGGL(gemm_tn_f64_kernel<1>, dim3(grid), dim3(256), TN_LDS_BYTES, ctx->stream, P);
  else if (ktype == NK_KERNEL_MATERN52)
    hipLaunchKernelGGL(gemm_tn_f64_kernel<2>, dim3(grid), dim3(256), TN_LDS_BYTES, ctx->stream, P);
  else
    hipLaunchKernelGGL(gemm_tn_f64_kernel<3>, dim3(grid), dim3(256), TN_LDS_BYTES, ctx->stream, P);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
