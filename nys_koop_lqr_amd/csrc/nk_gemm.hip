// fp64 GEMM engine for gfx950 built on v_mfma_f64_16x16x4_f64 (measured 64 cycles/instruction/SIMD, 77.7 TF
// chip-wide; tools/microbench_f64_v2.hip).
//
// Replaces the reference's NumPy `@` contractions: regressors.py:151,153,162,164 (Gram / cross-Gram over the
// n samples) and :156,166,167 (operator products), and is the building block of the O(m^3) stage.
//
//   C[M x N] = alpha * op(A) * op(B) + beta * C      (row-major, explicit leading dimensions)
//     transA = false: A stored M x K (k contiguous)      transA = true: A stored K x M (m contiguous)
//     transB = false: B stored K x N (n contiguous)      transB = true: B stored N x K (k contiguous)
//
// Tiling: one 256-thread workgroup (4 wave64) owns a 128 x 128 tile of C; each wave owns 64 x 64 = 4 x 4 MFMA
// tiles (16 accumulators x 4 f64 = 128 VGPRs).  K is walked in steps of 16 through two LDS buffers; both
// operands are kept in LDS as [k][m|n] with a row stride of 144 doubles so that the MFMA operand fetch
// (lane l reads [k = l>>4][i = l&15], one ds_read_b64) is bank-conflict free for each 32-lane half.
// Global loads for step t+1 are issued into registers before the MFMAs of step t and written to the other LDS
// buffer afterwards (one barrier per step).
//
// Split-K: the K range is cut into `splitk` slices, slice = blockIdx % splitk so that, with splitk a multiple of 8,
// every XCD (blocks are dealt round-robin over the 8 XCDs) streams its own K range and all tiles on that XCD
// share the same operand panels through the XCD's L2.  Partial tiles go to a slab and a second kernel reduces
// them in a fixed order (deterministic, no float atomics).
#include "nk_common.h"

namespace nk {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_STRIDE = 144;  // doubles; 144 mod 32 == 16 -> conflict-free operand fetch
constexpr int GEMM_THREADS = 256;
constexpr int LDS_BYTES = 2 /*buffers*/ * 2 /*operands*/ * BK * LDS_STRIDE * 8;

struct GemmParams {
  const double* A;
  const double* B;
  double* C;
  double* slab;  // nullptr: write C directly
  int64_t lda, ldb, ldc;
  int M, N, K;
  int tiles_m, tiles_n;
  int tri;
  int splitk;
  int klen;  // K elements per split (multiple of BK)
  int vecA, vecB;
  double alpha, beta;
  int nblocks;  // tiles * splitk of this problem
};
constexpr int GEMM_MAX_BATCH = 2;
struct GemmBatch {
  GemmParams p[GEMM_MAX_BATCH];
};

__device__ __forceinline__ void tile_from_index(int t, int tiles_m, int tiles_n, int tri, int& tm, int& tn) {
  if (tri == TRI_FULL) {
    tm = t / tiles_n;
    tn = t - tm * tiles_n;
  } else if (tri == TRI_UPPER_MIRROR) {
    int row = 0, rem = t;
    while (rem >= tiles_n - row) {
      rem -= tiles_n - row;
      ++row;
    }
    tm = row;
    tn = row + rem;
  } else {  // TRI_LOWER: tile row r holds min(r + 1, tiles_n) tiles (rows past the square part are full)
    int row = 0, rem = t;
    for (;;) {
      const int cnt = row + 1 < tiles_n ? row + 1 : tiles_n;
      if (rem < cnt) break;
      rem -= cnt;
      ++row;
    }
    tm = row;
    tn = rem;
  }
}

// Load this thread's 8 doubles of a [BK x 128] tile whose global rows are k (contiguous along x = m|n).
__device__ __forceinline__ void load_direct(const double* __restrict__ P, int64_t ld, int k0, int kend, int x0, int X,
                                            int vec, double (&r)[8]) {
  const int t = threadIdx.x;
  const int k = k0 + (t >> 4);
  const int c = x0 + (t & 15) * 8;
  if (k < kend) {
    const double* src = P + (int64_t)k * ld + c;
    if (vec && c + 7 < X) {
      const double2* s2 = reinterpret_cast<const double2*>(src);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double2 v = s2[q];
        r[2 * q] = v.x;
        r[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = (c + q < X) ? src[q] : 0.0;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = 0.0;
  }
}
__device__ __forceinline__ void store_direct(double* __restrict__ S, const double (&r)[8]) {
  const int t = threadIdx.x;
  double2* dst = reinterpret_cast<double2*>(S + (t >> 4) * LDS_STRIDE + (t & 15) * 8);
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = make_double2(r[2 * q], r[2 * q + 1]);
}

// Load this thread's 8 doubles of a [128 x BK] tile whose global rows are x (contiguous along k).
__device__ __forceinline__ void load_transposing(const double* __restrict__ P, int64_t ld, int k0, int kend, int x0,
                                                 int X, int vec, double (&r)[8]) {
  const int t = threadIdx.x;
  const int x = x0 + (t >> 1);
  const int k = k0 + (t & 1) * 8;
  if (x < X) {
    const double* src = P + (int64_t)x * ld + k;
    if (vec && k + 7 < kend) {
      const double2* s2 = reinterpret_cast<const double2*>(src);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double2 v = s2[q];
        r[2 * q] = v.x;
        r[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = (k + q < kend) ? src[q] : 0.0;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = 0.0;
  }
}
__device__ __forceinline__ void store_transposing(double* __restrict__ S, const double (&r)[8]) {
  const int t = threadIdx.x;
  double* dst = S + ((t & 1) * 8) * LDS_STRIDE + (t >> 1);
#pragma unroll
  for (int q = 0; q < 8; ++q) dst[q * LDS_STRIDE] = r[q];
}

template <bool TA, bool TB>
__device__ __forceinline__ void gemm_f64_body(const GemmBatch& batch) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const GemmParams& p = batch.p[blockIdx.y];  // blockIdx.y selects the problem of a batched launch
  if ((int)blockIdx.x >= p.nblocks) return;
  // small launches of this engine sit on latency-bound chains (blocked factorisation / substitution) that may share CUs
  // with the GEMM-bound side stream: issue ahead of co-resident waves
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(2);
  double* As = smem;                          // [2][BK][LDS_STRIDE]
  double* Bs = smem + 2 * BK * LDS_STRIDE;    // [2][BK][LDS_STRIDE]

  const int split = blockIdx.x % p.splitk;
  const int tidx = blockIdx.x / p.splitk;
  int tm, tn;
  tile_from_index(tidx, p.tiles_m, p.tiles_n, p.tri, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * p.klen;
  const int kend = min(p.K, kbeg + p.klen);
  const int ktiles = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g4 = lane >> 4;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  double ra[8], rb[8];
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (TA) load_direct(p.A, p.lda, k0, kend, m0, p.M, p.vecA, ra);
    else load_transposing(p.A, p.lda, k0, kend, m0, p.M, p.vecA, ra);
    if (TB) load_transposing(p.B, p.ldb, k0, kend, n0, p.N, p.vecB, rb);
    else load_direct(p.B, p.ldb, k0, kend, n0, p.N, p.vecB, rb);
  };
  auto sstore = [&](int buf) {
    double* a = As + buf * BK * LDS_STRIDE;
    double* b = Bs + buf * BK * LDS_STRIDE;
    if (TA) store_direct(a, ra); else store_transposing(a, ra);
    if (TB) store_transposing(b, rb); else store_direct(b, rb);
  };

  if (ktiles > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();

  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ktiles;
    if (more) gload(kt + 1);
    const double* a_base = As + buf * BK * LDS_STRIDE + wm * 64 + r16;
    const double* b_base = Bs + buf * BK * LDS_STRIDE + wn * 64 + r16;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double a[4], b[4];
      const int krow = (ks * 4 + g4) * LDS_STRIDE;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = a_base[krow + i * 16];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = b_base[krow + j * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds D[row = g4 + 4*reg][col = r16] of each 16x16 tile
  if (p.slab != nullptr) {
    double* out = p.slab + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = m0 + wm * 64 + i * 16 + g4 + 4 * reg;
          const int col = n0 + wn * 64 + j * 16 + r16;
          if (row < p.M && col < p.N) out[(int64_t)row * p.N + col] = acc[i][j][reg];
        }
  } else {
    const bool mirror = (p.tri == TRI_UPPER_MIRROR) && (tm != tn);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = m0 + wm * 64 + i * 16 + g4 + 4 * reg;
          const int col = n0 + wn * 64 + j * 16 + r16;
          if (row < p.M && col < p.N) {
            double v = p.alpha * acc[i][j][reg];
            if (p.beta != 0.0) v += p.beta * p.C[(int64_t)row * p.ldc + col];
            p.C[(int64_t)row * p.ldc + col] = v;
            if (mirror) p.C[(int64_t)col * p.ldc + row] = v;
          }
        }
  }
}
template <bool TA, bool TB>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_f64_kernel(GemmBatch batch) { gemm_f64_body<TA, TB>(batch); }
template <bool TA, bool TB>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_f64_kernel_batched(const nk::ArgPack<GemmBatch>* table) {
  gemm_f64_body<TA, TB>(table[blockIdx.z].v);
}
#define NK_GEMM_TWIN(A, B, tag)                                                                                       \
  static nk::TwinReg gemm_twin_reg_##tag(reinterpret_cast<const void*>(static_cast<void (*)(GemmBatch)>(gemm_f64_kernel<A, B>)), \
                                         reinterpret_cast<const void*>(gemm_f64_kernel_batched<A, B>),                  \
                                         sizeof(nk::ArgPack<GemmBatch>), "gemm_f64_kernel_" #tag);
NK_GEMM_TWIN(false, false, nn) NK_GEMM_TWIN(false, true, nt) NK_GEMM_TWIN(true, false, tn) NK_GEMM_TWIN(true, true, tt)


// C = alpha * sum_s slab[s] + beta * C, honouring the triangular tile modes.
__device__ __forceinline__ void gemm_reduce_kernel_body(const double* __restrict__ slab, int splitk, int M, int N, double alpha, double beta, double* __restrict__ C, int64_t ldc, int tri) {
  const int64_t total = (int64_t)M * N;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(e / N), col = (int)(e - (int64_t)row * N);
    const int tr = row / BM, tc = col / BN;
    if (tri == TRI_UPPER_MIRROR && tr > tc) continue;
    if (tri == TRI_LOWER && tr < tc) continue;
    double s = 0.0;
    for (int k = 0; k < splitk; ++k) s += slab[(int64_t)k * total + e];
    double v = alpha * s;
    if (beta != 0.0) v += beta * C[(int64_t)row * ldc + col];
    C[(int64_t)row * ldc + col] = v;
    if (tri == TRI_UPPER_MIRROR && tr != tc) C[(int64_t)col * ldc + row] = v;
  }
}
__global__ void __launch_bounds__(256) gemm_reduce_kernel(const double* __restrict__ slab, int splitk, int M, int N, double alpha, double beta, double* __restrict__ C, int64_t ldc, int tri) { gemm_reduce_kernel_body(slab, splitk, M, N, alpha, beta, C, ldc, tri); }
NK_BATCHED_TWIN(gemm_reduce_kernel, (256), const double*, int, int, int, double, double, double*, int64_t, int)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Fill the device parameters of one problem (no split-K slab handling: splitk must resolve to 1 for batched use).
static int gemm_prepare(nk_ctx* ctx, bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                        const GemmOpts& opts, GemmParams* out, int* ntiles_out) {
  NK_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1LL << 31) && K >= 0, "nk_gemm: dimension out of range");
  GemmParams& p = *out;
  p.A = A; p.B = B; p.C = C; p.slab = nullptr;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  p.tri = opts.tri;
  if (p.tri == TRI_UPPER_MIRROR) NK_REQUIRE(M == N, "nk_gemm: the mirrored mode needs a square C");
  if (p.tri == TRI_LOWER) NK_REQUIRE(M >= N, "nk_gemm: the lower mode needs M >= N");
  p.alpha = alpha; p.beta = beta;
  p.vecA = aligned16(A) && (lda % 2 == 0);
  p.vecB = aligned16(B) && (ldb % 2 == 0);
  int ntiles = p.tiles_m * p.tiles_n;
  if (p.tri == TRI_UPPER_MIRROR) ntiles = p.tiles_m * (p.tiles_m + 1) / 2;
  if (p.tri == TRI_LOWER) ntiles = p.tiles_n * (p.tiles_n + 1) / 2 + (p.tiles_m - p.tiles_n) * p.tiles_n;
  const int ktiles_total = (int)((K + BK - 1) / BK);
  int splitk = opts.splitk;
  if (splitk <= 0) {
    const int target = 4 * ctx->num_cu;  // two resident workgroups per CU, two rounds
    splitk = ntiles >= target / 2 ? 1 : (target + ntiles - 1) / ntiles;
    if (splitk >= 6) splitk = ((splitk + 7) / 8) * 8;  // align slices with the 8 XCDs
    const int max_split = ktiles_total / 8 > 0 ? ktiles_total / 8 : 1;  // at least 8 k-steps per slice
    if (splitk > max_split) splitk = max_split;
    if (splitk > 64) splitk = 64;
    if (splitk < 1) splitk = 1;
  }
  p.splitk = splitk;
  p.klen = ((ktiles_total + splitk - 1) / splitk) * BK;
  if (p.klen == 0) p.klen = BK;
  p.nblocks = ntiles * splitk;
  *ntiles_out = ntiles;
  return NK_OK;
}

static int gemm_set_attrs() {
  static bool attr_set = false;
  if (!attr_set) {
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<false, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<true, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_batched<false, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_batched<false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_batched<true, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_batched<true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  return NK_OK;
}

static void gemm_dispatch(nk_ctx* ctx, bool transA, bool transB, dim3 grid, const GemmBatch& b) {
  dim3 block(GEMM_THREADS);
  if (transA && transB) hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, block, LDS_BYTES, ctx->stream, b);
  else if (transA) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, block, LDS_BYTES, ctx->stream, b);
  else if (transB) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, block, LDS_BYTES, ctx->stream, b);
  else hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, block, LDS_BYTES, ctx->stream, b);
}

// Two independent small problems with the same transposition flags in ONE launch (blockIdx.y = problem).  Used by the
// paired Cholesky factorisations / triangular solves of the fit, whose per-step kernels are latency bound.
int launch_gemm_pair(nk_ctx* ctx, bool transA, bool transB, const GemmCall* calls, int ncalls) {
  NK_REQUIRE(ncalls >= 1 && ncalls <= GEMM_MAX_BATCH, "gemm_pair: 1..2 problems");
  GemmBatch b;
  int maxblocks = 0, live = 0;
  for (int q = 0; q < GEMM_MAX_BATCH; ++q) {
    if (q < ncalls && calls[q].M > 0 && calls[q].N > 0) {
      GemmOpts o = calls[q].opts;
      o.splitk = 1;
      int nt = 0;
      NK_TRY(gemm_prepare(ctx, transA, transB, calls[q].M, calls[q].N, calls[q].K, calls[q].alpha, calls[q].A,
                          calls[q].lda, calls[q].B, calls[q].ldb, calls[q].beta, calls[q].C, calls[q].ldc, o, &b.p[q], &nt));
      if (b.p[q].nblocks > maxblocks) maxblocks = b.p[q].nblocks;
      ++live;
    } else {
      b.p[q] = GemmParams{};
      b.p[q].nblocks = 0;
    }
  }
  if (live == 0) return NK_OK;
  NK_TRY(gemm_set_attrs());
  gemm_dispatch(ctx, transA, transB, dim3((unsigned)maxblocks, (unsigned)GEMM_MAX_BATCH), b);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// Small products (lift of a few states, the product with C after a rollout, p-column blocks): the 128 x 128 engine
// would put the whole problem on one or two CUs and take 20-30 us; here every 16 x 16 output tile is a workgroup, operands
// addressed through (row, column) strides so that one kernel serves all four transposition cases.
constexpr int GS_BK = 64;  // contraction slice per step
__device__ __forceinline__ void gemm_small_kernel_body(int M, int N, int K, double alpha, const double* __restrict__ A, int64_t sai, int64_t sak, const double* __restrict__ B, int64_t sbk, int64_t sbj, double beta, double* __restrict__ C, int64_t ldc) {
  // These products are latency bound (a lift of one state at m = 500 is 32 dependent global-load round trips with
  // 16-wide slices: 22 us): 64-wide slices, the loads of slice s + 1 in flight while slice s is multiplied, and four
  // accumulators per entry (k mod 4) instead of one 64-deep FMA chain.
  __shared__ double As[16][GS_BK + 1];
  __shared__ double Bs[GS_BK][17];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  // element -> (row, k) assignment of the four loads per operand: the unit-stride direction is the fast one
  const bool a_kfast = sak == 1, b_jfast = sbj == 1;
  int ai[4], ak[4], bk[4], bj[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q;
    ai[q] = a_kfast ? e >> 6 : e & 15;
    ak[q] = a_kfast ? e & 63 : e >> 4;
    bk[q] = b_jfast ? e >> 4 : e & 63;
    bj[q] = b_jfast ? e & 15 : e >> 6;
  }
  double ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ia = i0 + ai[q], ka = k0 + ak[q];
      ra[q] = (ia < M && ka < K) ? A[(int64_t)ia * sai + (int64_t)ka * sak] : 0.0;
      const int kb = k0 + bk[q], jb = j0 + bj[q];
      rb[q] = (kb < K && jb < N) ? B[(int64_t)kb * sbk + (int64_t)jb * sbj] : 0.0;
    }
  };
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += GS_BK) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      As[ai[q]][ak[q]] = ra[q];
      Bs[bk[q]][bj[q]] = rb[q];
    }
    __syncthreads();
    if (k0 + GS_BK < K) fetch(k0 + GS_BK);  // in flight during the products below
#pragma unroll
    for (int k = 0; k < GS_BK; ++k) acc[k & 3] = fma(As[ty][k], Bs[k][tx], acc[k & 3]);
    __syncthreads();
  }
  const int i = i0 + ty, j = j0 + tx;
  if (i < M && j < N) {
    const double sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    double* c = C + (int64_t)i * ldc + j;
    *c = (beta == 0.0) ? alpha * sum : fma(alpha, sum, beta * *c);
  }
}
__global__ void __launch_bounds__(256) gemm_small_kernel(int M, int N, int K, double alpha, const double* __restrict__ A, int64_t sai, int64_t sak, const double* __restrict__ B, int64_t sbk, int64_t sbj, double beta, double* __restrict__ C, int64_t ldc) { gemm_small_kernel_body(M, N, K, alpha, A, sai, sak, B, sbk, sbj, beta, C, ldc); }
NK_BATCHED_TWIN(gemm_small_kernel, (256), int, int, int, double, const double*, int64_t, int64_t, const double*, int64_t, int64_t, double, double*, int64_t)

int launch_gemm(nk_ctx* ctx, bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, const GemmOpts& opts,
                float* ms_kernel) {
  if (M <= 0 || N <= 0) return NK_OK;
  if (opts.tri == TRI_FULL && opts.splitk == 0 && ms_kernel == nullptr && M * N <= 32768 && K <= 1024) {
    const int64_t sai = transA ? 1 : lda, sak = transA ? lda : 1;   // A(i, k)
    const int64_t sbk = transB ? 1 : ldb, sbj = transB ? ldb : 1;   // B(k, j)
    hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)((N + 15) / 16), (unsigned)((M + 15) / 16)), dim3(256), 0,
                       ctx->stream, (int)M, (int)N, (int)K, alpha, A, sai, sak, B, sbk, sbj, beta, C, ldc);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  if (transA && !transB && opts.tri != TRI_LOWER && opts.splitk == 0 && K >= 128) {
    // contraction-major operands: LDS-DMA engine (nk_gemm_tn.hip) when the alignment contract holds
    TnProblem tp;
    tp.A = A; tp.B = B; tp.C = C; tp.lda = lda; tp.ldb = ldb; tp.ldc = ldc; tp.M = (int)M; tp.N = (int)N;
    tp.tri = opts.tri; tp.alpha = alpha; tp.beta = beta;
    if (tn_fast_ok(tp) && (opts.tri == TRI_FULL || M == N)) return launch_gemm_tn_multi(ctx, &tp, 1, K, 0, ms_kernel);
  }
  GemmBatch b;
  b.p[1] = GemmParams{};
  b.p[1].nblocks = 0;
  int ntiles = 0;
  NK_TRY(gemm_prepare(ctx, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, opts, &b.p[0], &ntiles));
  GemmParams& p = b.p[0];
  const int splitk = p.splitk;
  const ArenaMark mark = arena_mark(ctx);
  if (splitk > 1) {
    double* slab = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)splitk * M * N, &slab));
    p.slab = slab;
  }
  NK_TRY(gemm_set_attrs());
  if (ms_kernel) NK_HIP(hipEventRecord(ctx->ev[14], ctx->stream));
  gemm_dispatch(ctx, transA, transB, dim3((unsigned)p.nblocks, 1), b);
  NK_HIP(hipGetLastError());
  if (ms_kernel) {
    NK_HIP(hipEventRecord(ctx->ev[15], ctx->stream));
  }
  if (splitk > 1) {
    const int64_t total = M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > ctx->num_cu * 16) blocks = ctx->num_cu * 16;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p.slab, splitk, p.M, p.N, alpha,
                       beta, C, ldc, p.tri);
    NK_HIP(hipGetLastError());
  }
  if (ms_kernel) {
    NK_HIP(hipEventSynchronize(ctx->ev[15]));
    NK_HIP(hipEventElapsedTime(ms_kernel, ctx->ev[14], ctx->ev[15]));
  }
  arena_release(ctx, mark);  // stream order makes the slab reusable by later launches
  return NK_OK;
}

}  // namespace nk
