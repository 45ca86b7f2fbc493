// Internal declarations shared by the libnyskoop translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "nyskoop.h"
#include "nk_lockstep.h"

namespace nk {

void set_error(const char* fmt, ...);

#define NK_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      nk::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? NK_ERR_OOM : NK_ERR_HIP;                            \
    }                                                                                        \
  } while (0)

#define NK_TRY(call)            \
  do {                          \
    int rc_ = (call);           \
    if (rc_ != NK_OK) return rc_; \
  } while (0)

#define NK_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      nk::set_error(__VA_ARGS__); \
      return NK_ERR_BAD_ARG;      \
    }                             \
  } while (0)

// Grow-only HBM workspace: a list of chunks with bump pointers, reset at the start of every API call (and then
// coalesced into one chunk, so the steady state performs no hipMalloc).  All work of a context is on one stream,
// so releasing back to a mark makes the space reusable by later launches in stream order.
struct ArenaChunk {
  char* base = nullptr;
  size_t cap = 0;
  size_t off = 0;
};
struct Arena {
  std::vector<ArenaChunk> chunks;
  int cur = 0;
};
struct ArenaMark {
  int chunk;
  size_t off;
};

}  // namespace nk

struct nk_member_state;
struct nk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;       // CURRENT stream: every launcher uses this (swapped by nk::SideScope)
  hipStream_t stream_main = nullptr;
  hipStream_t stream_side = nullptr;  // second stream (lowest priority) for GEMM-bound work beside the main chain
  hipStream_t stream_copy = nullptr;  // device->host copies of nk_model_get_ops_async (overlap with the next call's kernels)
  hipStream_t stream_prep = nullptr;  // third stream (highest priority) for small latency-bound work queued early: its
                                      // kernels must get CU slots while a big main-stream launch still has workgroups
                                      // pending (the dispatcher serves queues in strict priority order)
  nk::Arena arena;                    // workspace of the main stream
  nk::Arena arena_side;               // workspace of the side stream (slabs must not be shared across streams)
  nk::Arena* cur_arena = nullptr;
  hipStream_t stream_la[2] = {nullptr, nullptr};  // look-ahead streams of the blocked Cholesky: [0] beside the prep stream
                                                  // (highest priority), [1] beside the main stream (middle)
  hipEvent_t ev_la[2][4] = {};                    // per look-ahead stream: panel-done / rest-done events, two of each (ring)
  int compute_f32 = 0;                            // 1: kernel blocks and Gram contractions of a fit on the fp32 engine
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_chain = nullptr;
  int* d_info = nullptr;       // device flags for factorisation failures (one int per paired system)
  unsigned long long* d_piv = nullptr;  // [min, max] Cholesky pivot per system slot (bit patterns), behind d_info's 4 slots
  unsigned long long* h_piv = nullptr;  // pinned host mirror
  double* d_scalars = nullptr; // small device scratch for reductions (64 doubles)
  double* h_scalars = nullptr; // pinned host mirror
  int* h_info = nullptr;       // pinned host mirror of d_info (kept apart from h_scalars: both streams may be in flight)
  double* d_zeros = nullptr;   // 4 KiB zero page (K-tail rows of the LDS-DMA GEMM)
  hipEvent_t ev[16];
  int num_cu = 256;
  int kmat_mode = 0;  // 0 auto (Gram form on MFMA for d >= 32), 1 always direct differences (NYSKOOP_KMAT=direct)
  // optional refinement of the two regularised solves with doubled-precision residuals (nk_set_refine): applied to a system
  // whose smallest / largest Cholesky pivot is below refine_pivot (0 = never, the default), refine_steps steps
  double refine_pivot = 0.0;
  int refine_steps = 2;
  int strict_spd = 0; // 1: a non-positive Cholesky pivot is an error (NK_ERR_NOT_SPD) instead of entering the
                      // rank-truncating pseudo-inverse path (NYSKOOP_STRICT_SPD=1 / nk_set_strict_spd)
  hipEvent_t ev_ext = nullptr;  // ordering against a caller's stream (nk_wait_stream)
  hipEvent_t ev_up[8] = {};     // uploads of the row blocks of a fit from host arrays (pipelined with the passes)
  nk_group* group = nullptr;    // member of a lock-step group (nk_lockstep.h): stream operations are recorded and merged
  nk_member_state* gstate = nullptr;
  double* h_stage = nullptr;    // page-locked, device-visible staging block for the small latency-bound calls (rollouts):
  size_t h_stage_bytes = 0;     // kernels read their inputs from it and write their results into it directly (no DMA)
};

struct nk_model {
  int device = 0;
  int32_t m = 0, d = 0, p = 0;
  int32_t ktype = 0;
  double sigma0 = 0.0;
  double jitter = 0.0;
  double* buf = nullptr;  // one allocation holding everything below
  size_t bytes = 0;
  double *A = nullptr, *B = nullptr, *C = nullptr, *W = nullptr, *S = nullptr, *Sinv = nullptr, *Z = nullptr,
         *winv = nullptr;  // winv: 1/lengthscale per dimension (d entries)
  bool has_ops = false;
  hipEvent_t ev_fetch = nullptr;  // completion of the last nk_model_get_ops_async (nullptr: nothing pending)
};

namespace nk {

// Route launches and workspace allocations to the side stream (or the given one; the side arena is shared, so work on the
// two auxiliary streams must be ordered by events) for the lifetime of the scope.
struct SideScope {
  nk_ctx* c;
  hipStream_t s0;
  Arena* a0;
  explicit SideScope(nk_ctx* ctx, hipStream_t s = nullptr) : c(ctx), s0(ctx->stream), a0(ctx->cur_arena) {
    c->stream = s ? s : c->stream_side;
    c->cur_arena = &c->arena_side;
  }
  ~SideScope() {
    c->stream = s0;
    c->cur_arena = a0;
  }
};

// factorisation failure flags: slots 0-1 belong to the main stream, 2-3 to the side stream (each stream may have a paired
// factorisation in flight)
inline int info_base(const nk_ctx* c) { return c->stream != c->stream_main ? 2 : 0; }

// the highest coefficient of exp_nonpos (0x3e5ade156a5dcb37) held in a vector register pair the compiler cannot re-materialise
__device__ __forceinline__ double exp_nonpos_ca() {
  double ca = __longlong_as_double(0x3e5ade156a5dcb37LL);
  asm volatile("" : "+v"(ca));
  return ca;
}
// exp(x) for x <= 0, the arithmetic of the device library's exp (same range reduction, same degree-11 Horner form with the same
// coefficients, same operation order: the same bits) -- but with the coefficients as SCALAR operands of the fused multiply-adds.
// The compiler's form keeps the accumulator in the destination (v_fmac_f64) and re-materialises every coefficient into a
// vector register pair in front of it: 18 v_mov_b32 per value, plus an upper range check that a non-positive argument
// does not need.  In the epilogue of the Gram-form kernel blocks every vector instruction
// takes issue slots from the matrix pipe the other workgroup of the CU is using: 42 -> 21 instructions per exp.
__device__ __forceinline__ double exp_nonpos(double x, double ca_v) {
  const double k = __builtin_rint(x * __longlong_as_double(0x3ff71547652b82feLL));   // log2(e)
  double r = __builtin_fma(__longlong_as_double((long long)0xbfe62e42fefa39efULL), k, x);   // -ln2, high part
  r = __builtin_fma(__longlong_as_double((long long)0xbc7abc9e3b39803fULL), k, r);          // -ln2, low part
  double p;
#define NK_EXP_STEP(c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(r), "v"(p), "s"(c))
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(r), "v"(ca_v), "s"(__longlong_as_double(0x3e928af3fca7ab0cLL)));
  NK_EXP_STEP(__longlong_as_double(0x3ec71dee623fde64LL));
  NK_EXP_STEP(__longlong_as_double(0x3efa01997c89e6b0LL));
  NK_EXP_STEP(__longlong_as_double(0x3f2a01a014761f6eLL));
  NK_EXP_STEP(__longlong_as_double(0x3f56c16c1852b7b0LL));
  NK_EXP_STEP(__longlong_as_double(0x3f81111111122322LL));
  NK_EXP_STEP(__longlong_as_double(0x3fa55555555502a1LL));
  NK_EXP_STEP(__longlong_as_double(0x3fc5555555555511LL));
  NK_EXP_STEP(__longlong_as_double(0x3fe000000000000bLL));
#undef NK_EXP_STEP
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  // (the library's lower range check, which also keeps a NaN argument a NaN; its upper one cannot fire for x <= 0)
  return x < -1075.0 ? 0.0 : ldexp(p, (int)k);
}
__device__ __forceinline__ double exp_nonpos(double x) { return exp_nonpos(x, exp_nonpos_ca()); }

// ---- lock-step groups (nk_group.hip) ----------------------------------------------------------------------------
nk_group* group_new(int device, int size);
void group_attach(nk_group* g, int slot, nk_ctx* c);
void group_detach(nk_ctx* c);
int group_enter(nk_ctx* c);
int group_leave(nk_ctx* c);
void group_stats(nk_ctx* c, uint64_t out[4]);
// slow-path counters (nk_runtime_counters)
enum { CNT_CHAIN_GIVEUP = 0, CNT_JACOBI_GIVEUP = 1, CNT_RANK_TRUNCATED = 2, CNT_SQRT_RETRY = 3, CNT_REFINED = 4, CNT_N = 5 };
void count_event(int which);
uint64_t read_counter(int which);
hipError_t real_stream_sync(hipStream_t s);
hipError_t real_event_sync(hipEvent_t e);

// ---- workspace -------------------------------------------------------------------------------------------
int arena_reset(nk_ctx* ctx);
ArenaMark arena_mark(nk_ctx* ctx);
void arena_release(nk_ctx* ctx, ArenaMark mk);
int arena_alloc(nk_ctx* ctx, size_t bytes, void** out);
template <typename T>
inline int arena_alloc_t(nk_ctx* ctx, size_t count, T** out) {
  return arena_alloc(ctx, count * sizeof(T), reinterpret_cast<void**>(out));
}
bool is_device_ptr(const void* p);

// ---- device launchers (all asynchronous on ctx->stream; device pointers only) ---------------------------------
// kernel matrix out[i][j] = k(A[i,:], B[j,:]); winv = 1/lengthscale (d entries, device)
int launch_kmat(nk_ctx* ctx, int ktype, const double* A, int64_t lda, int64_t nA, const double* B, int64_t ldb,
                int64_t nB, int d, const double* winv, double sigma0, double* out, int64_t ldo);

enum { TRI_FULL = 0, TRI_UPPER_MIRROR = 1, TRI_LOWER = 2 };
struct GemmOpts {
  int tri = TRI_FULL;  // TRI_UPPER_MIRROR: compute tiles with tn >= tm and mirror; TRI_LOWER: tiles tn <= tm only
  int splitk = 0;      // 0 = choose automatically
};
// C = alpha*op(A)*op(B) + beta*C ; transA: A stored KxM ; transB: B stored NxK
int launch_gemm(nk_ctx* ctx, bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                const GemmOpts& opts = GemmOpts(), float* ms_kernel = nullptr);

struct GemmCall {
  int64_t M = 0, N = 0, K = 0;
  double alpha = 1.0, beta = 0.0;
  const double* A = nullptr;
  const double* B = nullptr;
  double* C = nullptr;
  int64_t lda = 0, ldb = 0, ldc = 0;
  GemmOpts opts;
};
// up to two independent problems with the same transposition flags in one launch (no split-K)
int launch_gemm_pair(nk_ctx* ctx, bool transA, bool transB, const GemmCall* calls, int ncalls);

struct TnSkip;  // device-side launch control, defined with the TN engine below
// elementwise / reductions
int launch_add_diag(nk_ctx* ctx, double* A, int64_t lda, int n, double v);
int launch_copy2d(nk_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols);
int launch_axpby2d(nk_ctx* ctx, double a, const double* X, int64_t ldx, double b, double* Y, int64_t ldy,
                   int64_t rows, int64_t cols);  // Y = a*X + b*Y
int launch_scale_add_identity(nk_ctx* ctx, double a, const double* X, int64_t ldx, double c, double* Y, int64_t ldy,
                              int n, const TnSkip* skip = nullptr);  // Y = a*X + c*I
int launch_fill(nk_ctx* ctx, double* A, int64_t lda, int64_t rows, int64_t cols, double v);
int launch_frob_minus_identity(nk_ctx* ctx, const double* M, int64_t ldm, int n, double* d_out,
                               const TnSkip* skip = nullptr);  // sum (M-I)^2
int launch_max_abs_rowsum(nk_ctx* ctx, const double* M, int64_t ldm, int n, double* d_out);
int launch_colsum_sqdiff(nk_ctx* ctx, const double* P, int64_t ldp, const double* Y, int64_t ldy, int64_t rows,
                         int cols, double* d_colsum);  // colsum[j] = sum_i (P[i][j]-Y[i][j])^2

// dense SPD machinery built on the GEMM engine
int sqrtm_spd(nk_ctx* ctx, const double* P, int64_t ldp, int m, double* S, double* Sinv, int* iters, double* resid);
// the same in two phases, so that a caller can queue the latency-bound half (Cholesky factor of P and its inverse) early
// and the GEMM-bound half (Newton-Schulz polar iteration) later, both on the current stream; no host synchronisation in
// sqrtm_prepare.  The plan's buffers live in the current arena until sqrtm_finish returns.
struct SqrtPlan {
  const double* P = nullptr;
  int64_t ldp = 0;
  int m = 0;
  double* W = nullptr;     // 2m x m: lower Cholesky factor L of P on top, L^-T below
  double* Linv = nullptr;  // inverted diagonal blocks
  double* X0 = nullptr;    // L^T / sqrt(c)   (c = ||P||_inf)
  double* X0t = nullptr;   // L / sqrt(c)
  double* d_sc = nullptr;  // device scalars: c, ||P||_F^2, trace(P), ||L^-1||_F^2; then the iteration state
                           // [5] convergence flag (step + 1), [6] residual at that step, [7] last residual
  ArenaMark mark{0, 0};
  // verdict of sqrtm_finish: known at once for the synchronous forms, otherwise in ctx->h_scalars[8..10] once the
  // stream has been synchronised (sqrtm_verdict)
  bool deferred = false;
  int rc = 0, iters = 0, kmax = 0;
  double resid = 0.0;
  // Optional rigorous lower bound of the smallest eigenvalue of P known to the caller (the jitter of K_mm + jitter I):
  // with it the step budget needs nothing from the factorisation, so sqrtm_finish queues the iteration without waiting
  // for sqrtm_prepare's kernels (`early`); a failed factorisation or an exhausted budget then surfaces in
  // sqrtm_verdict as NK_SQRT_RETRY and the caller falls back to sqrtm_spd_coupled.
  double lambda_min_hint = 0.0;
  bool early = false;
  // The factorisation chain of sqrtm_prepare is paused before block step `pause_step` until `pause_event` (recorded by
  // the caller BEFORE sqrtm_prepare is called) has completed -- see nk_nystrom_fit: the chain must not run beside the
  // fused Gram launch.
  hipEvent_t pause_event = nullptr;
  int pause_step = -1;
};
constexpr int NK_SQRT_RETRY = 1;  // internal (positive) verdict: redo the square root with the coupled iteration
int sqrtm_spd_coupled(nk_ctx* ctx, const double* P, int64_t ldp, int m, double* S, double* Sinv, int* iters, double* resid);
int sqrtm_prepare(nk_ctx* ctx, const double* P, int64_t ldp, int m, SqrtPlan* plan);
int sqrtm_finish(nk_ctx* ctx, SqrtPlan* plan, double* S, double* Sinv);
// after the stream that ran sqrtm_finish has been synchronised: NK_OK / NK_ERR_NO_CONVERGENCE, iteration count, residual
int sqrtm_verdict(nk_ctx* ctx, SqrtPlan* plan, int* iters, double* resid);
// in-place lower Cholesky of P (m x m, ld), Linv workspace holds inverted diagonal blocks
int cholesky_lower(nk_ctx* ctx, double* P, int64_t ldp, int m, double* Linv /* nblk*NB*NB */);
int cholesky_solve(nk_ctx* ctx, const double* L, int64_t ldl, int m, const double* Linv, double* R, int64_t ldr,
                   int nrhs);  // R <- (L L^T)^{-1} R in place
// the same for TWO independent systems advanced in lock step, every per-block kernel launched once for both
// (the chains are latency bound: pairing halves their length)
struct CholSys {
  double* P = nullptr;   // matrix, overwritten by its lower factor
  int64_t ldp = 0;
  int m = 0;
  // Augmented form (cholesky_aug_pair): `extra` further rows below the m x m matrix hold the right-hand sides
  // TRANSPOSED (extra x m); the factorisation's panel / trailing updates carry them along (= forward substitution for
  // free) and a backward pass leaves X^T = (P^-1 R)^T = R^T P^-1 in their place.
  int extra = 0;
  bool backward = true;  // false: stop after the factorisation (the extra rows then hold R^T L^-T)
  double* Linv = nullptr;
  double* pivlog = nullptr;  // optional device array of m doubles: the factorisation logs every pivot here
  double* R = nullptr;   // right-hand sides (solve only), overwritten by the solution
  int64_t ldr = 0;
  int nrhs = 0;
};
int cholesky_lower_pair(nk_ctx* ctx, const CholSys* sys, int nsys);
int cholesky_lower_pair_async(nk_ctx* ctx, const CholSys* sys, int nsys);  // no host synchronisation
int cholesky_check_pair(nk_ctx* ctx, const CholSys* sys, int nsys);       // verdict of the async factorisation
int cholesky_solve_pair(nk_ctx* ctx, const CholSys* sys, int nsys);
// factor + both substitutions, no host sync; mark / mark_step: optional event recorded after block step `mark_step`
// pause / pause_step: before block step pause_step the chain's stream waits for the (already recorded) event `pause`
int cholesky_aug_pair_async(nk_ctx* ctx, const CholSys* sys, int nsys, hipEvent_t pause = nullptr, int pause_step = -1);
constexpr int CHOL_NB = 64;
// workspace per diagonal block of a blocked factorisation (CholSys::Linv): the inverted 64 x 64 diagonal block of the factor
// followed by the diagonal block itself, both dense row-major with exact zeros above the diagonal and identity padding
// beyond a short last block.  The factor rides along because every product with the explicit inverse is followed by one
// correction step  x += (b - x L_jj) L_jj^-1  (see chol_panel_kernel): multiplying by an explicit inverse alone is not
// backward stable -- its error grows with cond(L_jj), 1e6 on the kernel matrices here.
constexpr int CHOL_WS = 2 * CHOL_NB * CHOL_NB + 8;  // (+ one verdict word, padded to 64 bytes)
// the verdict word behind the two blocks: 1.0 when ||L_jj||_F ||L_jj^-1||_F exceeds this and the products with the inverse
// take the correction step (potrf_diag_kernel)
constexpr double CHOL_FIX_KAPPA = 8.0 * CHOL_NB;
// NYSKOOP_CHOL_FIX (read per launch: same-process A/B; for measurements only): 0 = never correct, 2 = always, default 1 = where
// the verdict word says so
inline int chol_fix_enabled() {
  const char* e = getenv("NYSKOOP_CHOL_FIX");
  return (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1);
}
// trailing update C -= P P^T (K = 64) of up to two systems (nk_trail.hip); false: not that shape, use launch_gemm_pair
bool launch_chol_trail_pair(nk_ctx* ctx, const GemmCall* calls, int ncalls, int* rc);
// ... fused with the diagonal-block factorisation of the NEXT block step `blk` (nb[q]: its order, 0 = system q has none; the
// block is calls[q].C): one launch instead of two, the diagonal kernel off the critical path.  false: not that shape, nothing
// launched (the caller issues the two launches separately)
bool launch_chol_trail_potrf_pair(nk_ctx* ctx, const GemmCall* calls, int ncalls, const int* nb, double* const* Linv, int blk,
                                  double* const* plog, int* rc);
inline bool chol_fuse_enabled() {  // NYSKOOP_CHOL_FUSE=0 (read per call): separate launches, for A/B runs and the bit-identity test
  const char* e = getenv("NYSKOOP_CHOL_FUSE");
  return !(e && e[0] == '0');
}
// panel product P <- P Linv_jj^T (64 x 64) of up to two systems (nk_trail.hip); false: not that shape
bool launch_chol_panel_pair(nk_ctx* ctx, const GemmCall* calls, int ncalls, int* rc);
// E_q <- E_q L_q^-1 on the extra rows of up to two factored systems, one launch (nk_trsm.hip)
int launch_trsm_right_lower_pair(nk_ctx* ctx, const CholSys* sys, int nsys);

// E_out (rows x m) = E * pinv(P) for symmetric P with LAPACK gelsd's cut-off (singular values <= rcond * sigma_max are
// dropped): one-sided Jacobi SVD on the device (nk_pinv.hip), the fallback of the regularised solves
struct PinvInfo {
  int rank = 0, sweeps = 0;
  bool converged = false;
  double sigma_max = 0.0, sigma_min_kept = 0.0, sigma_min = 0.0;
};
int pinv_right_divide(nk_ctx* ctx, const double* P, int64_t ldp, int m, const double* E, int64_t lde, int rows,
                      double* E_out, int64_t ldeo, double rcond, PinvInfo* info);
// which systems of the last (paired) factorisation on the current stream met a non-positive pivot (synchronises)
// piv_ratio (optional, nsys entries): smallest / largest pivot of each factorisation (0 when it failed)
int cholesky_fail_flags(nk_ctx* ctx, const CholSys* sys, int nsys, int* failed /* nsys entries */, double* piv_ratio = nullptr);

// matrix-vector step of the lifted recursion for up to 8 trajectories (nk_rollout.hip)
int launch_lifted_step(nk_ctx* ctx, const double* G, int64_t ldg, int m, int mz, int pu, const double* z, int64_t zstride,
                       const double* u, int64_t ustride, const double* bias, double* out, int64_t ostride, int batch);

// the whole recursion z_{t+1} = G [z_t; u_t] + bias for a batch of trajectories in ONE launch, G resident in LDS (m <= 128);
// optionally preceded by the lift of the initial states by the same workgroups (nk_rollout.hip)
struct ChainArgs {
  const double* G = nullptr; int64_t ldg = 0; int m = 0, pu = 0;
  const double* z0 = nullptr; int64_t z0_stride = 0;
  bool lift = false;
  const double* x0 = nullptr; int64_t x0_stride = 0;
  const double* Zl = nullptr; int d = 0; const double* winv = nullptr; const double* Sinv = nullptr; int ktype = 0;
  double sigma0 = 0.0;
  const double* U = nullptr; int64_t u_stride = 0;
  const double* bias = nullptr; int64_t bias_stride = 0;
  double* Zall = nullptr; int64_t z_stride = 0;
  int T = 0, batch = 0;
};
int launch_ref_minus_traj(nk_ctx* ctx, const double* ref, int64_t ref_stride, const double* Phi, int64_t phi_stride,
                          double* D, int64_t d_stride, int steps, int m, int batch);
bool lifted_chain_ok(int m, int pu, int d_lift);
int launch_lifted_chain(nk_ctx* ctx, const ChainArgs& a);
// m > 128: multi-workgroup recursion, one launch per group of trajectories that is resident at once (nk_rollout.hip)
bool lifted_chain_mw_ok(const nk_ctx* ctx, int m, int pu);
int chain_mw_workgroups(int m);
int chain_mw_group(int m, int batch);  // trajectories one workgroup advances together
int launch_lifted_chain_mw(nk_ctx* ctx, const ChainArgs& a);
int lifted_chain_mw_reset(nk_ctx* ctx);                                       // clears the device status words
int lifted_chain_mw_fetch_status(nk_ctx* ctx);                                // queues their copy to the host
bool lifted_chain_mw_timed_out(nk_ctx* ctx, int* row, int* step, int* traj);  // after the stream has been synchronised

}  // namespace nk

namespace nk {
// ---- fast TN multi-problem GEMM (LDS-DMA staged), see nk_gemm_tn.hip -------------------------------------------
// C_p[M_p x N_p] = alpha_p * A_p^T B_p + beta_p * C_p for up to 4 problems that share the contraction length K:
// A_p stored K x M_p (lda), B_p stored K x N_p (ldb).  One launch covers every 128x128 tile of every problem.
enum { KTRIM_NONE = 0, KTRIM_B_UPPER = 1 /* B (K x N) upper triangular */, KTRIM_A_LOWER = 2 /* A (K x M) lower triangular */ };
struct TnProblem {
  const double* A = nullptr;
  const double* B = nullptr;
  double* C = nullptr;
  int64_t lda = 0, ldb = 0, ldc = 0;
  int M = 0, N = 0;
  int tri = TRI_FULL;  // TRI_UPPER_MIRROR for symmetric products (A == B)
  int ktrim = 0;       // KTRIM_B_UPPER / KTRIM_A_LOWER: a triangular operand, the k range where it is zero is skipped
  double alpha = 1.0, beta = 0.0;
  double* Ct = nullptr;  // optional: also store the transpose, Ct[col][row] = C[row][col] (N x M, leading dim ldct)
  int64_t ldct = 0;
  const double* A_even = nullptr;  // operand selection (TnSkip::select): alternatives of A / B for an even step count
  const double* B_even = nullptr;
  double* Caff = nullptr;  // optional second output Caff = aff_a * C + aff_c * I (same leading dimension as C)
  double aff_a = 0.0, aff_c = 0.0;
};
// Device-side control of a launch, so that an iteration with a data-dependent length can be queued in full without host
// round trips.  state: the launch returns at once when state[0] != 0 && state[0] <= step (state[0] = number of steps after
// which the iteration converged, written on the device).  select: the parity of select[0] picks A/B (odd) or
// A_even/B_even (even) -- the buffer of a ping-pong iteration that holds the final iterate.
struct TnSkip {
  const double* state = nullptr;
  int step = 0;
  const double* select = nullptr;
  // single symmetric product C: the reduce kernel also leaves per-workgroup partial sums of (C - I)^2 here (at least
  // 8 * tiles entries); *resid_count receives how many
  double* resid_partials = nullptr;
  int* resid_count = nullptr;
};
bool tn_fast_ok(const TnProblem& p);  // alignment / leading-dimension requirements of the LDS-DMA path
// fp32 engine (nk_gemm_tn_f32.hip): operands fp32 contraction-major, results fp64
struct TnProblemF {
  const float* A = nullptr;
  const float* B = nullptr;
  double* C = nullptr;
  int64_t lda = 0, ldb = 0, ldc = 0;
  int M = 0, N = 0;
  int tri = TRI_FULL;
  double beta = 0.0;
};
bool tnf_fast_ok(const TnProblemF& p);
int launch_gemm_tn_f32_multi(nk_ctx* ctx, const TnProblemF* probs, int nprob, int64_t K, int splitk /*0=auto*/,
                             float* ms_kernel = nullptr, bool sync_timing = true);
int prep_rows_f32(nk_ctx* ctx, const double* X, int64_t ldx, int64_t rows, int d, const double* winv, const double* center,
                  float* Xt, int64_t ldt, float* sq);
int launch_cvt_f64_f32(nk_ctx* ctx, const double* src, int64_t lds_, float* dst, int64_t ldd, int64_t rows, int cols);
int launch_kmat_gram_f32(nk_ctx* ctx, int ktype, const float* At, int64_t ldat, const float* sqa, int64_t nA, const float* Bt,
                         int64_t ldbt, const float* sqb, int64_t nB, int d, double sigma0, float* out, int64_t ldo);
int launch_gemm_tn_multi(nk_ctx* ctx, const TnProblem* probs, int nprob, int64_t K, int splitk /*0=auto*/,
                         float* ms_kernel = nullptr, bool sync_timing = true, const TnSkip* skip = nullptr);
int launch_transpose(nk_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int rows, int cols);
// X += dX if the refinement still contracts (verdict on the device, in state[4]: alive, |dX|^2 last accepted, accepted
// steps, |dX_0| / |X|); no host synchronisation
constexpr int refine_partial_blocks() { return 512; }
int launch_refine_apply(nk_ctx* ctx, const double* dX, int64_t ldd, double* X, int64_t ldx, int64_t rows, int64_t cols, int step,
                        double* state, double* partial);
// Res (nr x mq) = R - X P, dot products in doubled precision (compensated): the residual of the refinement steps
int launch_resid_dd(nk_ctx* ctx, const double* X, int64_t ldx, const double* P, int64_t ldp, const double* R, int64_t ldr,
                    double* Res, int64_t ldres, int nr, int mq);
// Gram-form kernel matrix on the MFMA engine (nk_gemm_tn.hip): prep_rows centres/scales/transposes rows to
// contraction-major and returns their squared norms; launch_kmat_gram evaluates k() in the GEMM epilogue
int launch_colmean(nk_ctx* ctx, const double* Z, int64_t ldz, int rows, int d, double* mean);
int prep_rows(nk_ctx* ctx, const double* X, int64_t ldx, int64_t rows, int d, const double* winv, const double* center,
              double* Xt, int64_t ldt, double* sq);
int launch_kmat_gram(nk_ctx* ctx, int ktype, const double* At, int64_t ldat, const double* sqa, int64_t nA,
                     const double* Bt, int64_t ldbt, const double* sqb, int64_t nB, int d, double sigma0, double* out,
                     int64_t ldo);
}  // namespace nk

// ---- cross-lane helpers for fp64 on gfx950 (no LDS traffic): DPP moves inside a 16-lane row, v_permlane16_swap /
//      v_permlane32_swap between rows and half waves (checked lane by lane on the device by tools/reduce_probe.hip)
#if defined(__HIPCC__)
namespace nk {
typedef unsigned chain_u2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// rows 1 and 3 (lanes 16-31, 48-63) of a change places with rows 0 and 2 of b
__device__ __forceinline__ void swap16_f64(double& a, double& b) {
  const chain_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const chain_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x);
  b = __hiloint2double((int)hi.y, (int)lo.y);
}
// lanes 32-63 of a change places with lanes 0-31 of b
__device__ __forceinline__ void swap32_f64(double& a, double& b) {
  const chain_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const chain_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x);
  b = __hiloint2double((int)hi.y, (int)lo.y);
}
// sum of one value over the 64 lanes, in every lane, without the LDS crossbar (a __shfl_down reduction is six dependent
// ds_bpermute round trips)
__device__ __forceinline__ double wave_sum64_dpp(double c) {
  c += dpp_f64<0xB1>(c);   // quad_perm [1,0,3,2]
  c += dpp_f64<0x4E>(c);   // quad_perm [2,3,0,1]
  c += dpp_f64<0x141>(c);  // row_half_mirror
  c += dpp_f64<0x140>(c);  // row_mirror: every lane of a 16-lane row holds the row total
  double a = c, b = c;
  swap16_f64(a, b);
  c = a + b;               // rows 0+1, 2+3
  a = c; b = c;
  swap32_f64(a, b);
  return a + b;
}

}  // namespace nk
#endif
