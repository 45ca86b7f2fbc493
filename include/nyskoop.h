/*
 * nyskoop.h -- C-ABI of libnyskoop.so: MI355X (gfx950) Nystrom-Koopman regression hot path.
 *
 * The reference (LCSL/nys-koop-lqr) has no FFI: its boundary is the Python class surface of
 * regressors.py.  Each entry point below names the reference interface it replaces (file:line into
 * /root/reference).  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - All matrices are row-major float64 with an explicit leading dimension (in elements), so strided views
 *     such as X[:, :d] of an n x (d+p) array are passed without copying (regressors.py:52,142).
 *   - Every data pointer may be a HOST pointer or a DEVICE (HIP) pointer; the library detects which
 *     (hipPointerGetAttributes).  Host buffers are staged through HBM by the library; device buffers are used
 *     in place.  All buffers are caller-owned; the library never retains a caller pointer after return.
 *   - Return value: NK_OK (0) or a negative NK_ERR_* code; nk_last_error() gives a thread-local message.
 *     No exceptions or aborts cross the ABI.
 *   - One nk_ctx per (thread, device).  Calls on distinct contexts are re-entrant; a context is not
 *     thread-safe.  A context owns its HIP streams (nk_stream() returns the main one) and a grow-only HBM workspace.
 *   - There is NO CPU fallback: without a usable HIP device nk_create fails with NK_ERR_NO_DEVICE.
 */
#ifndef NYSKOOP_H
#define NYSKOOP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NK_ABI_VERSION 2

enum {
  NK_OK = 0,
  NK_ERR_BAD_ARG = -1,        /* NULL / negative size / lengthscale-dimension mismatch (sklearn ValueError) */
  NK_ERR_HIP = -2,            /* a HIP runtime call failed */
  NK_ERR_NOT_SPD = -3,        /* Cholesky met a non-positive pivot and strict mode is on (nk_set_strict_spd); by
                                 default the solve falls back to gelsd's rank-truncated minimum-norm solution */
  NK_ERR_OOM = -4,            /* HBM allocation failed */
  NK_ERR_NO_CONVERGENCE = -5, /* matrix square-root iteration did not converge */
  NK_ERR_NO_DEVICE = -6       /* no gfx950 device visible */
};

/* kernel families: regressors.py:15-22 (RBF, anisotropic), :24-26 (Matern nu=2.5), :28-30 (DotProduct) */
enum { NK_KERNEL_RBF = 0, NK_KERNEL_MATERN52 = 1, NK_KERNEL_LINEAR = 2 };

typedef struct nk_kernel_desc {
  int32_t type;              /* NK_KERNEL_* */
  int32_t d;                 /* state dimension the kernel acts on */
  int32_t n_lengthscale;     /* 1 (isotropic) or d (anisotropic); ignored for LINEAR */
  int32_t reserved;
  const double* lengthscale; /* HOST pointer, n_lengthscale entries */
  double sigma0;             /* LINEAR only: k(x,y) = x.y + sigma0^2 */
} nk_kernel_desc;

/* per-fit diagnostics, all times in milliseconds measured with HIP events on the context's stream */
typedef struct nk_fit_stats {
  double ms_total;     /* whole nk_nystrom_fit call, device side (main stream, first to last event) */
  double ms_upload;    /* host->HBM staging and K(Z,Z) (0 when inputs are device pointers) */
  double ms_kmat;      /* the two n x m kernel blocks (of the first pass when the rows are processed in passes) */
  double ms_gram;      /* the fused Gram launch (+ kernel blocks and Gram launches of later passes) */
  double ms_sqrt;      /* matrix square root of K_mm on the side stream; OVERLAPS the factorisation chain */
  double ms_solve;     /* operator products after the two streams join (the factorisations and substitutions run
                          between ms_gram and this stage, concurrently with ms_sqrt) */
  double ms_gram_kernel_avg; /* average duration of one fused Gram launch (HIP events around the kernel) */
  int32_t gram_kernel_launches;
  int32_t sqrt_iters;
  double sqrt_residual;      /* ||X^T X - I||_F / sqrt(m) at the last convergence check (< 1e-7; the returned
                                square root is one quadratically convergent step beyond that iterate) */
  double gram_flops;         /* algorithmic flop of the Gram contractions actually issued */
  double kmat_pairs;         /* number of (row,row,dim) triples evaluated by the kernel-matrix builds */
  /* numerical rank used for the two regularised systems (regressors.py:155,165): m+p and m when the Cholesky
   * factorisations succeeded (full rank), otherwise the number of singular values kept by the pseudo-inverse path -- what
   * scipy.linalg.lstsq (gelsd) reports as `rank` */
  int32_t rank_inner;
  int32_t rank_inner_rec;
  /* smallest / largest Cholesky pivot of the two regularised systems (0 when the factorisation failed); below the
   * threshold of nk_set_refine (off by default) the solve is refined with doubled-precision residuals while the corrections contract;
   * `refined` = steps applied to inner + 16 x steps applied to inner_rec (0 = none); refine_ratio_* = |first correction| /
   * |solution| (an estimate of cond x backward error of the factor; the first step is applied when it is <= 1/4) */
  double pivot_ratio_inner;
  double pivot_ratio_inner_rec;
  int32_t refined;
  int32_t reserved_;
  double refine_ratio_inner;
  double refine_ratio_inner_rec;
} nk_fit_stats;

typedef struct nk_ctx nk_ctx;
typedef struct nk_model nk_model;

/* ---- library / context ------------------------------------------------------------------------------ */
int nk_version(void);
const char* nk_last_error(void);
int nk_device_count(void);
int nk_create(int device, nk_ctx** out);
int nk_destroy(nk_ctx* ctx);
int nk_synchronize(nk_ctx* ctx);
/* the hipStream_t all work of this context is launched on (for event timing by the caller) */
void* nk_stream(nk_ctx* ctx);
/* how nk_nystrom_fit builds the two n x m kernel blocks: 0 = automatic (Gram form |a|^2+|b|^2-2ab on the MFMA engine
 * when d >= 32, direct differences otherwise), 1 = always direct differences like scipy cdist (regressors.py:141-142
 * -> sklearn -> cdist).  K(Z,Z), lift queries and nk_kernel_matrix always use direct differences.
 * Also settable with the environment variable NYSKOOP_KMAT=direct before nk_create. */
int nk_set_kmat_mode(nk_ctx* ctx, int mode);
/* Rank-deficient regularised systems.  scipy.linalg.lstsq (regressors.py:155,165; LAPACK gelsd, rcond = eps) silently
 * returns the minimum-norm solution with singular values <= eps * sigma_max dropped.  By default (strict = 0) the library
 * does the same -- a one-sided Jacobi SVD on the device with the same cut-off -- whenever its Cholesky factorisation meets
 * a non-positive pivot or an isolated cluster of rounding-level pivots (<= 8 order eps d_max, separated from the other pivots
 * by a factor 1000: an exact null space), i.e. a matrix that is singular to working precision.  (A system whose Cholesky succeeds with healthy pivots is solved at full rank even if its singular values
 * reach below eps * sigma_max: there gelsd's rank decision is taken inside its own rounding noise and no two solvers
 * agree on it -- DESIGN.md section 3.)  strict = 1 turns the fallback into NK_ERR_NOT_SPD (also: environment variable
 * NYSKOOP_STRICT_SPD=1 before nk_create).  strict = 2 is the lstsq-shaped mode: BOTH regularised systems of every fit go
 * through the SVD and are cut at eps * sigma_max exactly as gelsd cuts -- 10-50 x slower for small fits and, on the
 * ill-conditioned candidates it was asked for, no closer to the reference than the default (profiles/r03_cloth_units.txt);
 * kept for callers who want lstsq's rank rule whatever it costs. */
int nk_set_strict_spd(nk_ctx* ctx, int strict);
/* Optional refinement of the two regularised solves of a fit (regressors.py:155,165).  The blocked Cholesky solve is
 * backward stable (every product with an inverted diagonal block takes a correction step from the data); a system whose
 * smallest / largest pivot is below `pivot_ratio` can in addition be refined `steps` times with residuals accumulated in
 * doubled precision, which returns the system's own solution to working precision whatever its condition (as long as
 * cond x eps < 1/4: a step is applied only while the corrections contract).  pivot_ratio = 0 (default; also the
 * environment variable NYSKOOP_REFINE_PIVOT before nk_create) = never.  nk_fit_stats.refined / refine_ratio_* report it. */
int nk_set_refine(nk_ctx* ctx, double pivot_ratio, int32_t steps);
/* Arithmetic of the two O(n m d) kernel blocks and the O(n m^2) Gram contractions of nk_nystrom_fit / nk_nystrom_gram
 * (regressors.py:141-142,151,153,162,164).  NK_DTYPE_F64 (default): fp64 end to end, the only mode that meets the 1e-6
 * operator bar.  NK_DTYPE_F32 (the stress configuration of BASELINE.json: n = 1e6, m = 8000, d = 1024, "fp32"): rows and
 * kernel values are rounded to fp32 and multiplied on the fp32 matrix pipe (twice the fp64 rate, half the bytes); the
 * Gram accumulators are kept in fp64 (fp32 partial sums never run over more than 32 rows) and everything m x m --
 * regularised solves, square root, operators, lift, predict, rollouts -- stays fp64.  Needs d >= 32 and shared input /
 * output landmarks; contexts of a lock-step group ignore it. */
#define NK_DTYPE_F64 0
#define NK_DTYPE_F32 1
int nk_set_compute_dtype(nk_ctx* ctx, int dtype);
/* Stream ordering for DEVICE-pointer arguments: work already queued on `producer_stream` (a hipStream_t; NULL = the
 * legacy default stream) is ordered before everything this context launches afterwards -- an event recorded on the
 * producer stream that all of the context's streams wait for; the host does not block.  Call it before handing the
 * library a device buffer that another stream is still writing (a torch tensor that is the output of a pending
 * all-reduce, for example): the context's streams are non-blocking and do not synchronise with any other stream
 * implicitly.  Results are complete when a call returns (every entry point synchronises its streams before returning
 * unless documented otherwise), so no ordering is needed in the other direction. */
int nk_wait_stream(nk_ctx* ctx, void* producer_stream);
/* ---- lock-step groups: batched execution of many SMALL fits (the (candidate, fold) units of the hyper-parameter sweep,
 * benchmark_lqr_cloth.py:39-66; multi-seed sweeps :168-203).  One small fit is a chain of a few hundred launch-bound
 * kernels that leaves the chip idle; a group runs `size` of them in lock step.  nk_group_create returns `size` member
 * contexts; each is driven by its own host thread through the ordinary entry points (nk_nystrom_fit,
 * nk_score_neg_rmse, ...).  Inside the library a member's launches are recorded, and whenever members wait for the
 * device the recorded sequences are merged -- equal launches become one launch with blockIdx.z = member -- and issued
 * on one shared stream.  Results are bit-identical to an ordinary context (same kernels, same arguments).
 * nk_group_enter / nk_group_leave bracket a unit of work of one member: members inside a unit wait for one another at
 * their synchronisation points; a member outside a unit never blocks the others (its calls still work, unbatched).
 * Enter all members that take part in a round before any of them starts (any thread may call nk_group_enter).
 * Destroy the members with nk_destroy; the group goes with its last member.
 * nk_group_stats: {flushes, merged launches, single launches, member-launches covered by merged launches}. */
int nk_group_create(int device, int size, nk_ctx** members);
/* ---- the hyper-parameter sweep as ONE call: replaces the fit/predict/score loop GridSearchCV runs over (candidate, fold)
 *   units (benchmark_lqr_cloth.py:52-65 and the classic / hjb twins; sklearn: clone -> fit(X_train, Y_train) ->
 *   'neg_root_mean_squared_error' on the held-out fold).  Unit u fits on all rows of X, Y except [test_begin, test_end)
 *   with the landmarks Y[landmark_rows[0..m)] (rows of the DATA SET, i.e. after mapping training-row indices past the
 *   fold) and scores the held-out rows.  `members`: the contexts of ONE lock-step group (nk_group_create); the units are
 *   run n_members at a time, one host thread per member inside the library, their kernel launches merged.
 *   X: n x (d+p), Y: n x d (host or device).  scores[u] = the unit's score; status[u] = NK_OK or the error code of a
 *   unit whose fit failed (its score is NaN, like GridSearchCV's error_score=nan). ------------------------------------ */
typedef struct nk_cv_unit {
  const nk_kernel_desc* kernel;
  double gamma;
  double jitter;
  int32_t m;
  int32_t reserved;
  int64_t test_begin, test_end;
  const int64_t* landmark_rows; /* m row indices into Y */
} nk_cv_unit;
int nk_cv_grid(nk_ctx* const* members, int32_t n_members, const double* X, int64_t ldx, const double* Y, int64_t ldy,
               int64_t n, int32_t d, int32_t p, const nk_cv_unit* units, int32_t n_units, double* scores, int32_t* status);
int nk_group_enter(nk_ctx* member);
int nk_group_leave(nk_ctx* member);
int nk_group_stats(nk_ctx* member, uint64_t* out4);
/* Process-wide counters of the slow paths that are otherwise silent (they cost time, never correctness):
 *   out[0] single-launch lifted recursions (nk_rollout / nk_closed_loop*, 128 < m <= 2048) that gave up waiting for a
 *          workgroup that was not resident and were repeated with one launch per step (5-10x slower);
 *   out[1] single-launch Jacobi sweeps (rank-truncating branch of the fit inside a lock-step group) that gave up the
 *          same way and finished with one launch per round;
 *   out[2] fits whose regularised system(s) took the rank-truncating branch (regressors.py:155,165: lstsq / gelsd);
 *   out[3] fits that repeated the matrix square root with the factorisation-free iteration;
 *   out[4] fits whose regularised solves were refined with doubled-precision residuals (nk_set_refine).
 * n = number of entries the caller provides (<= 5 are written). */
int nk_runtime_counters(uint64_t* out, int32_t n);
/* Releases everything the library still holds on every device -- live contexts (their streams, events and workspaces),
 * live models, the model-buffer pool and page-locked host blocks -- after waiting for pending work.  Handles that were
 * live become invalid; destroying them afterwards is a harmless no-op, so language bindings may call this from an
 * exit hook that runs BEFORE the HIP runtime's own static destructors and keep their finalisers.  Idempotent. */
int nk_shutdown(void);

/* page-locked host memory for result arrays (device->host copies into it run at the PCIe rate and skip first-touch
 * page faults); nk_host_free(NULL) is a no-op. */
void* nk_host_alloc(uint64_t bytes);
void nk_host_free(void* ptr);

/* ---- kernel matrix: replaces `kern.kernel(A, B)` (regressors.py:22,26,30 -> sklearn RBF/Matern/DotProduct
 *      __call__): out[i][j] = k(A[i,:], B[j,:]),  A: nA x d, B: nB x d, out: nA x nB. ------------------------ */
int nk_kernel_matrix(nk_ctx* ctx, const nk_kernel_desc* kd,
                     const double* A, int64_t lda, int64_t nA,
                     const double* B, int64_t ldb, int64_t nB,
                     double* out, int64_t ldo);

/* ---- fit: replaces KoopmanNystromRegressor.fit given landmarks (regressors.py:136-169).
 *   X: n x (d+p) rows [state | input] (the array the reference's fit(X, Y) receives), Y: n x d.
 *   row_ranges: optional 2*n_ranges int64 [begin,end) pairs selecting the training rows (K-fold training
 *     sets are two contiguous ranges, benchmark_lqr_cloth.py:52-65); NULL = all n rows.
 *   Zin / Zout: m x d landmark rows (nystrom_centers_input/_output transposed; regressors.py:129-134);
 *     Zin may be NULL or equal to Zout (the reference's default, :133-134).
 *   gamma, jitter: regressors.py:127,120.   The fitted operators live in *model (device resident). ------ */
int nk_nystrom_fit(nk_ctx* ctx, const nk_kernel_desc* kd,
                   const double* X, int64_t ldx, const double* Y, int64_t ldy,
                   int64_t n, int32_t d, int32_t p,
                   const int64_t* row_ranges, int32_t n_ranges,
                   const double* Zin, int64_t ldzi, const double* Zout, int64_t ldzo, int32_t m,
                   double gamma, double jitter,
                   nk_model** model, nk_fit_stats* stats);

/* ---- the same fit split at its one exchange point, for SAMPLE-SHARDED fits over several GPUs: every
 *   rank holds all landmarks and a slice of the rows, accumulates the four Gram blocks of its rows with nk_nystrom_gram
 *   (regressors.py:151,153,162,164 without the regularisers), the packed accumulators are summed over the ranks (one
 *   all-reduce of nk_gram_doubles(m,d,p) doubles: 102 MB at m=2000, d=384), and nk_nystrom_solve finishes the fit from the
 *   sum (n_total = number of rows over all ranks, regressors.py:127).  gram: host or device memory; layout
 *   [G1 (m+p)x(m+p) ; G2 m x (m+p)] row-major, then at the next even offset [G3 m x m ; G4 d x m].
 *   nk_nystrom_gram followed by nk_nystrom_solve on one rank equals nk_nystrom_fit. ----------------------------------- */
int nk_gram_doubles(int32_t m, int32_t d, int32_t p, int64_t* count);
int nk_nystrom_gram(nk_ctx* ctx, const nk_kernel_desc* kd,
                    const double* X, int64_t ldx, const double* Y, int64_t ldy,
                    int64_t n, int32_t d, int32_t p,
                    const int64_t* row_ranges, int32_t n_ranges,
                    const double* Zin, int64_t ldzi, const double* Zout, int64_t ldzo, int32_t m,
                    double* gram, nk_fit_stats* stats);
int nk_nystrom_solve(nk_ctx* ctx, const nk_kernel_desc* kd,
                     const double* Zin, int64_t ldzi, const double* Zout, int64_t ldzo, int32_t m, int32_t d, int32_t p,
                     const double* gram, int64_t n_total, double gamma, double jitter,
                     nk_model** model, nk_fit_stats* stats);

/* rebuild a device model from host copies (un-pickling a regressor, benchmark_lqr_cloth.py:266-267 /
 * closed_loop_lqr_control.m:158-161); recomputes K_mm^{-1/2} from the landmarks once. A,B,C,W may be NULL. */
int nk_model_create(nk_ctx* ctx, const nk_kernel_desc* kd, const double* Zout, int64_t ldz,
                    int32_t m, int32_t d, int32_t p, double jitter,
                    const double* A, const double* B, const double* C, const double* W,
                    nk_model** model);
int nk_model_destroy(nk_model* model);

/* which: 'A' m x m, 'B' m x p, 'C' d x m, 'W' d x (m+p), 'S' m x m (K_mm^{1/2}), 'I' m x m (K_mm^{-1/2}),
 *        'Z' m x d landmarks.   regressors.py:158-159,166,169. */
int nk_model_get(nk_ctx* ctx, const nk_model* model, char which, double* out, int64_t ldo);
int nk_model_dims(const nk_model* model, int32_t* m, int32_t* d, int32_t* p);
/* all fitted operators in one call: G = [A B] (m x (m+p), the layout of regressors.py:157-159), C and W, copied
 * concurrently on the context's streams (two DMA engines) when the outputs are host buffers.  Any of the three output
 * pointers may be NULL. */
int nk_model_get_ops(nk_ctx* ctx, const nk_model* model, double* G, int64_t ldg, double* C, int64_t ldc, double* W,
                     int64_t ldw);
/* the same without waiting: returns once the copies are queued on the context's copy stream, so that they overlap with
 * whatever the caller launches next (page-locked destination buffers, see nk_host_alloc, are needed for a true
 * overlap).  The destinations must not be read before nk_model_wait(model) has returned; nk_model_destroy waits by
 * itself.  One fetch per model is in flight at a time. */
int nk_model_get_ops_async(nk_ctx* ctx, nk_model* model, double* G, int64_t ldg, double* C, int64_t ldc, double* W,
                           int64_t ldw);
int nk_model_wait(nk_model* model);

/* ---- lift: replaces KoopmanNystromRegressor.lift (regressors.py:171-178) with K_mm^{-1/2} cached.
 *   Xq: nq x d query rows; out: nq x m (row i = phi(x_i); the reference returns the transpose, m x nq). ---- */
int nk_lift(nk_ctx* ctx, const nk_model* model, const double* Xq, int64_t ldx, int64_t nq,
            double* out, int64_t ldo);

/* ---- predict: replaces KoopmanRegressor.predict (regressors.py:48-55). Xaug: nq x (d+p); out: nq x d. -- */
int nk_predict(nk_ctx* ctx, const nk_model* model, const double* Xaug, int64_t ldx, int64_t nq,
               double* out, int64_t ldo);

/* ---- CV score: sklearn scorer 'neg_root_mean_squared_error' on a held-out block
 *      (benchmark_lqr_cloth.py:55): -(mean over columns of sqrt(mean over rows of (Y - predict(X))^2)). ---- */
int nk_score_neg_rmse(nk_ctx* ctx, const nk_model* model, const double* Xaug, int64_t ldx,
                      const double* Ytrue, int64_t ldy, int64_t nq, double* score);

/* ---- open-loop rollout: replaces the loop of validate_dyn_sys (benchmark_lqr_cloth.py:23-32) for a batch
 *   of trajectories.  x0: batch x d initial states; U: batch x T x p controls (row t of trajectory b at
 *   U + (b*T + t)*p); out_x: batch x T x d with out_x[b][0] = C lift(x0_b), out_x[b][t+1] = C(A z_t + B u_t);
 *   out_z (optional, may be NULL): batch x T x m lifted states. ------------------------------------------ */
int nk_rollout(nk_ctx* ctx, const nk_model* model, const double* x0, int64_t ldx0,
               const double* U, int32_t T, int32_t batch, double* out_x, double* out_z);

/* ---- closed loop in lifted space: replaces the loop of lqr_control (benchmark_lqr_cloth.py:79-84).
 *   K: p x m gain; phi0, phi_ref: m-vectors; out_x: steps x d visited states C phi_t; out_u: steps x p. ---- */
int nk_closed_loop(nk_ctx* ctx, const nk_model* model, const double* K, const double* phi0,
                   const double* phi_ref, int32_t steps, double* out_x, double* out_u);
/* the same for `batch` independent loops that share the gain: phi0, phi_ref: batch x m; out_x: batch x steps x d;
 * out_u: batch x steps x p (multi-seed / multi-reference sweeps of benchmark_lqr_cloth.py:212-270). */
int nk_closed_loop_batch(nk_ctx* ctx, const nk_model* model, const double* K, const double* phi0,
                         const double* phi_ref, int32_t steps, int32_t batch, double* out_x, double* out_u);
/* ---- rollout of explicit operators without a model (any estimator that exposes A, B, C: the exact-kernel comparator of
 *   benchmark_lqr_hjb.py:334-381, un-pickled gains): z0: batch x m lifted initial states; A: m x m, B: m x p, C: d x m
 *   (row-major, host or device); U, out_x, out_z as in nk_rollout. --------------------------------------------------- */
int nk_linear_rollout(nk_ctx* ctx, const double* A, const double* B, const double* C, int32_t m, int32_t d, int32_t p,
                      const double* z0, const double* U, int32_t T, int32_t batch, double* out_x, double* out_z);

/* ---- building blocks exported for parity tests and reuse (device or host pointers) --------------------- */
/* C[M x N] = alpha * op(A) op(B) + beta * C;  transA: A is stored K x M;  transB: B is stored N x K. */
/* C[M x N] (fp64) = A^T B with fp32 operands stored K x M / K x N (the fp32 engine of nk_set_compute_dtype as a building
 * block: fp32 products, fp32 partial sums over at most 32 rows, fp64 beyond).  Operands 16-byte aligned, lda / ldb
 * multiples of 4, M, N >= 4. */
int nk_gemm_f32(nk_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                double* C, int64_t ldc);
int nk_gemm(nk_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha,
            const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
/* S = P^{1/2}, Sinv = P^{-1/2} for symmetric positive definite P (m x m); regressors.py:140,163,175. */
int nk_sqrtm_spd(nk_ctx* ctx, const double* P, int64_t ldp, int32_t m, double* S, double* Sinv,
                 int32_t* iters, double* residual);
/* X = P^{-1} R for symmetric positive definite P (m x m), R: m x nrhs; regressors.py:155,165.  A numerically singular P
 * gets lstsq's minimum-norm solution (see nk_set_strict_spd). */
int nk_solve_spd(nk_ctx* ctx, const double* P, int64_t ldp, int32_t m, const double* R, int64_t ldr,
                 int32_t nrhs, double* X, int64_t ldxo);

/* ---- diagnostics ---------------------------------------------------------------------------------------- */
/* Runs `reps` fused Gram launches (the dominant kernel of nk_nystrom_fit) on a synthetic n x (2m+p) feature matrix and
 * n x d targets resident in HBM; returns the average kernel time (HIP events on the context's stream) and the
 * algorithmic flop of one launch.  For rocprofv3 --pmc runs that should not pay for a whole fit. */
int nk_bench_gram(nk_ctx* ctx, int64_t n, int32_t m, int32_t p, int32_t d, int32_t reps, double* ms_avg, double* flop);

#ifdef __cplusplus
}
#endif
#endif /* NYSKOOP_H */
