// extern "C" entry points of libnyskoop.so (include/nyskoop.h): context, staging of host/device buffers, and the
// fit / lift / predict / score / rollout pipelines assembled from the device launchers.
#include "nk_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <set>
#include <thread>

namespace nk {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

bool is_device_ptr(const void* p) {
  if (p == nullptr) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory: clear the sticky error
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// ---- arena -----------------------------------------------------------------------------------------------------
static int arena_new_chunk(nk_ctx* ctx, Arena& a, size_t bytes) {
  ArenaChunk c;
  c.cap = bytes;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&c.base), bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("HBM workspace allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return NK_ERR_OOM;
  }
  a.chunks.push_back(c);
  if (getenv("NYSKOOP_TRACE")) fprintf(stderr, "[nyskoop] arena: new chunk of %zu MiB (%zu chunks)\n", bytes >> 20, a.chunks.size());
  return NK_OK;
}

static int arena_reset_one(nk_ctx* ctx, Arena& a) {
  if (a.chunks.size() > 1) {  // coalesce: steady state is one chunk and no hipMalloc on the hot path
    NK_HIP(hipStreamSynchronize(ctx->stream_main));
    NK_HIP(hipStreamSynchronize(ctx->stream_side));
    NK_HIP(hipStreamSynchronize(ctx->stream_prep));
    size_t total = 0;
    for (auto& c : a.chunks) {
      total += c.cap;
      (void)hipFree(c.base);
    }
    a.chunks.clear();
    NK_TRY(arena_new_chunk(ctx, a, total));
  }
  for (auto& c : a.chunks) c.off = 0;
  a.cur = 0;
  return NK_OK;
}

int arena_reset(nk_ctx* ctx) {
  ctx->stream = ctx->stream_main;
  ctx->cur_arena = &ctx->arena;
  NK_TRY(arena_reset_one(ctx, ctx->arena));
  return arena_reset_one(ctx, ctx->arena_side);
}

ArenaMark arena_mark(nk_ctx* ctx) {
  Arena& a = *ctx->cur_arena;
  if (a.chunks.empty()) return ArenaMark{0, 0};
  return ArenaMark{a.cur, a.chunks[a.cur].off};
}

void arena_release(nk_ctx* ctx, ArenaMark mk) {
  Arena& a = *ctx->cur_arena;
  if (a.chunks.empty()) return;
  for (int i = mk.chunk + 1; i < (int)a.chunks.size(); ++i) a.chunks[i].off = 0;
  a.cur = mk.chunk;
  a.chunks[a.cur].off = mk.off;
}

int arena_alloc(nk_ctx* ctx, size_t bytes, void** out) {
  Arena& a = *ctx->cur_arena;
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  for (;;) {
    if (!a.chunks.empty()) {
      ArenaChunk& c = a.chunks[a.cur];
      if (c.off + bytes <= c.cap) {
        *out = c.base + c.off;
        c.off += bytes;
        return NK_OK;
      }
      if (a.cur + 1 < (int)a.chunks.size()) {
        ++a.cur;
        a.chunks[a.cur].off = 0;
        continue;
      }
    }
    size_t want = bytes;
    const size_t min_chunk = (size_t)256 << 20;
    if (want < min_chunk) want = min_chunk;
    if (!a.chunks.empty() && want < a.chunks.back().cap) want = a.chunks.back().cap;
    NK_TRY(arena_new_chunk(ctx, a, want));
    a.cur = (int)a.chunks.size() - 1;
  }
}

// ---- staging ---------------------------------------------------------------------------------------------------
// Device view of a caller matrix (rows x cols, leading dimension ld). Host data is copied into the arena.
struct MatIn {
  const double* ptr = nullptr;
  int64_t ld = 0;
  bool staged = false;
};
static int stage_in(nk_ctx* ctx, const double* p, int64_t ld, int64_t rows, int64_t cols, MatIn* out) {
  if (rows <= 0 || cols <= 0) {
    out->ptr = p;
    out->ld = ld;
    return NK_OK;
  }
  if (is_device_ptr(p)) {
    out->ptr = p;
    out->ld = ld;
    out->staged = false;
    return NK_OK;
  }
  const int64_t ldd = cols + (cols & 1);
  double* d = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)rows * ldd, &d));
  NK_HIP(hipMemcpy2DAsync(d, (size_t)ldd * 8, p, (size_t)ld * 8, (size_t)cols * 8, (size_t)rows, hipMemcpyHostToDevice,
                          ctx->stream));
  out->ptr = d;
  out->ld = ldd;
  out->staged = true;
  return NK_OK;
}
struct MatOut {
  double* dev = nullptr;
  int64_t ld = 0;
  double* host = nullptr;
  int64_t host_ld = 0;
  int64_t rows = 0, cols = 0;
};
static int stage_out(nk_ctx* ctx, double* p, int64_t ld, int64_t rows, int64_t cols, MatOut* out) {
  out->rows = rows;
  out->cols = cols;
  if (is_device_ptr(p)) {
    out->dev = p;
    out->ld = ld;
    out->host = nullptr;
    return NK_OK;
  }
  const int64_t ldd = cols + (cols & 1);
  NK_TRY(arena_alloc_t(ctx, (size_t)(rows > 0 ? rows : 1) * ldd, &out->dev));
  out->ld = ldd;
  out->host = p;
  out->host_ld = ld;
  return NK_OK;
}
static int finish_out(nk_ctx* ctx, const MatOut& o) {
  if (o.host && o.rows > 0 && o.cols > 0)
    NK_HIP(hipMemcpy2DAsync(o.host, (size_t)o.host_ld * 8, o.dev, (size_t)o.ld * 8, (size_t)o.cols * 8, (size_t)o.rows,
                            hipMemcpyDeviceToHost, ctx->stream));
  return NK_OK;
}

static int check_ctx(nk_ctx* ctx) {
  if (!ctx) {
    set_error("null context");
    return NK_ERR_BAD_ARG;
  }
  tl_ctx = ctx;  // the calling thread's current context (a context is used by one thread at a time)
  NK_HIP(hipSetDevice(ctx->device));
  return arena_reset(ctx);
}

// 1/lengthscale per dimension on the device (ones for the linear kernel)
static int make_winv(nk_ctx* ctx, const nk_kernel_desc* kd, int d, double* dst_dev) {
  NK_REQUIRE(kd != nullptr, "null kernel descriptor");
  NK_REQUIRE(kd->type >= NK_KERNEL_RBF && kd->type <= NK_KERNEL_LINEAR, "unknown kernel type %d", kd->type);
  NK_REQUIRE(kd->d == d, "kernel descriptor is for %d dimensions, data has %d", kd->d, d);
  std::vector<double> w((size_t)d, 1.0);
  if (kd->type != NK_KERNEL_LINEAR) {
    NK_REQUIRE(kd->lengthscale != nullptr, "kernel lengthscale pointer is null");
    // sklearn _check_length_scale: an anisotropic kernel must match the data dimension
    NK_REQUIRE(kd->n_lengthscale == 1 || kd->n_lengthscale == d,
               "Anisotropic kernel must have the same number of dimensions as data (%d!=%d)", kd->n_lengthscale, d);
    for (int k = 0; k < d; ++k) {
      const double l = kd->lengthscale[kd->n_lengthscale == 1 ? 0 : k];
      NK_REQUIRE(l > 0.0 && std::isfinite(l), "lengthscale[%d] = %g is not positive", k, l);
      w[k] = 1.0 / l;
    }
  }
  NK_HIP(hipMemcpyAsync(dst_dev, w.data(), sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));  // w is a stack vector
  return NK_OK;
}

// Model buffers are recycled through a small per-process free list: hipFree synchronises the whole device and a
// sweep creates and drops one model per fit.
struct ModelBuf {
  int device;
  size_t bytes;
  double* ptr;
};
// Everything the library owns is registered here, so that nk_shutdown can release it in a defined order BEFORE the HIP
// runtime's static destructors run, and so that destroying a handle twice (or after nk_shutdown) is a no-op.
static std::mutex g_pool_mu;
static std::vector<ModelBuf> g_pool;
static std::mutex g_reg_mu;
static std::set<nk_ctx*> g_ctxs;
static std::set<nk_model*> g_models;
static std::set<void*> g_host_blocks;

static double* pool_take(int device, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = 0; i < g_pool.size(); ++i)
    if (g_pool[i].device == device && g_pool[i].bytes == bytes) {
      double* p = g_pool[i].ptr;
      g_pool.erase(g_pool.begin() + i);
      return p;
    }
  return nullptr;
}
static void pool_give(int device, size_t bytes, double* ptr) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < 512) {  // a lock-step group keeps one model per member in flight
      g_pool.push_back(ModelBuf{device, bytes, ptr});
      return;
    }
  }
  (void)hipSetDevice(device);
  (void)hipFree(ptr);
}
static void pool_drain() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto& b : g_pool) {
    (void)hipSetDevice(b.device);
    (void)hipFree(b.ptr);
  }
  g_pool.clear();
}

static int model_alloc(nk_ctx* ctx, int m, int d, int p, nk_model** out) {
  nk_model* mdl = new nk_model();
  mdl->device = ctx->device;
  mdl->m = m; mdl->d = d; mdl->p = p;
  const size_t mp = (size_t)m + p;
  const size_t total = (size_t)m * mp /*G=[A B]*/ + (size_t)d * m /*C*/ + (size_t)d * mp /*W*/ + 2 * (size_t)m * m +
                       (size_t)m * d /*Z*/ + (size_t)d /*winv*/ + 64;
  mdl->bytes = total * sizeof(double);
  mdl->buf = pool_take(ctx->device, mdl->bytes);
  if (!mdl->buf) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&mdl->buf), mdl->bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      delete mdl;
      set_error("model allocation of %zu bytes failed: %s", total * sizeof(double), hipGetErrorString(e));
      return NK_ERR_OOM;
    }
  }
  double* q = mdl->buf;
  auto take = [&](size_t cnt) { double* r = q; q += (cnt + 1) & ~(size_t)1; return r; };
  mdl->A = take((size_t)m * mp);
  mdl->B = mdl->A + m;  // view into G = [A | B], leading dimension m + p
  mdl->C = take((size_t)d * m);
  mdl->W = take((size_t)d * mp);
  mdl->S = take((size_t)m * m);
  mdl->Sinv = take((size_t)m * m);
  mdl->Z = take((size_t)m * d);
  mdl->winv = take((size_t)d);
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_models.insert(mdl);
  }
  *out = mdl;
  return NK_OK;
}

// phi (nq x m, ld ldo) = k(Xq, Z) * Sinv, processed in row chunks
static int lift_device(nk_ctx* ctx, const nk_model* mdl, const double* Xq, int64_t ldx, int64_t nq, double* out,
                       int64_t ldo) {
  const int m = mdl->m;
  const int64_t chunk = 32768;
  const ArenaMark mk = arena_mark(ctx);
  double* Kq = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)(nq < chunk ? nq : chunk) * m, &Kq));
  for (int64_t r0 = 0; r0 < nq; r0 += chunk) {
    const int64_t len = nq - r0 < chunk ? nq - r0 : chunk;
    NK_TRY(launch_kmat(ctx, mdl->ktype, Xq + r0 * ldx, ldx, len, mdl->Z, mdl->d, m, mdl->d, mdl->winv, mdl->sigma0, Kq,
                       m));
    NK_TRY(launch_gemm(ctx, false, false, len, m, m, 1.0, Kq, m, mdl->Sinv, m, 0.0, out + r0 * ldo, ldo));
  }
  arena_release(ctx, mk);
  return NK_OK;
}

// out (nq x d) = [phi(X) | U] W^T
static int predict_device(nk_ctx* ctx, const nk_model* mdl, const double* Xaug, int64_t ldx, int64_t nq, double* out,
                          int64_t ldo) {
  const int m = mdl->m, d = mdl->d, p = mdl->p;
  const int64_t chunk = 32768;
  const ArenaMark mk = arena_mark(ctx);
  double* phi = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)(nq < chunk ? nq : chunk) * m, &phi));
  for (int64_t r0 = 0; r0 < nq; r0 += chunk) {
    const int64_t len = nq - r0 < chunk ? nq - r0 : chunk;
    NK_TRY(lift_device(ctx, mdl, Xaug + r0 * ldx, ldx, len, phi, m));
    NK_TRY(launch_gemm(ctx, false, true, len, d, m, 1.0, phi, m, mdl->W, m + p, 0.0, out + r0 * ldo, ldo));
    if (p > 0)
      NK_TRY(launch_gemm(ctx, false, true, len, d, p, 1.0, Xaug + r0 * ldx + d, ldx, mdl->W + m, m + p, 1.0,
                         out + r0 * ldo, ldo));
  }
  arena_release(ctx, mk);
  return NK_OK;
}

// NYSKOOP_TRACE=1: host-side wall clock of the fit's phases on stderr (diagnostics)
struct HostTrace {
  bool on;
  std::chrono::steady_clock::time_point t0, last;
  HostTrace() : on(getenv("NYSKOOP_TRACE") != nullptr) { t0 = last = std::chrono::steady_clock::now(); }
  void mark(const char* what) {
    if (!on) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[nk trace] %-28s +%8.3f ms (total %8.3f)\n", what,
            std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
};

static float ev_ms(nk_ctx* ctx, int a, int b) {
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]);
  return ms;
}

}  // namespace nk

using namespace nk;

extern "C" {

int nk_version(void) { return NK_ABI_VERSION; }

const char* nk_last_error(void) { return g_err.c_str(); }

int nk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int nk_create(int device, nk_ctx** out) {
  NK_REQUIRE(out != nullptr, "nk_create: null output pointer");
  *out = nullptr;
  int n = nk_device_count();
  if (n <= 0) {
    set_error("no HIP device visible: libnyskoop has no CPU fallback");
    return NK_ERR_NO_DEVICE;
  }
  NK_REQUIRE(device >= 0 && device < n, "nk_create: device %d out of range (0..%d)", device, n - 1);
  NK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  NK_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; libnyskoop is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return NK_ERR_NO_DEVICE;
  }
  nk_ctx* ctx = new nk_ctx();
  ctx->device = device;
  ctx->num_cu = prop.multiProcessorCount;
  // Three priority levels (the dispatcher serves hardware queues in strict priority order, so a queue with pending
  // workgroups starves every lower one):
  //   prep stream: highest -- a short latency-bound chain queued beside the big kernel-block / Gram launches of the main
  //                stream; its tiny kernels must get the first CU slots that free up;
  //   main stream: middle  -- the big launches and the factorisation chains;
  //   side stream: lowest  -- the GEMM-bound square-root iteration that fills the chip beside the factorisation chain.
  int prio_lo = 0, prio_hi = 0;
  NK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  const int prio_mid = (prio_lo - prio_hi >= 2) ? (prio_lo + prio_hi) / 2 : prio_hi;
  NK_HIP(hipStreamCreateWithPriority(&ctx->stream_main, hipStreamNonBlocking, prio_mid));
  NK_HIP(hipStreamCreateWithPriority(&ctx->stream_side, hipStreamNonBlocking, prio_lo));
  NK_HIP(hipStreamCreateWithPriority(&ctx->stream_prep, hipStreamNonBlocking, prio_hi));
  NK_HIP(hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking));
  NK_HIP(hipStreamCreateWithPriority(&ctx->stream_la[0], hipStreamNonBlocking, prio_hi));
  NK_HIP(hipStreamCreateWithPriority(&ctx->stream_la[1], hipStreamNonBlocking, prio_mid));
  for (int q = 0; q < 2; ++q)
    for (int i = 0; i < 4; ++i) NK_HIP(hipEventCreateWithFlags(&ctx->ev_la[q][i], hipEventDisableTiming));
  ctx->stream = ctx->stream_main;
  ctx->cur_arena = &ctx->arena;
  NK_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  NK_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  NK_HIP(hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming));
  NK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_info), 256));
  ctx->d_piv = reinterpret_cast<unsigned long long*>(ctx->d_info + 16);  // 8 x 8 bytes behind the 4 flag slots
  NK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_scalars), 64 * sizeof(double)));
  // (mapped + coherent, explicitly: in lock-step groups the device stores to these mirrors directly, nk_group.hip)
  NK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_scalars), 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  NK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_info), 256, hipHostMallocMapped | hipHostMallocCoherent));
  ctx->h_piv = reinterpret_cast<unsigned long long*>(ctx->h_info + 16);
  for (int i = 0; i < 16; ++i) NK_HIP(hipEventCreate(&ctx->ev[i]));
  NK_HIP(hipEventCreateWithFlags(&ctx->ev_ext, hipEventDisableTiming));
  for (int i = 0; i < 8; ++i) NK_HIP(hipEventCreateWithFlags(&ctx->ev_up[i], hipEventDisableTiming));
  const char* km = getenv("NYSKOOP_KMAT");
  ctx->kmat_mode = (km && strcmp(km, "direct") == 0) ? 1 : 0;
  const char* st = getenv("NYSKOOP_STRICT_SPD");
  ctx->strict_spd = (st && atoi(st) != 0) ? 1 : 0;
  if (const char* rp = getenv("NYSKOOP_REFINE_PIVOT")) ctx->refine_pivot = atof(rp);
  if (const char* rs = getenv("NYSKOOP_REFINE_STEPS")) ctx->refine_steps = std::max(0, atoi(rs));
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_ctxs.insert(ctx);
  }
  *out = ctx;
  return NK_OK;
}

static void destroy_ctx_unregistered(nk_ctx* ctx) {
  (void)hipSetDevice(ctx->device);
  if (tl_ctx == ctx) tl_ctx = nullptr;
  group_detach(ctx);  // flushes what the member recorded; the group goes with its last member
  (void)hipStreamSynchronize(ctx->stream_main);
  (void)hipStreamSynchronize(ctx->stream_side);
  (void)hipStreamSynchronize(ctx->stream_prep);
  (void)hipStreamSynchronize(ctx->stream_copy);
  for (int q = 0; q < 2; ++q) {
    if (ctx->stream_la[q]) { (void)hipStreamSynchronize(ctx->stream_la[q]); (void)hipStreamDestroy(ctx->stream_la[q]); }
    for (int i = 0; i < 4; ++i) if (ctx->ev_la[q][i]) (void)hipEventDestroy(ctx->ev_la[q][i]);
  }
  for (auto& c : ctx->arena.chunks) (void)hipFree(c.base);
  for (auto& c : ctx->arena_side.chunks) (void)hipFree(c.base);
  (void)hipEventDestroy(ctx->ev_fork);
  (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_chain) (void)hipEventDestroy(ctx->ev_chain);
  if (ctx->ev_ext) (void)hipEventDestroy(ctx->ev_ext);
  for (int i = 0; i < 8; ++i) if (ctx->ev_up[i]) (void)hipEventDestroy(ctx->ev_up[i]);
  (void)hipStreamDestroy(ctx->stream_side);
  (void)hipStreamDestroy(ctx->stream_prep);
  (void)hipStreamDestroy(ctx->stream_copy);
  (void)hipFree(ctx->d_info);
  (void)hipFree(ctx->d_scalars);
  if (ctx->d_zeros) (void)hipFree(ctx->d_zeros);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  (void)hipHostFree(ctx->h_scalars);
  (void)hipHostFree(ctx->h_info);
  for (int i = 0; i < 16; ++i) (void)hipEventDestroy(ctx->ev[i]);
  (void)hipStreamDestroy(ctx->stream_main);
  delete ctx;
}

int nk_destroy(nk_ctx* ctx) {
  if (!ctx) return NK_OK;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_ctxs.erase(ctx) == 0) return NK_OK;  // already destroyed (nk_shutdown, or a second call)
  }
  destroy_ctx_unregistered(ctx);
  return NK_OK;
}

static int model_wait_unchecked(nk_model* mdl);
static void destroy_model_unregistered(nk_model* model, bool to_pool) {
  (void)model_wait_unchecked(model);  // a pending asynchronous fetch still reads the buffers
  if (to_pool) {
    pool_give(model->device, model->bytes, model->buf);
  } else {
    (void)hipSetDevice(model->device);
    (void)hipFree(model->buf);
  }
  delete model;
}

int nk_shutdown(void) {
  tl_ctx = nullptr;
  std::set<nk_ctx*> ctxs;
  std::set<nk_model*> models;
  std::set<void*> blocks;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    ctxs.swap(g_ctxs);
    models.swap(g_models);
    blocks.swap(g_host_blocks);
  }
  for (nk_ctx* c : ctxs) {  // drain everything first: models and pinned blocks may still be targets of queued copies
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream_main);
    (void)hipStreamSynchronize(c->stream_side);
    (void)hipStreamSynchronize(c->stream_prep);
    (void)hipStreamSynchronize(c->stream_copy);
  }
  for (nk_model* m : models) destroy_model_unregistered(m, false);
  for (nk_ctx* c : ctxs) destroy_ctx_unregistered(c);
  pool_drain();
  for (void* b : blocks) (void)hipHostFree(b);
  return NK_OK;
}

int nk_set_compute_dtype(nk_ctx* ctx, int dtype) {
  NK_REQUIRE(ctx != nullptr && (dtype == NK_DTYPE_F64 || dtype == NK_DTYPE_F32), "nk_set_compute_dtype: bad argument");
  ctx->compute_f32 = dtype == NK_DTYPE_F32 ? 1 : 0;
  return NK_OK;
}

int nk_set_strict_spd(nk_ctx* ctx, int strict) {
  NK_REQUIRE(ctx != nullptr && strict >= 0 && strict <= 2, "nk_set_strict_spd: bad argument");
  ctx->strict_spd = strict;
  return NK_OK;
}

int nk_set_refine(nk_ctx* ctx, double pivot_ratio, int32_t steps) {
  NK_REQUIRE(ctx != nullptr && pivot_ratio >= 0.0 && pivot_ratio <= 1.0 && steps >= 0 && steps <= 8, "nk_set_refine: bad argument");
  ctx->refine_pivot = pivot_ratio;
  ctx->refine_steps = steps;
  return NK_OK;
}

int nk_wait_stream(nk_ctx* ctx, void* producer_stream) {
  NK_REQUIRE(ctx != nullptr, "nk_wait_stream: null context");
  NK_HIP(hipSetDevice(ctx->device));
  if (ctx->group) {  // the group's shared stream is fed at flush time: wait for the producer here, once
    hipError_t e = real_stream_sync(reinterpret_cast<hipStream_t>(producer_stream));
    if (e != hipSuccess) { set_error("synchronising the producer stream failed: %s", hipGetErrorString(e)); return NK_ERR_HIP; }
    return NK_OK;
  }
  NK_HIP(hipEventRecord(ctx->ev_ext, reinterpret_cast<hipStream_t>(producer_stream)));
  NK_HIP(hipStreamWaitEvent(ctx->stream_main, ctx->ev_ext, 0));
  NK_HIP(hipStreamWaitEvent(ctx->stream_side, ctx->ev_ext, 0));
  NK_HIP(hipStreamWaitEvent(ctx->stream_prep, ctx->ev_ext, 0));
  NK_HIP(hipStreamWaitEvent(ctx->stream_copy, ctx->ev_ext, 0));
  return NK_OK;
}

int nk_group_create(int device, int size, nk_ctx** out) {
  NK_REQUIRE(out != nullptr && size >= 1 && size <= 256, "nk_group_create: bad argument");
  for (int i = 0; i < size; ++i) out[i] = nullptr;
  for (int i = 0; i < size; ++i) {
    int rc = nk_create(device, &out[i]);
    if (rc != NK_OK) {
      for (int j = 0; j < i; ++j) { nk_destroy(out[j]); out[j] = nullptr; }
      return rc;
    }
  }
  nk_group* g = group_new(device, size);
  if (!g) {
    for (int i = 0; i < size; ++i) { nk_destroy(out[i]); out[i] = nullptr; }
    set_error("nk_group_create: could not create the shared stream / argument table");
    return NK_ERR_HIP;
  }
  for (int i = 0; i < size; ++i) group_attach(g, i, out[i]);
  return NK_OK;
}

int nk_group_enter(nk_ctx* ctx) {
  NK_REQUIRE(ctx != nullptr, "nk_group_enter: null context");
  return group_enter(ctx);
}

int nk_group_leave(nk_ctx* ctx) {
  NK_REQUIRE(ctx != nullptr, "nk_group_leave: null context");
  tl_ctx = ctx;
  const int rc = group_leave(ctx);
  tl_ctx = nullptr;  // the thread is outside the unit: nothing it does next may be recorded for (or wait on) this member
  return rc;
}

int nk_runtime_counters(uint64_t* out, int32_t n) {
  NK_REQUIRE(out != nullptr && n >= 0, "nk_runtime_counters: bad argument");
  for (int i = 0; i < n && i < CNT_N; ++i) out[i] = nk::read_counter(i);
  return NK_OK;
}

int nk_group_stats(nk_ctx* ctx, uint64_t* out4) {
  NK_REQUIRE(ctx != nullptr && out4 != nullptr, "nk_group_stats: null argument");
  group_stats(ctx, out4);
  return NK_OK;
}

int nk_cv_grid(nk_ctx* const* members, int32_t n_members, const double* X, int64_t ldx, const double* Y, int64_t ldy,
               int64_t n, int32_t d, int32_t p, const nk_cv_unit* units, int32_t n_units, double* scores, int32_t* status) {
  NK_REQUIRE(members && n_members >= 1 && X && Y && units && scores, "nk_cv_grid: null argument");
  NK_REQUIRE(n > 0 && d > 0 && p >= 0 && n_units >= 0 && ldx >= d + p && ldy >= d, "nk_cv_grid: bad sizes");
  for (int k = 0; k < n_members; ++k) NK_REQUIRE(members[k] != nullptr, "nk_cv_grid: null member context");
  for (int u = 0; u < n_units; ++u) {
    const nk_cv_unit& cu = units[u];
    NK_REQUIRE(cu.kernel && cu.landmark_rows && cu.m > 0, "nk_cv_grid: unit %d: null kernel / landmarks", u);
    NK_REQUIRE(0 <= cu.test_begin && cu.test_begin < cu.test_end && cu.test_end <= n, "nk_cv_grid: unit %d: bad test fold", u);
    for (int j = 0; j < cu.m; ++j)
      NK_REQUIRE(cu.landmark_rows[j] >= 0 && cu.landmark_rows[j] < n, "nk_cv_grid: unit %d: landmark row out of range", u);
  }
  if (n_units == 0) return NK_OK;
  nk_ctx* lead = members[0];
  NK_HIP(hipSetDevice(lead->device));
  // the data set lives in HBM once for all units
  const double *Xd = X, *Yd = Y;
  int64_t ldxd = ldx, ldyd = ldy;
  double *Xown = nullptr, *Yown = nullptr;
  struct Free { double*& a; double*& b; ~Free() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } freer{Xown, Yown};
  if (!is_device_ptr(X)) {
    ldxd = d + p + ((d + p) & 1);
    NK_HIP(hipMalloc(reinterpret_cast<void**>(&Xown), (size_t)n * ldxd * 8));
    NK_HIP(hipMemcpy2D(Xown, (size_t)ldxd * 8, X, (size_t)ldx * 8, (size_t)(d + p) * 8, (size_t)n, hipMemcpyHostToDevice));
    Xd = Xown;
  }
  if (!is_device_ptr(Y)) {
    ldyd = d + (d & 1);
    NK_HIP(hipMalloc(reinterpret_cast<void**>(&Yown), (size_t)n * ldyd * 8));
    NK_HIP(hipMemcpy2D(Yown, (size_t)ldyd * 8, Y, (size_t)ldy * 8, (size_t)d * 8, (size_t)n, hipMemcpyHostToDevice));
    Yd = Yown;
  }
  // landmark rows are gathered on the host from a host copy of Y (one download if Y came as a device pointer)
  std::vector<double> Yhost;
  const double* Yh = Y;
  int64_t ldyh = ldy;
  if (is_device_ptr(Y)) {
    Yhost.resize((size_t)n * d);
    NK_HIP(hipMemcpy2D(Yhost.data(), (size_t)d * 8, Y, (size_t)ldy * 8, (size_t)d * 8, (size_t)n, hipMemcpyDeviceToHost));
    Yh = Yhost.data();
    ldyh = d;
  }
  const int B = n_members;
  // one host thread per member; rounds of B units; everybody is inside its unit before anybody starts (round barrier)
  struct Round {
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    void wait(int parties) {
      std::unique_lock<std::mutex> lk(mu);
      const uint64_t g = gen;
      if (++arrived == parties) { arrived = 0; ++gen; lk.unlock(); cv.notify_all(); return; }
      cv.wait(lk, [&] { return gen != g; });
    }
  };
  // Two phases.  A unit whose regularised system is numerically rank deficient takes gelsd's branch: a Jacobi SVD of ~1e4
  // launches (0.2 s at m = 500) during which the other members of its round have nothing to merge with and wait at the
  // next barrier -- a 405-unit cloth grid with 30 such units spent 2 of its 2.4 s that way.  So the first phase runs every
  // unit in strict mode (the factorisation reports the condition and the unit stops there), and the units that reported
  // it are run again TOGETHER in a second phase with the fallback enabled: their Jacobi sweeps merge into shared launches.
  // Same kernels on the same data in both orders: the scores do not depend on the schedule.
  std::vector<int> saved_strict((size_t)B);
  for (int k = 0; k < B; ++k) saved_strict[(size_t)k] = members[k]->strict_spd;
  std::mutex deferred_mu;
  std::vector<int> deferred;
  auto run_units = [&](const std::vector<int>& list, bool defer_rank_deficient) {
    Round round;
    const int count = (int)list.size();
    const int n_rounds = (count + B - 1) / B;
    auto worker = [&](int k) {
      nk_ctx* ctx = members[k];
      std::vector<double> Z;
      for (int r = 0; r < n_rounds; ++r) {
        const int slot = r * B + k;
        const bool mine = slot < count;
        if (mine) (void)group_enter(ctx);
        round.wait(B);
        if (mine) {
          const int u = list[(size_t)slot];
          const nk_cv_unit& cu = units[u];
          Z.resize((size_t)cu.m * d);
          for (int j = 0; j < cu.m; ++j) memcpy(&Z[(size_t)j * d], Yh + cu.landmark_rows[j] * ldyh, (size_t)d * 8);
          const int64_t rr[4] = {0, cu.test_begin, cu.test_end, n};
          nk_model* mdl = nullptr;
          int rc = nk_nystrom_fit(ctx, cu.kernel, Xd, ldxd, Yd, ldyd, n, d, p, rr, 2, nullptr, 0, Z.data(), d, cu.m, cu.gamma,
                                  cu.jitter, &mdl, nullptr);
          double sc = std::nan("");
          if (rc == NK_OK)
            rc = nk_score_neg_rmse(ctx, mdl, Xd + cu.test_begin * ldxd, ldxd, Yd + cu.test_begin * ldyd, ldyd,
                                   cu.test_end - cu.test_begin, &sc);
          if (mdl) nk_model_destroy(mdl);
          tl_ctx = ctx;
          const int rc_leave = group_leave(ctx);  // flushes what the unit recorded after its last synchronisation
          if (rc == NK_OK) rc = rc_leave;
          if (rc == NK_ERR_NOT_SPD && defer_rank_deficient) {
            std::lock_guard<std::mutex> lk(deferred_mu);
            deferred.push_back(u);
          } else {
            scores[u] = rc == NK_OK ? sc : std::nan("");
            if (status) status[u] = rc;
          }
        }
        round.wait(B);
      }
      tl_ctx = nullptr;
    };
    std::vector<std::thread> threads;
    threads.reserve((size_t)B);
    for (int k = 0; k < B; ++k) threads.emplace_back(worker, k);
    for (auto& t : threads) t.join();
  };
  std::vector<int> all((size_t)n_units);
  for (int u = 0; u < n_units; ++u) all[(size_t)u] = u;
  bool all_lenient = true;
  for (int k = 0; k < B; ++k) all_lenient = all_lenient && saved_strict[(size_t)k] == 0;
  if (!all_lenient) {  // the caller wants the error (strict contexts): one phase, nothing to defer
    run_units(all, false);
    return NK_OK;
  }
  const bool cv_trace = getenv("NYSKOOP_CV_TRACE") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  for (int k = 0; k < B; ++k) members[k]->strict_spd = 1;
  run_units(all, true);
  for (int k = 0; k < B; ++k) members[k]->strict_spd = saved_strict[(size_t)k];
  const auto t_mid = std::chrono::steady_clock::now();
  if (!deferred.empty()) {
    std::sort(deferred.begin(), deferred.end());
    for (int k = 0; k < B; ++k) members[k]->strict_spd = 0;
    run_units(deferred, false);
    for (int k = 0; k < B; ++k) members[k]->strict_spd = saved_strict[(size_t)k];
  }
  if (cv_trace)
    fprintf(stderr, "[nyskoop] cv_grid: %d units in %.3f s, %zu rank-deficient units again in %.3f s (%d members)\n", n_units,
            std::chrono::duration<double>(t_mid - t_start).count(), deferred.size(),
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_mid).count(), B);
  return NK_OK;
}

int nk_synchronize(nk_ctx* ctx) {
  NK_REQUIRE(ctx != nullptr, "null context");
  tl_ctx = ctx;
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

void* nk_stream(nk_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

void* nk_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    set_error("hipHostMalloc of %llu bytes failed", (unsigned long long)bytes);
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_host_blocks.insert(p);
  return p;
}

void nk_host_free(void* ptr) {
  if (!ptr) return;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_host_blocks.erase(ptr) == 0) return;  // not ours, or already released by nk_shutdown
  }
  (void)hipHostFree(ptr);
}

int nk_set_kmat_mode(nk_ctx* ctx, int mode) {
  NK_REQUIRE(ctx != nullptr && (mode == 0 || mode == 1), "nk_set_kmat_mode: bad argument");
  ctx->kmat_mode = mode;
  return NK_OK;
}

int nk_kernel_matrix(nk_ctx* ctx, const nk_kernel_desc* kd, const double* A, int64_t lda, int64_t nA, const double* B,
                     int64_t ldb, int64_t nB, double* out, int64_t ldo) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(kd && A && B && out, "nk_kernel_matrix: null argument");
  NK_REQUIRE(nA >= 0 && nB >= 0 && kd->d > 0, "nk_kernel_matrix: negative size");
  NK_REQUIRE(lda >= kd->d && ldb >= kd->d && ldo >= nB, "nk_kernel_matrix: leading dimension too small");
  if (nA == 0 || nB == 0) return NK_OK;
  const int d = kd->d;
  double* winv = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)d, &winv));
  NK_TRY(make_winv(ctx, kd, d, winv));
  MatIn a, b;
  NK_TRY(stage_in(ctx, A, lda, nA, d, &a));
  NK_TRY(stage_in(ctx, B, ldb, nB, d, &b));
  MatOut o;
  NK_TRY(stage_out(ctx, out, ldo, nA, nB, &o));
  NK_TRY(launch_kmat(ctx, kd->type, a.ptr, a.ld, nA, b.ptr, b.ld, nB, d, winv, kd->sigma0, o.dev, o.ld));
  NK_TRY(finish_out(ctx, o));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

// Packed Gram accumulator of a fit (nk_nystrom_gram / nk_nystrom_solve): [G1 (m+p)x(m+p) ; G2 m x (m+p)] row-major with
// leading dimension m+p, then (at an even offset) [G3 m x m ; G4 d x m] with leading dimension m.
static inline size_t gram_block1(int m, int p) { return (((size_t)(2 * m + p) * (m + p)) + 1) & ~(size_t)1; }
static inline size_t gram_doubles(int m, int d, int p) { return gram_block1(m, p) + (size_t)(m + d) * m; }

enum { FIT_FULL = 0, FIT_GRAM = 1, FIT_SOLVE = 2 };

// One implementation for the three entry points: FIT_FULL (nk_nystrom_fit), FIT_GRAM (accumulate the Gram blocks of the
// given rows into gram_io and stop), FIT_SOLVE (start from the accumulated Gram blocks in gram_io, n = total row count).
static int fit_impl(nk_ctx* ctx, const nk_kernel_desc* kd, const double* X, int64_t ldx, const double* Y, int64_t ldy,
                    int64_t n, int32_t d, int32_t p, const int64_t* row_ranges, int32_t n_ranges, const double* Zin,
                    int64_t ldzi, const double* Zout, int64_t ldzo, int32_t m, double gamma, double jitter,
                    nk_model** model, nk_fit_stats* stats, int mode, double* gram_io) {
  HostTrace tr;
  NK_TRY(check_ctx(ctx));
  tr.mark("check_ctx/arena_reset");
  NK_REQUIRE(kd && Zout, "nk_nystrom_fit: null argument");
  NK_REQUIRE(mode == FIT_SOLVE || (X && Y), "nk_nystrom_fit: null data pointer");
  NK_REQUIRE(mode == FIT_GRAM || model != nullptr, "nk_nystrom_fit: null model pointer");
  NK_REQUIRE(mode == FIT_FULL || gram_io != nullptr, "nk_nystrom_gram/solve: null accumulator");
  NK_REQUIRE(n > 0 && d > 0 && p >= 0 && m > 0, "nk_nystrom_fit: sizes must be positive (n=%lld d=%d p=%d m=%d)",
             (long long)n, d, p, m);
  NK_REQUIRE(mode == FIT_SOLVE || (ldx >= d + p && ldy >= d), "nk_nystrom_fit: leading dimension too small");
  NK_REQUIRE(ldzo >= d, "nk_nystrom_fit: leading dimension too small");
  NK_REQUIRE(std::isfinite(gamma) && std::isfinite(jitter), "nk_nystrom_fit: gamma/jitter not finite");
  if (model) *model = nullptr;
  std::vector<int64_t> rng;
  if (mode != FIT_SOLVE && row_ranges && n_ranges > 0) {
    for (int i = 0; i < n_ranges; ++i) {
      const int64_t b = row_ranges[2 * i], e = row_ranges[2 * i + 1];
      NK_REQUIRE(0 <= b && b <= e && e <= n, "nk_nystrom_fit: row range %d = [%lld,%lld) outside [0,%lld)", i,
                 (long long)b, (long long)e, (long long)n);
      if (e > b) { rng.push_back(b); rng.push_back(e); }
    }
  } else {
    rng.push_back(0); rng.push_back(n);
  }
  int64_t n_eff = 0;
  for (size_t i = 0; i < rng.size(); i += 2) n_eff += rng[i + 1] - rng[i];
  NK_REQUIRE(n_eff > 0, "nk_nystrom_fit: no training rows selected");
  const bool same_centers = (Zin == nullptr) || (Zin == Zout && ldzi == ldzo);
  const int mp = m + p;
  const double gamma_n = gamma * (double)n_eff;  // regressors.py:127

  nk_model* mdl = nullptr;
  NK_TRY(model_alloc(ctx, m, d, p, &mdl));
  // on an early (error) return: drain both streams before the model buffers go back to the pool
  struct Guard {
    nk_ctx* c;
    nk_model* m;
    ~Guard() {
      if (m) {
        (void)hipStreamSynchronize(c->stream_main);
        (void)hipStreamSynchronize(c->stream_side);
        (void)hipStreamSynchronize(c->stream_prep);
        (void)hipStreamSynchronize(c->stream_la[0]);
        (void)hipStreamSynchronize(c->stream_la[1]);
        nk_model_destroy(m);
      }
    }
  } guard{ctx, mdl};
  mdl->ktype = kd->type; mdl->sigma0 = kd->sigma0; mdl->jitter = jitter;

  tr.mark("validate + model_alloc");
  hipEvent_t* ev = ctx->ev;
  NK_HIP(hipEventRecord(ev[0], ctx->stream));
  NK_TRY(make_winv(ctx, kd, d, mdl->winv));
  MatIn x, y, zi, zo;
  // Large HOST arrays (how the reference's fit(X, Y) is called: 620 MB at the headline shape, ~11 ms of PCIe): the rows
  // are uploaded in `host_passes` blocks on the side stream, block k + 1 while the kernel blocks and the Gram launch of
  // block k run (the contraction is accumulated over passes anyway, below).  Only the first block's upload is exposed.
  int host_passes = 1;
  if (mode != FIT_SOLVE) {
    const char* e = getenv("NYSKOOP_HOST_PASSES");
    const int want = e ? atoi(e) : 6;  // measured at the headline shape: 3 -> 47.5, 4 -> 47.2, 6 -> 46.2, 8 -> 46.6 ms per fit
    if (want > 1 && rng.size() == 2 && !ctx_recording(ctx) && (double)n_eff * (2.0 * d + p) * 8.0 >= 64e6 &&
        !is_device_ptr(X) && !is_device_ptr(Y))
      host_passes = want > 8 ? 8 : want;
  }
  if (mode != FIT_SOLVE && host_passes > 1) {
    x.ld = (d + p + 1) & ~(int64_t)1;
    y.ld = (d + 1) & ~(int64_t)1;
    double *xd = nullptr, *yd = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)n * x.ld, &xd));
    NK_TRY(arena_alloc_t(ctx, (size_t)n * y.ld, &yd));
    x.ptr = xd; y.ptr = yd; x.staged = y.staged = true;
  } else if (mode != FIT_SOLVE) {
    NK_TRY(stage_in(ctx, X, ldx, n, d + p, &x));
    NK_TRY(stage_in(ctx, Y, ldy, n, d, &y));
  }
  if (mode != FIT_SOLVE && rng.size() > 2 && (double)n_eff * (2.0 * d + p) * 8.0 <= 256e6) {
    // several row ranges (a K-fold training set is two): gather the rows into contiguous scratch once, so that everything
    // downstream sees ONE piece of n_eff rows whatever the split point -- the kernel blocks and the fused Gram launch then
    // have the same shape for every fold (which is also what lets the units of a sweep share launches, nk_lockstep.h)
    const int64_t ldxg = (d + p + 1) & ~(int64_t)1, ldyg = (d + 1) & ~(int64_t)1;
    double *xg = nullptr, *yg = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)n_eff * ldxg, &xg));
    NK_TRY(arena_alloc_t(ctx, (size_t)n_eff * ldyg, &yg));
    int64_t o = 0;
    for (size_t i = 0; i < rng.size(); i += 2) {
      const int64_t b = rng[i], len = rng[i + 1] - rng[i];
      NK_TRY(launch_copy2d(ctx, x.ptr + b * x.ld, x.ld, xg + o * ldxg, ldxg, len, d + p));
      NK_TRY(launch_copy2d(ctx, y.ptr + b * y.ld, y.ld, yg + o * ldyg, ldyg, len, d));
      o += len;
    }
    x.ptr = xg; x.ld = ldxg; y.ptr = yg; y.ld = ldyg;
    rng.assign({(int64_t)0, n_eff});
  }
  NK_TRY(stage_in(ctx, Zout, ldzo, m, d, &zo));
  if (same_centers) zi = zo; else NK_TRY(stage_in(ctx, Zin, ldzi, m, d, &zi));
  NK_TRY(launch_copy2d(ctx, zo.ptr, zo.ld, mdl->Z, d, m, d));
  NK_HIP(hipEventRecord(ev[1], ctx->stream));
  tr.mark("staging issued");

  // ---- landmark kernels (regressors.py:139,143,144) -----------------------------------------------------------------
  double *Kmm = nullptr, *Kj = nullptr, *Kj_in = nullptr, *Kxo = nullptr;
  bool landmarks_aside = false;
  if (mode != FIT_GRAM) {
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &Kmm));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &Kj));
  if (same_centers) {
    Kj_in = Kj;
    Kxo = Kmm;
  } else {
    NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &Kj_in));
    NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &Kxo));
  }
  // Nothing needs the landmark matrices before the fused Gram launch has ended (the square root's preparation chain, the
  // regularisers of the two systems), and at the headline shape K_mm is 160 us on 16 workgroups: for large fits they are
  // built on the preparation stream, beside the row preparation and the kernel blocks instead of in front of them
  // (ev[8]: ready; the main stream waits for it where it assembles the systems).
  {
    const char* la = getenv("NYSKOOP_LANDMARKS_ASIDE");  // 0: on the main stream, in front of the kernel blocks (read per fit: A/B runs)
    landmarks_aside = mode == FIT_FULL && n_eff >= 20000 && m >= 1024 && !ctx_recording(ctx) && !(la && la[0] == '0');
  }
  {
    hipStream_t s0 = ctx->stream;
    if (landmarks_aside) {
      NK_HIP(hipEventRecord(ev[12], ctx->stream));  // the staged landmarks / lengthscales are ready
      ctx->stream = ctx->stream_prep;
      NK_HIP(hipStreamWaitEvent(ctx->stream, ev[12], 0));
    }
    int rc_l = launch_kmat(ctx, kd->type, zo.ptr, zo.ld, m, zo.ptr, zo.ld, m, d, mdl->winv, kd->sigma0, Kmm, m);
    if (rc_l == NK_OK) rc_l = launch_copy2d(ctx, Kmm, m, Kj, m, m, m);
    if (rc_l == NK_OK) rc_l = launch_add_diag(ctx, Kj, m, m, jitter);
    if (rc_l == NK_OK && !same_centers) {
      rc_l = launch_kmat(ctx, kd->type, zi.ptr, zi.ld, m, zi.ptr, zi.ld, m, d, mdl->winv, kd->sigma0, Kj_in, m);
      if (rc_l == NK_OK) rc_l = launch_add_diag(ctx, Kj_in, m, m, jitter);
      if (rc_l == NK_OK) rc_l = launch_kmat(ctx, kd->type, zi.ptr, zi.ld, m, zo.ptr, zo.ld, m, d, mdl->winv, kd->sigma0, Kxo, m);
    }
    const hipError_t he = rc_l == NK_OK ? hipEventRecord(ev[8], ctx->stream) : hipSuccess;  // the landmark matrices are ready
    ctx->stream = s0;
    NK_TRY(rc_l);
    NK_HIP(he);
  }
  }
  // Gram accumulators (regressors.py:151,153,162,164), one packed block (see gram_doubles):
  //   G1 = Phi_in^T Phi_in (symmetric), G2 = Phi_out^T Phi_in (= cross), G3 = Phi_out^T Phi_out (symmetric),
  //   G4 = Y^T Phi_out (= left_rec).  G2 sits directly below G1 and G4 below G3: the right-hand sides of the two
  //   regularised systems ride along the blocked factorisations as extra rows (cholesky_aug_pair_async).
  double *G1 = nullptr, *G2 = nullptr, *G3 = nullptr, *G4 = nullptr;
  const int64_t ldd = d + (d & 1);
  NK_TRY(arena_alloc_t(ctx, gram_doubles(m, d, p), &G1));
  G2 = G1 + (size_t)mp * mp;
  G3 = G1 + gram_block1(m, p);
  G4 = G3 + (size_t)m * m;
  float ms_gram_kernel = 0.f;
  int gram_launches = 0;
  bool gram_deferred = false;
  const bool timed = stats != nullptr;
  // ---- the matrix square root S = (K_mm + jitter I)^{1/2}, S^-1 (regressors.py:140,163) runs beside the main stream's work in
  //      two parts, queued by the two lambdas below: the latency-bound preparation (preparation stream) and the GEMM-bound
  //      iteration with the products that depend on S only (side stream), both behind the fused Gram launch and beside the
  //      factorisation chain of the regularised systems.  (Round 3 also measured the whole square root queued BEFORE the Gram
  //      launch, beside the kernel blocks, with the Gram launch waiting for it: 44.5 against 42.9 ms per fit -- the chain's
  //      ~100 small kernels each wait for a workgroup slot of the long-running kernel blocks, the square root takes 13 ms
  //      there instead of 9, and what the tail gains (the factorisation chain alone: 4.2 ms) the wait gives back.)
  SqrtPlan splan;
  int it = 0;
  double resid = 0.0;
  double *Sinvt = nullptr, *T1t = nullptr, *X1 = nullptr;
  auto alloc_sqrt_bufs = [&]() -> int {
    if (Sinvt == nullptr) NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &Sinvt));
    if (T1t == nullptr) NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &T1t));
    if (X1 == nullptr) NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &X1));
    return NK_OK;
  };
  // products that depend on the square root only (current stream)
  auto sqrt_products = [&]() -> int {
    NK_TRY(launch_transpose(ctx, mdl->Sinv, m, Sinvt, m, m, m));
    if (same_centers) {
      // K_xo = K_mm = S^2 - jitter I, hence K_xo S^-1 = S - jitter S^-1: no product (and a smaller rounding error than
      // the product, whose terms are ||K|| ||S^-1|| large)
      NK_TRY(launch_copy2d(ctx, mdl->S, m, T1t, m, m, m));
      NK_TRY(launch_axpby2d(ctx, -jitter, mdl->Sinv, m, 1.0, T1t, m, m, m));
    } else {
      NK_TRY(launch_transpose(ctx, Kxo, m, X1, m, m, m));
      NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, X1, m, mdl->Sinv, m, 0.0, T1t, m));
    }
    return NK_OK;
  };
  // Cholesky factor of K_mm + jitter and its inverse, the latency-bound half of the square root: small kernels on the
  // preparation stream (queued behind the kernel-block / Gram launches so that the main stream is never kept waiting for
  // the host)
  auto queue_prep = [&]() -> int {
    SideScope prep(ctx, ctx->stream_prep);
    NK_HIP(hipStreamWaitEvent(ctx->stream, ev[8], 0));
    // kernel matrices are positive semi-definite: the jitter bounds the smallest eigenvalue of K_mm + jitter I from
    // below, which lets the iteration be queued before this factorisation has run (SqrtPlan::lambda_min_hint)
    splan.lambda_min_hint = jitter > 0.0 ? jitter : 0.0;
    // This chain of ~100 small high-priority kernels must not run beside the fused Gram launch: that launch fills the
    // chip in exact rounds of 3-ms workgroups, and a Gram workgroup whose slot a chain kernel holds at a round boundary
    // finds the next free slot a whole round later -- measured: the launch takes 26.5-27.0 ms inside the fit against 25.0
    // alone.  So the chain waits for the Gram launch and runs after it -- ahead of the square-root iteration, which has
    // that much slack against the factorisation chain of the regularised systems.  NYSKOOP_PREP_PAUSE = fraction of the
    // block steps to run BEFORE the pause, beside the kernel blocks (1 = never pause).  Measured on one box, ms per fit:
    // 1 -> 44.1, 0.75 -> 44.1, 0.5 -> 43.2, 0.25 -> 43.2, 0 (default) -> 42.9 (kernel blocks 6.9 -> 6.3, Gram 26.7 -> 25.5).
    if (mode == FIT_FULL && n_eff >= 20000 && m >= 1024) {
      static const double frac = getenv("NYSKOOP_PREP_PAUSE") ? atof(getenv("NYSKOOP_PREP_PAUSE")) : 0.0;
      const int nb = (m + CHOL_NB - 1) / CHOL_NB;
      if (frac < 1.0) {
        splan.pause_event = ctx->ev_fork;  // recorded behind the last Gram launch
        splan.pause_step = std::max(0, std::min(nb - 1, (int)(frac * nb)));
      }
    }
    NK_TRY(sqrtm_prepare(ctx, Kj, m, m, &splan));
    NK_HIP(hipEventRecord(ev[9], ctx->stream));
    return NK_OK;
  };
  // the GEMM-bound iteration, then S^-T and K_xo S^-1, on the side stream
  auto queue_side = [&]() -> int {
    NK_TRY(alloc_sqrt_bufs());
    SideScope side(ctx);
    NK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));  // starts when the Gram launch is done
    NK_HIP(hipStreamWaitEvent(ctx->stream, ev[9], 0));
    NK_HIP(hipEventRecord(ev[6], ctx->stream));
    NK_TRY(sqrtm_finish(ctx, &splan, mdl->S, mdl->Sinv));
    NK_HIP(hipEventRecord(ev[7], ctx->stream));
    // still on the side stream (the factorisation chain is usually not finished yet): S^-T and K_xo S^-1
    NK_TRY(sqrt_products());
    NK_HIP(hipEventRecord(ctx->ev_join, ctx->stream));
    return NK_OK;
  };
  if (mode == FIT_SOLVE) {
    // the accumulated Gram blocks come from the caller (host or device memory)
    NK_HIP(hipMemcpyAsync(G1, gram_io, gram_doubles(m, d, p) * sizeof(double),
                          is_device_ptr(gram_io) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    NK_HIP(hipEventRecord(ev[2], ctx->stream));
  } else {
  // ---- feature matrix F = [K_nm_in | U | (pad) | K_nm_out], sample-major (regressors.py:141-142,147), built and
  //      contracted in PASSES of at most `pass_rows` rows so that the workspace stays bounded for very large n (the
  //      Gram accumulators are updated with beta = 1 from the second pass on); C4 (3.2 GB) is a single pass.
  // fp32 engine (nk_set_compute_dtype): the feature matrix, the prepared rows and Y as the operand of G4 are fp32; the
  // Gram accumulators come back fp64 (nk_gemm_tn_f32.hip)
  const bool f32 = ctx->compute_f32 != 0 && ctx->kmat_mode == 0 && d >= 32 && !ctx_recording(ctx) && m >= 4 && d >= 4;
  const int64_t off_out = f32 ? ((mp + 3) & ~3) : ((mp + 1) & ~1);
  const int64_t ldf = f32 ? ((off_out + m + 3) & ~(int64_t)3) : ((off_out + m + 1) & ~(int64_t)1);
  const double felem = f32 ? 4.0 : 8.0;
  int64_t pass_rows;
  {
    const char* b = getenv("NYSKOOP_F_BUDGET_GB");
    const double budget = (b ? atof(b) : 48.0) * 1073741824.0;
    pass_rows = (int64_t)(budget / ((double)ldf * felem));
    if (pass_rows < 1024) pass_rows = 1024;
    pass_rows &= ~(int64_t)1023;  // whole k-steps per pass (the assembly k loops take K ranges of full steps only)
    if (host_passes > 1) {  // pipelined upload: at least `host_passes` passes (more if the workspace budget says so)
      const int64_t per = ((n_eff + host_passes - 1) / host_passes + 1023) & ~(int64_t)1023;
      if (per < pass_rows) pass_rows = per;
    }
  }
  struct Piece { int64_t b, len; };
  std::vector<std::vector<Piece>> passes(1);
  {
    int64_t used = 0;
    for (size_t i = 0; i < rng.size(); i += 2) {
      int64_t b = rng[i];
      const int64_t e = rng[i + 1];
      while (b < e) {
        if (used == pass_rows) { passes.emplace_back(); used = 0; }
        const int64_t len = std::min(e - b, pass_rows - used);
        passes.back().push_back(Piece{b, len});
        used += len;
        b += len;
      }
    }
  }
  const int64_t f_rows = passes.size() > 1 ? pass_rows : n_eff;
  double* F = nullptr;
  NK_TRY(arena_alloc_t(ctx, f32 ? ((size_t)f_rows * ldf + 1) / 2 + 64 : (size_t)f_rows * ldf + 64, &F));
  const bool gram_form = ctx->kmat_mode == 0 && d >= 32;
  // Gram-form kernel blocks (MFMA engine): rows centred on the landmark mean, scaled by 1/l, transposed
  int64_t maxlen = 0;
  for (auto& ps : passes) for (auto& pc : ps) maxlen = std::max(maxlen, pc.len);
  const int64_t ldt = (maxlen + 1) & ~(int64_t)1, ldzt = (m + 1) & ~1;
  double *center = nullptr, *Rt = nullptr, *sqr = nullptr, *Zto = nullptr, *sqzo = nullptr, *Zti = nullptr, *sqzi = nullptr;
  double *Rt2 = nullptr, *sqr2 = nullptr;
  const bool overlap_prep = gram_form && passes.size() == 1 && passes[0].size() == 1;
  if (gram_form) {
    NK_TRY(arena_alloc_t(ctx, (size_t)d, &center));
    if (!f32) {
      NK_TRY(arena_alloc_t(ctx, (size_t)d * ldt, &Rt));
      NK_TRY(arena_alloc_t(ctx, (size_t)maxlen, &sqr));
    }
    if (overlap_prep && !f32) {
      NK_TRY(arena_alloc_t(ctx, (size_t)d * ldt, &Rt2));
      NK_TRY(arena_alloc_t(ctx, (size_t)maxlen, &sqr2));
    }
    NK_TRY(arena_alloc_t(ctx, (size_t)d * ldzt, &Zto));
    NK_TRY(arena_alloc_t(ctx, (size_t)m, &sqzo));
    if (kd->type == NK_KERNEL_LINEAR) NK_TRY(launch_fill(ctx, center, d, 1, d, 0.0));  // x.y is not shift invariant
    else NK_TRY(launch_colmean(ctx, zo.ptr, zo.ld, m, d, center));
    NK_TRY(prep_rows(ctx, zo.ptr, zo.ld, m, d, mdl->winv, center, Zto, ldzt, sqzo));
    if (same_centers) {
      Zti = Zto; sqzi = sqzo;
    } else {
      NK_TRY(arena_alloc_t(ctx, (size_t)d * ldzt, &Zti));
      NK_TRY(arena_alloc_t(ctx, (size_t)m, &sqzi));
      NK_TRY(prep_rows(ctx, zi.ptr, zi.ld, m, d, mdl->winv, center, Zti, ldzt, sqzi));
    }
  }
  float *F32 = reinterpret_cast<float*>(F), *Rt32 = nullptr, *sq32 = nullptr, *Zt32 = nullptr, *sqz32 = nullptr, *Y32 = nullptr;
  const int64_t ldt32 = (maxlen + 3) & ~(int64_t)3, ldzt32 = (m + 3) & ~3, ldy32 = (d + 3) & ~3;
  if (f32) {
    NK_REQUIRE(same_centers, "fp32 engine: separate input landmarks are not supported");
    double* tmp = nullptr;
    NK_TRY(arena_alloc_t(ctx, ((size_t)d * ldt32 + 1) / 2 + 2, &tmp)); Rt32 = reinterpret_cast<float*>(tmp);
    NK_TRY(arena_alloc_t(ctx, ((size_t)maxlen + 1) / 2 + 2, &tmp)); sq32 = reinterpret_cast<float*>(tmp);
    NK_TRY(arena_alloc_t(ctx, ((size_t)d * ldzt32 + 1) / 2 + 2, &tmp)); Zt32 = reinterpret_cast<float*>(tmp);
    NK_TRY(arena_alloc_t(ctx, ((size_t)m + 1) / 2 + 2, &tmp)); sqz32 = reinterpret_cast<float*>(tmp);
    NK_TRY(arena_alloc_t(ctx, ((size_t)f_rows * ldy32 + 1) / 2 + 2, &tmp)); Y32 = reinterpret_cast<float*>(tmp);
    NK_TRY(prep_rows_f32(ctx, zo.ptr, zo.ld, m, d, mdl->winv, center, Zt32, ldzt32, sqz32));
  }
  const bool multi_pass = passes.size() > 1;
  const bool pipelined = host_passes > 1;
  // upload of the rows of pass ip from the caller's host arrays, on the side stream (the call blocks the HOST while the
  // runtime moves pageable memory through its bounce buffers; the device works on the previous pass meanwhile)
  auto upload_pass = [&](size_t ip) -> int {
    for (const Piece& pc : passes[ip]) {
      NK_HIP(hipMemcpy2DAsync(const_cast<double*>(x.ptr) + pc.b * x.ld, (size_t)x.ld * 8, X + pc.b * ldx, (size_t)ldx * 8,
                              (size_t)(d + p) * 8, (size_t)pc.len, hipMemcpyHostToDevice, ctx->stream_side));
      NK_HIP(hipMemcpy2DAsync(const_cast<double*>(y.ptr) + pc.b * y.ld, (size_t)y.ld * 8, Y + pc.b * ldy, (size_t)ldy * 8,
                              (size_t)d * 8, (size_t)pc.len, hipMemcpyHostToDevice, ctx->stream_side));
    }
    NK_HIP(hipEventRecord(ctx->ev_up[ip & 7], ctx->stream_side));
    return NK_OK;
  };
  if (pipelined) {
    NK_HIP(hipEventRecord(ev[12], ctx->stream));  // the side stream starts after whatever the main stream has queued
    NK_HIP(hipStreamWaitEvent(ctx->stream_side, ev[12], 0));
    NK_TRY(upload_pass(0));
  }
  for (size_t ip = 0; ip < passes.size(); ++ip) {
    const std::vector<Piece>& ps = passes[ip];
    const double beta = ip == 0 ? 0.0 : 1.0;
    if (pipelined) NK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_up[ip & 7], 0));
    if (f32) {
      // -- fp32 engine: kernel blocks, then ONE fused Gram launch with fp64 results
      int64_t o32 = 0;
      for (const Piece& pc : ps) {
        const double* xs = x.ptr + pc.b * x.ld;
        const double* ys = y.ptr + pc.b * y.ld;
        NK_TRY(prep_rows_f32(ctx, xs, x.ld, pc.len, d, mdl->winv, center, Rt32, ldt32, sq32));
        NK_TRY(launch_kmat_gram_f32(ctx, kd->type, Rt32, ldt32, sq32, pc.len, Zt32, ldzt32, sqz32, m, d, kd->sigma0,
                                    F32 + o32 * ldf, ldf));
        NK_TRY(prep_rows_f32(ctx, ys, y.ld, pc.len, d, mdl->winv, center, Rt32, ldt32, sq32));
        NK_TRY(launch_kmat_gram_f32(ctx, kd->type, Rt32, ldt32, sq32, pc.len, Zt32, ldzt32, sqz32, m, d, kd->sigma0,
                                    F32 + o32 * ldf + off_out, ldf));
        if (p > 0) NK_TRY(launch_cvt_f64_f32(ctx, xs + d, x.ld, F32 + o32 * ldf + m, ldf, pc.len, p));
        NK_TRY(launch_cvt_f64_f32(ctx, ys, y.ld, Y32 + o32 * ldy32, ldy32, pc.len, d));
        o32 += pc.len;
      }
      if (ip == 0) {
        NK_HIP(hipEventRecord(ev[2], ctx->stream));
        tr.mark("kmat issued");
      }
      TnProblemF pf[4];
      pf[0].A = F32; pf[0].B = F32; pf[0].lda = pf[0].ldb = ldf; pf[0].M = pf[0].N = mp; pf[0].C = G1; pf[0].ldc = mp;
      pf[0].tri = TRI_UPPER_MIRROR;
      pf[1].A = F32 + off_out; pf[1].B = F32; pf[1].lda = pf[1].ldb = ldf; pf[1].M = m; pf[1].N = mp; pf[1].C = G2; pf[1].ldc = mp;
      pf[2].A = F32 + off_out; pf[2].B = F32 + off_out; pf[2].lda = pf[2].ldb = ldf; pf[2].M = pf[2].N = m; pf[2].C = G3;
      pf[2].ldc = m; pf[2].tri = TRI_UPPER_MIRROR;
      pf[3].A = Y32; pf[3].lda = ldy32; pf[3].M = d; pf[3].N = m; pf[3].C = G4; pf[3].ldc = m; pf[3].B = F32 + off_out;
      pf[3].ldb = ldf;
      for (int q = 0; q < 4; ++q) pf[q].beta = beta;
      float ms1 = 0.f;
      NK_TRY(launch_gemm_tn_f32_multi(ctx, pf, 4, o32, 0, timed ? &ms1 : nullptr, multi_pass));
      if (multi_pass) ms_gram_kernel += ms1; else gram_deferred = timed;
      gram_launches += 1;
      if (pipelined && ip + 1 < passes.size()) NK_TRY(upload_pass(ip + 1));
      continue;
    }
    // -- kernel blocks of this pass
    int64_t o = 0;
    for (const Piece& pc : ps) {
      const double* xs = x.ptr + pc.b * x.ld;
      const double* ys = y.ptr + pc.b * y.ld;
      if (gram_form && overlap_prep) {
        // single piece: the (HBM-bound) preparation of the Y rows runs on the side stream beside the (MFMA-bound) kernel
        // block of the X rows, into its own scratch
        NK_HIP(hipEventRecord(ev[12], ctx->stream));  // landmarks, centre and the staged data are ready
        {
          SideScope side(ctx);
          NK_HIP(hipStreamWaitEvent(ctx->stream, ev[12], 0));
          NK_TRY(prep_rows(ctx, ys, y.ld, pc.len, d, mdl->winv, center, Rt2, ldt, sqr2));
          NK_HIP(hipEventRecord(ev[13], ctx->stream));
        }
        NK_TRY(prep_rows(ctx, xs, x.ld, pc.len, d, mdl->winv, center, Rt, ldt, sqr));
        NK_TRY(launch_kmat_gram(ctx, kd->type, Rt, ldt, sqr, pc.len, Zti, ldzt, sqzi, m, d, kd->sigma0, F + o * ldf, ldf));
        NK_HIP(hipStreamWaitEvent(ctx->stream, ev[13], 0));
        NK_TRY(launch_kmat_gram(ctx, kd->type, Rt2, ldt, sqr2, pc.len, Zto, ldzt, sqzo, m, d, kd->sigma0,
                                F + o * ldf + off_out, ldf));
      } else if (gram_form) {
        NK_TRY(prep_rows(ctx, xs, x.ld, pc.len, d, mdl->winv, center, Rt, ldt, sqr));
        NK_TRY(launch_kmat_gram(ctx, kd->type, Rt, ldt, sqr, pc.len, Zti, ldzt, sqzi, m, d, kd->sigma0, F + o * ldf, ldf));
        NK_TRY(prep_rows(ctx, ys, y.ld, pc.len, d, mdl->winv, center, Rt, ldt, sqr));
        NK_TRY(launch_kmat_gram(ctx, kd->type, Rt, ldt, sqr, pc.len, Zto, ldzt, sqzo, m, d, kd->sigma0,
                                F + o * ldf + off_out, ldf));
      } else {
        NK_TRY(launch_kmat(ctx, kd->type, xs, x.ld, pc.len, zi.ptr, zi.ld, m, d, mdl->winv, kd->sigma0, F + o * ldf, ldf));
        NK_TRY(launch_kmat(ctx, kd->type, ys, y.ld, pc.len, zo.ptr, zo.ld, m, d, mdl->winv, kd->sigma0,
                           F + o * ldf + off_out, ldf));
      }
      if (p > 0) NK_TRY(launch_copy2d(ctx, xs + d, x.ld, F + o * ldf + m, ldf, pc.len, p));
      o += pc.len;
    }
    const int64_t rows = o;
    if (ip == 0) {
      NK_HIP(hipEventRecord(ev[2], ctx->stream));
      tr.mark("kmat issued");
    }
    // -- contraction of this pass: ONE fused launch when the operands meet the LDS-DMA alignment contract
    TnProblem pr[4];
    pr[0].A = F; pr[0].B = F; pr[0].lda = pr[0].ldb = ldf; pr[0].M = pr[0].N = mp; pr[0].C = G1; pr[0].ldc = mp;
    pr[0].tri = TRI_UPPER_MIRROR;
    pr[1].A = F + off_out; pr[1].B = F; pr[1].lda = pr[1].ldb = ldf; pr[1].M = m; pr[1].N = mp; pr[1].C = G2;
    pr[1].ldc = mp;
    pr[2].A = F + off_out; pr[2].B = F + off_out; pr[2].lda = pr[2].ldb = ldf; pr[2].M = pr[2].N = m; pr[2].C = G3;
    pr[2].ldc = m; pr[2].tri = TRI_UPPER_MIRROR;
    pr[3].A = y.ptr + ps[0].b * y.ld; pr[3].lda = y.ld; pr[3].M = d; pr[3].N = m; pr[3].C = G4; pr[3].ldc = m;
    pr[3].B = F + off_out; pr[3].ldb = ldf;
    for (int q = 0; q < 4; ++q) pr[q].beta = beta;
    const bool single = ps.size() == 1;
    const bool fast = tn_fast_ok(pr[0]) && tn_fast_ok(pr[1]) && tn_fast_ok(pr[2]);
    const bool fast_y = fast && tn_fast_ok(pr[3]);
    bool y_done = false;
    if (fast) {
      const int np = (single && fast_y) ? 4 : 3;
      float ms1 = 0.f;
      // (pipelined uploads: no per-launch timing, it would make the host wait for the launch before the next upload)
      NK_TRY(launch_gemm_tn_multi(ctx, pr, np, rows, 0, (timed && !pipelined) ? &ms1 : nullptr, multi_pass));
      if (multi_pass) ms_gram_kernel += ms1; else gram_deferred = timed;
      gram_launches += 1;
      y_done = np == 4;
    } else {  // unaligned operands (odd m+p): generic engine
      GemmOpts sym;
      sym.tri = TRI_UPPER_MIRROR;
      float t3[3] = {0.f, 0.f, 0.f};
      NK_TRY(launch_gemm(ctx, true, false, mp, mp, rows, 1.0, F, ldf, F, ldf, beta, G1, mp, sym, timed ? &t3[0] : nullptr));
      NK_TRY(launch_gemm(ctx, true, false, m, mp, rows, 1.0, F + off_out, ldf, F, ldf, beta, G2, mp, GemmOpts(),
                         timed ? &t3[1] : nullptr));
      NK_TRY(launch_gemm(ctx, true, false, m, m, rows, 1.0, F + off_out, ldf, F + off_out, ldf, beta, G3, m, sym,
                         timed ? &t3[2] : nullptr));
      ms_gram_kernel += t3[0] + t3[1] + t3[2];
      gram_launches += 3;
    }
    if (!y_done) {
      int64_t oo = 0;
      bool first = true;
      for (const Piece& pc : ps) {
        NK_TRY(launch_gemm(ctx, true, false, d, m, pc.len, 1.0, y.ptr + pc.b * y.ld, y.ld, F + oo * ldf + off_out, ldf,
                           (first && ip == 0) ? 0.0 : 1.0, G4, m));
        oo += pc.len;
        first = false;
      }
    }
    if (pipelined && ip + 1 < passes.size()) NK_TRY(upload_pass(ip + 1));
  }
  }  // mode != FIT_SOLVE
  NK_HIP(hipEventRecord(ev[3], ctx->stream));
  if (mode == FIT_GRAM) {
    // hand the accumulated blocks to the caller and stop (the model only carried the kernel parameters)
    NK_HIP(hipMemcpyAsync(gram_io, G1, gram_doubles(m, d, p) * sizeof(double),
                          is_device_ptr(gram_io) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    if (stats) {
      memset(stats, 0, sizeof(*stats));
      stats->ms_total = ev_ms(ctx, 0, 3);
      stats->ms_upload = (x.staged || y.staged) ? ev_ms(ctx, 0, 1) : 0.0;
      stats->ms_kmat = ev_ms(ctx, 1, 2);
      stats->ms_gram = ev_ms(ctx, 2, 3);
      if (gram_deferred) ms_gram_kernel = ev_ms(ctx, 14, 15);
      stats->ms_gram_kernel_avg = gram_launches ? ms_gram_kernel / gram_launches : 0.0;
      stats->gram_kernel_launches = gram_launches;
    }
    if (ctx->arena.chunks.size() > 1 || ctx->arena_side.chunks.size() > 1) NK_TRY(arena_reset(ctx));
    return NK_OK;  // the guard returns the model buffers to the pool
  }
  NK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));  // the square-root iteration starts when the Gram launch is done
  tr.mark("gram issued");

  NK_TRY(queue_prep());

  // ---- the two regularised systems (regressors.py:151,162) are assembled, factorised AND solved on the main stream
  //      without waiting for the square root: with inner and inner_rec symmetric,
  //        [A B] = S^-1 (cross inner^-1) blkdiag(K_xo S^-1, I)            cross = G2     (regressors.py:152-156)
  //        C     = (left_rec inner_rec^-1) S                               left_rec = G4  (regressors.py:163-166)
  //      so the right-hand sides are cross^T (m columns) and left_rec^T (only d columns instead of the reference's m).
  if (landmarks_aside) NK_HIP(hipStreamWaitEvent(ctx->stream, ev[8], 0));  // (long done: built beside the kernel blocks)
  NK_TRY(launch_axpby2d(ctx, gamma_n, Kj_in, m, 1.0, G1, mp, m, m));               // inner = G1 + gamma_n*blkdiag(K, I)
  if (p > 0) NK_TRY(launch_add_diag(ctx, G1 + (int64_t)m * mp + m, mp, p, gamma_n));
  NK_TRY(launch_axpby2d(ctx, gamma_n, Kj, m, 1.0, G3, m, m, m));                   // inner_rec = gamma_n K + G3
  // The factorisations below work in place.  A copy of the assembled systems and their right-hand sides (one device
  // copy of the packed block: 0.1 % of a fit) is what the rank-truncating fallback starts from if a pivot turns out
  // non-positive (regressors.py:155,165: lstsq / gelsd semantics).
  double* Gsave = nullptr;
  NK_TRY(arena_alloc_t(ctx, gram_doubles(m, d, p), &Gsave));
  NK_HIP(hipMemcpyAsync(Gsave, G1, gram_doubles(m, d, p) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  double *Linv = nullptr, *Linv2 = nullptr, *V1 = nullptr, *Wc = nullptr, *Ct = nullptr;
  const int nblk = (mp + CHOL_NB - 1) / CHOL_NB;
  NK_TRY(arena_alloc_t(ctx, (size_t)nblk * CHOL_WS, &Linv));
  NK_TRY(arena_alloc_t(ctx, (size_t)nblk * CHOL_WS, &Linv2));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &V1));
  NK_TRY(alloc_sqrt_bufs());
  NK_TRY(arena_alloc_t(ctx, (size_t)m * ldd, &Wc));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * ldd, &Ct));
  CholSys sys[2];
  double* pivlog = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)mp + m + 4, &pivlog));
  sys[0].P = G1; sys[0].ldp = mp; sys[0].m = mp; sys[0].Linv = Linv; sys[0].extra = m;   // [inner; cross]
  sys[1].P = G3; sys[1].ldp = m; sys[1].m = m; sys[1].Linv = Linv2; sys[1].extra = d;    // [inner_rec; left_rec]
  sys[0].pivlog = pivlog; sys[1].pivlog = pivlog + mp + (mp & 1);
  // both systems advance in lock step (paired launches); the per-block kernels are latency bound and leave the chip
  // mostly idle ...
  NK_TRY(cholesky_aug_pair_async(ctx, sys, 2));  // G2 <- cross inner^-1 (m x mp) ; G4 <- left_rec inner_rec^-1 (d x m)
  // (tried: holding the GEMM-bound iteration back until this latency-bound chain is done -- 43.3 against 42.7 ms per fit;
  // with look-ahead in both chains 43.7 / 47.5: the overlap of the two, slow as each becomes, is still the best schedule)
  tr.mark("cholesky + solves issued");

  NK_TRY(queue_side());
  NK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  NK_HIP(hipEventRecord(ev[4], ctx->stream));
  tr.mark("side stream joined (queued)");
  // (the verdict of the factorisations is read at the end of the call: a blocking check here would leave the GPU idle
  // while the host wakes up and queues the products; on a failed factorisation they compute on garbage, harmlessly)

  // ---- operator products; every product is P^T Q with P stored contraction-major (fast TN engine) -----------------------
  //   [A B] = S^-1 (cross inner^-1) blkdiag(K_xo S^-1, I)   with  cross inner^-1 = [V1^T | V2^T] in G2
  //   (T1t holds K_xo S^-1, computed on the side stream)
  auto operator_products = [&]() -> int {
    NK_TRY(launch_transpose(ctx, G2, mp, V1, m, m, m));                                         // V1 (m x m)
    NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, V1, m, T1t, m, 0.0, X1, m));             // X1 = V1^T (K_xo S^-1)
    NK_TRY(launch_gemm(ctx, true, false, m, m, m, 1.0, Sinvt, m, X1, m, 0.0, mdl->A, mp));      // A = S^-1 X1
    if (p > 0) NK_TRY(launch_gemm(ctx, true, false, m, p, m, 1.0, Sinvt, m, G2 + m, mp, 0.0, mdl->B, mp));  // B = S^-1 V2^T
    //   C = (left_rec inner_rec^-1) S : C^T = S^T Wc with Wc = G4^T
    NK_TRY(launch_transpose(ctx, G4, m, Wc, ldd, d, m));
    NK_TRY(launch_gemm(ctx, true, false, m, d, m, 1.0, mdl->S, m, Wc, ldd, 0.0, Ct, ldd));      // C^T = S^T Wc
    NK_TRY(launch_transpose(ctx, Ct, ldd, mdl->C, m, m, d));
    NK_TRY(launch_gemm(ctx, true, false, d, mp, m, 1.0, Ct, ldd, mdl->A, mp, 0.0, mdl->W, mp));  // W = C G (:167)
    return NK_OK;
  };
  NK_TRY(operator_products());
  NK_HIP(hipEventRecord(ev[5], ctx->stream));
  tr.mark("solve issued");
  int chol_failed[2] = {0, 0};
  double piv_ratio[2] = {1.0, 1.0};
  NK_TRY(cholesky_fail_flags(ctx, sys, 2, chol_failed, piv_ratio));  // synchronises the main stream (which has joined the side stream)
  tr.mark("final sync");
  int rank_sys[2] = {mp, m};
  bool redo_products = false;
  if (ctx->strict_spd == 2) chol_failed[0] = chol_failed[1] = -1;  // lstsq-shaped: always the SVD with gelsd's cut-off
  if (chol_failed[0] || chol_failed[1]) {
    if (ctx->strict_spd == 1) {  // NK_ERR_NOT_SPD
      set_error("Cholesky: system %d is numerically rank deficient (non-positive or rounding-level pivot; the reference's "
                "lstsq truncates here) and strict mode is on", chol_failed[0] ? 0 : 1);
      return NK_ERR_NOT_SPD;
    }
    // numerically rank-deficient system(s): lstsq's (gelsd's) minimum-norm solution, singular values <= eps * sigma_max
    // dropped (nk_pinv.hip)
    const double rcond = 2.220446049250313e-16;
    for (int q = 0; q < 2; ++q) {
      if (!chol_failed[q]) continue;
      PinvInfo pi;
      if (q == 0)  // cross inner^+  ->  G2
        NK_TRY(pinv_right_divide(ctx, Gsave, mp, mp, Gsave + (size_t)mp * mp, mp, m, G2, mp, rcond, &pi));
      else         // left_rec inner_rec^+  ->  G4
        NK_TRY(pinv_right_divide(ctx, Gsave + gram_block1(m, p), m, m, Gsave + gram_block1(m, p) + (size_t)m * m, m, d, G4,
                                 m, rcond, &pi));
      if (!pi.converged) {
        set_error("rank-revealing fallback: Jacobi SVD of system %d did not converge in %d sweeps", q, pi.sweeps);
        return NK_ERR_NO_CONVERGENCE;
      }
      rank_sys[q] = pi.rank;
    }
    // (tried in round 3: solving the systems whose pivots decay gradually through the rounding level -- no spectral gap,
    // the gamma = 1e-7 candidates of the cloth grid -- by a minimally shifted Cholesky instead of the SVD.  4 x faster grid
    // (0.29 s), but a shift of 4 m eps ||P|| is 2000 x gelsd's eps sigma_max cut-off: 15 of the 405 units moved 1.5e-2 .. 0.37
    // away from the reference's score, against <= 1e-2 with the SVD.  Dropped.)
    count_event(CNT_RANK_TRUNCATED);
    redo_products = true;
  }
  // ---- optional refinement of ill-conditioned systems (nk_set_refine / NYSKOOP_REFINE_PIVOT; off by default).  Each step
  //      forms the residual R - X inner from the SAVED system in doubled precision (launch_resid_dd: a plain fp64 residual
  //      is all rounding error and makes things worse) and solves for the correction with the same factor; a step is
  //      applied only while the corrections contract (decided on the device).  The solution then is the system's own to
  //      working precision -- what is left against the reference is the reference's rounding (gelsd) and the Gram
  //      products' summation order.  With the backward-stable blocked solve (chol_panel_kernel) this buys little: config 2
  //      A 1.5e-4 -> 1.15e-4 from the reference whose own row-order spread is 1.0e-4; it costs 10 flop per term on the
  //      vector ALU (0.15 s on the 405-unit cloth grid), hence opt-in.
  int refined[2] = {0, 0};
  {
    const double refine_below = ctx->refine_pivot;
    const int refine_steps = ctx->refine_steps;
    double* refine_state = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)8, &refine_state));
    for (int q = 0; q < 2 && refine_steps > 0; ++q) {
      if (chol_failed[q] || !(piv_ratio[q] > 0.0) || piv_ratio[q] >= refine_below) continue;
      const int mq = sys[q].m, nr = sys[q].extra;                       // system size, number of right-hand sides (rows)
      const size_t off = q == 0 ? 0 : gram_block1(m, p);
      const double* Pq = Gsave + off;                                   // saved system (symmetric)
      const double* Rq = Gsave + off + (size_t)mq * mq;                 // saved right-hand-side rows (nr x mq)
      double* Xq = G1 + off + (size_t)mq * mq;                          // solution rows (nr x mq) = R P^-1
      const ArenaMark mk = arena_mark(ctx);
      double *Res = nullptr, *ResT = nullptr, *partial = nullptr;
      const int64_t ldt_ = nr + (nr & 1);
      NK_TRY(arena_alloc_t(ctx, (size_t)nr * mq, &Res));
      NK_TRY(arena_alloc_t(ctx, (size_t)mq * ldt_, &ResT));
      NK_TRY(arena_alloc_t(ctx, (size_t)2 * refine_partial_blocks(), &partial));
      CholSys y = sys[q];
      y.R = ResT; y.ldr = ldt_; y.nrhs = nr;
      for (int step = 0; step < refine_steps; ++step) {
        NK_TRY(launch_resid_dd(ctx, Xq, mq, Pq, mq, Rq, mq, Res, mq, nr, mq));                    // Res = R - X P
        NK_TRY(launch_transpose(ctx, Res, mq, ResT, ldt_, nr, mq));                               // columns for the solve
        NK_TRY(cholesky_solve_pair(ctx, &y, 1));                                                  // P dX^T = Res^T
        NK_TRY(launch_transpose(ctx, ResT, ldt_, Res, mq, mq, nr));
        // X += dX while the corrections contract (a numerically singular system that happened to factor is left alone)
        NK_TRY(launch_refine_apply(ctx, Res, mq, Xq, mq, nr, mq, step, refine_state + 4 * q, partial));
      }
      NK_HIP(hipMemcpyAsync(ctx->h_scalars + 16 + 4 * q, refine_state + 4 * q, 4 * sizeof(double), hipMemcpyDeviceToHost,
                            ctx->stream));
      arena_release(ctx, mk);
      refined[q] = -1;  // verdict in h_scalars[16 + 4 q ..] after the synchronisation below
      redo_products = true;
    }
  }
  {
    const int vr = sqrtm_verdict(ctx, &splan, &it, &resid);  // the iteration was queued without host round trips
    if (vr == NK_SQRT_RETRY) {
      count_event(CNT_SQRT_RETRY);
      // K_mm + jitter I is not positive definite to working precision (or the eigenvalue bound did not hold): the
      // coupled iteration needs no factorisation; then everything that depends on the square root once more
      NK_TRY(sqrtm_spd_coupled(ctx, Kj, m, m, mdl->S, mdl->Sinv, &it, &resid));
      NK_TRY(sqrt_products());
      redo_products = true;
    } else {
      NK_TRY(vr);
    }
  }
  if (redo_products) {
    NK_TRY(operator_products());
    NK_HIP(hipStreamSynchronize(ctx->stream));
  }
  double refine_ratio[2] = {0.0, 0.0};
  for (int q = 0; q < 2; ++q)
    if (refined[q] < 0) {
      refined[q] = (int)ctx->h_scalars[16 + 4 * q + 2];  // steps accepted by the contraction guard
      refine_ratio[q] = ctx->h_scalars[16 + 4 * q + 3];
    }
  if (refined[0] > 0 || refined[1] > 0) count_event(CNT_REFINED);
  mdl->has_ops = true;

  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->ms_total = ev_ms(ctx, 0, 5);
    stats->ms_upload = (x.staged || y.staged) ? ev_ms(ctx, 0, 1) : 0.0;
    stats->ms_kmat = ev_ms(ctx, 1, 2);
    stats->ms_gram = ev_ms(ctx, 2, 3);
    stats->ms_sqrt = ev_ms(ctx, 6, 7);  // on the side stream, overlapping the kernel-block and Gram stages
    stats->ms_solve = ev_ms(ctx, 4, 5);
    if (gram_deferred) ms_gram_kernel = ev_ms(ctx, 14, 15);
    stats->ms_gram_kernel_avg = gram_launches ? ms_gram_kernel / gram_launches : 0.0;
    stats->gram_kernel_launches = gram_launches;
    stats->sqrt_iters = it;
    stats->sqrt_residual = resid;
    const double ne = (double)n_eff;
    const double t128 = 128.0;
    auto tiles = [&](double v) { return std::ceil(v / t128); };
    const double tmp_ = tiles(mp), tm_ = tiles(m);
    // flop actually issued by the three big tile sets (upper-triangular tile sets for the symmetric Grams)
    stats->gram_flops = 2.0 * ne * t128 * t128 * (tmp_ * (tmp_ + 1) / 2 + tm_ * tmp_ + tm_ * (tm_ + 1) / 2) +
                        2.0 * ne * (double)d * m;
    stats->kmat_pairs = 2.0 * ne * m * d + (same_centers ? 1.0 : 3.0) * (double)m * m * d;
    stats->rank_inner = rank_sys[0];
    stats->rank_inner_rec = rank_sys[1];
    stats->pivot_ratio_inner = piv_ratio[0];
    stats->pivot_ratio_inner_rec = piv_ratio[1];
    stats->refined = refined[0] + 16 * refined[1];
    stats->refine_ratio_inner = refine_ratio[0];
    stats->refine_ratio_inner_rec = refine_ratio[1];
  }
  tr.mark("stats");
  if (ctx->arena.chunks.size() > 1 || ctx->arena_side.chunks.size() > 1) NK_TRY(arena_reset(ctx));  // coalesce now (everything is synchronised), not in the next call
  guard.m = nullptr;
  *model = mdl;
  return NK_OK;
}

int nk_nystrom_fit(nk_ctx* ctx, const nk_kernel_desc* kd, const double* X, int64_t ldx, const double* Y, int64_t ldy,
                   int64_t n, int32_t d, int32_t p, const int64_t* row_ranges, int32_t n_ranges, const double* Zin,
                   int64_t ldzi, const double* Zout, int64_t ldzo, int32_t m, double gamma, double jitter,
                   nk_model** model, nk_fit_stats* stats) {
  NK_REQUIRE(model != nullptr, "nk_nystrom_fit: null argument");
  return fit_impl(ctx, kd, X, ldx, Y, ldy, n, d, p, row_ranges, n_ranges, Zin, ldzi, Zout, ldzo, m, gamma, jitter, model,
                  stats, FIT_FULL, nullptr);
}

int nk_gram_doubles(int32_t m, int32_t d, int32_t p, int64_t* count) {
  NK_REQUIRE(count && m > 0 && d > 0 && p >= 0, "nk_gram_doubles: bad argument");
  *count = (int64_t)gram_doubles(m, d, p);
  return NK_OK;
}

int nk_nystrom_gram(nk_ctx* ctx, const nk_kernel_desc* kd, const double* X, int64_t ldx, const double* Y, int64_t ldy,
                    int64_t n, int32_t d, int32_t p, const int64_t* row_ranges, int32_t n_ranges, const double* Zin,
                    int64_t ldzi, const double* Zout, int64_t ldzo, int32_t m, double* gram, nk_fit_stats* stats) {
  return fit_impl(ctx, kd, X, ldx, Y, ldy, n, d, p, row_ranges, n_ranges, Zin, ldzi, Zout, ldzo, m, 0.0, 0.0, nullptr,
                  stats, FIT_GRAM, gram);
}

int nk_nystrom_solve(nk_ctx* ctx, const nk_kernel_desc* kd, const double* Zin, int64_t ldzi, const double* Zout,
                     int64_t ldzo, int32_t m, int32_t d, int32_t p, const double* gram, int64_t n_total, double gamma,
                     double jitter, nk_model** model, nk_fit_stats* stats) {
  NK_REQUIRE(model != nullptr, "nk_nystrom_solve: null argument");
  return fit_impl(ctx, kd, nullptr, 0, nullptr, 0, n_total, d, p, nullptr, 0, Zin, ldzi, Zout, ldzo, m, gamma, jitter,
                  model, stats, FIT_SOLVE, const_cast<double*>(gram));
}

int nk_model_create(nk_ctx* ctx, const nk_kernel_desc* kd, const double* Zout, int64_t ldz, int32_t m, int32_t d,
                    int32_t p, double jitter, const double* A, const double* B, const double* C, const double* W,
                    nk_model** model) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(kd && Zout && model, "nk_model_create: null argument");
  NK_REQUIRE(m > 0 && d > 0 && p >= 0 && ldz >= d, "nk_model_create: bad sizes");
  *model = nullptr;
  nk_model* mdl = nullptr;
  NK_TRY(model_alloc(ctx, m, d, p, &mdl));
  struct Guard { nk_model* m; ~Guard() { if (m) nk_model_destroy(m); } } guard{mdl};
  mdl->ktype = kd->type; mdl->sigma0 = kd->sigma0; mdl->jitter = jitter;
  NK_TRY(make_winv(ctx, kd, d, mdl->winv));
  MatIn z;
  NK_TRY(stage_in(ctx, Zout, ldz, m, d, &z));
  NK_TRY(launch_copy2d(ctx, z.ptr, z.ld, mdl->Z, d, m, d));
  double* Kj = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &Kj));
  NK_TRY(launch_kmat(ctx, kd->type, mdl->Z, d, m, mdl->Z, d, m, d, mdl->winv, kd->sigma0, Kj, m));
  NK_TRY(launch_add_diag(ctx, Kj, m, m, jitter));
  NK_TRY(sqrtm_spd(ctx, Kj, m, m, mdl->S, mdl->Sinv, nullptr, nullptr));
  const int mp = m + p;
  if (A && C) {
    NK_REQUIRE(p == 0 || B != nullptr, "nk_model_create: B missing");
    MatIn a, b, c, w;
    NK_TRY(stage_in(ctx, A, m, m, m, &a));
    NK_TRY(launch_copy2d(ctx, a.ptr, a.ld, mdl->A, mp, m, m));
    if (p > 0) {
      NK_TRY(stage_in(ctx, B, p, m, p, &b));
      NK_TRY(launch_copy2d(ctx, b.ptr, b.ld, mdl->B, mp, m, p));
    }
    NK_TRY(stage_in(ctx, C, m, d, m, &c));
    NK_TRY(launch_copy2d(ctx, c.ptr, c.ld, mdl->C, m, d, m));
    if (W) {
      NK_TRY(stage_in(ctx, W, mp, d, mp, &w));
      NK_TRY(launch_copy2d(ctx, w.ptr, w.ld, mdl->W, mp, d, mp));
    } else {
      NK_TRY(launch_gemm(ctx, false, false, d, mp, m, 1.0, mdl->C, m, mdl->A, mp, 0.0, mdl->W, mp));
    }
    mdl->has_ops = true;
  }
  NK_HIP(hipStreamSynchronize(ctx->stream));
  guard.m = nullptr;
  *model = mdl;
  return NK_OK;
}

int nk_model_destroy(nk_model* model) {
  if (!model) return NK_OK;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_models.erase(model) == 0) return NK_OK;  // already destroyed (nk_shutdown, or a second call)
  }
  destroy_model_unregistered(model, true);
  return NK_OK;
}

int nk_model_dims(const nk_model* model, int32_t* m, int32_t* d, int32_t* p) {
  NK_REQUIRE(model != nullptr, "null model");
  if (m) *m = model->m;
  if (d) *d = model->d;
  if (p) *p = model->p;
  return NK_OK;
}

int nk_model_get(nk_ctx* ctx, const nk_model* mdl, char which, double* out, int64_t ldo) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl && out, "nk_model_get: null argument");
  const int m = mdl->m, d = mdl->d, p = mdl->p, mp = m + p;
  const double* src = nullptr;
  int64_t lds = 0, rows = 0, cols = 0;
  switch (which) {
    case 'A': src = mdl->A; lds = mp; rows = m; cols = m; break;
    case 'B': src = mdl->B; lds = mp; rows = m; cols = p; break;
    case 'C': src = mdl->C; lds = m; rows = d; cols = m; break;
    case 'W': src = mdl->W; lds = mp; rows = d; cols = mp; break;
    case 'S': src = mdl->S; lds = m; rows = m; cols = m; break;
    case 'I': src = mdl->Sinv; lds = m; rows = m; cols = m; break;
    case 'Z': src = mdl->Z; lds = d; rows = m; cols = d; break;
    default: set_error("nk_model_get: unknown selector '%c'", which); return NK_ERR_BAD_ARG;
  }
  if ((which == 'A' || which == 'B' || which == 'C' || which == 'W') && !mdl->has_ops) {
    set_error("nk_model_get: model holds no fitted operators");
    return NK_ERR_BAD_ARG;
  }
  if (rows == 0 || cols == 0) return NK_OK;
  NK_REQUIRE(ldo >= cols, "nk_model_get: leading dimension too small");
  NK_HIP(hipMemcpy2DAsync(out, (size_t)ldo * 8, src, (size_t)lds * 8, (size_t)cols * 8, (size_t)rows,
                          is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

int nk_model_get_ops(nk_ctx* ctx, const nk_model* mdl, double* G, int64_t ldg, double* Cm, int64_t ldc, double* W,
                     int64_t ldw) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl != nullptr, "nk_model_get_ops: null model");
  if (!mdl->has_ops) {
    set_error("nk_model_get_ops: model holds no fitted operators");
    return NK_ERR_BAD_ARG;
  }
  const int m = mdl->m, d = mdl->d, mp = m + mdl->p;
  NK_REQUIRE((!G || ldg >= mp) && (!Cm || ldc >= m) && (!W || ldw >= mp), "nk_model_get_ops: leading dimension too small");
  auto copy = [&](double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols,
                  hipStream_t st) -> hipError_t {
    const hipMemcpyKind kind = is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (ldd == cols && lds == cols) return hipMemcpyAsync(dst, src, (size_t)rows * cols * 8, kind, st);
    return hipMemcpy2DAsync(dst, (size_t)ldd * 8, src, (size_t)lds * 8, (size_t)cols * 8, (size_t)rows, kind, st);
  };
  // the big block on one stream, the two small ones on another: two DMA engines work at once
  if (G) NK_HIP(copy(G, ldg, mdl->A, mp, m, mp, ctx->stream_main));
  if (Cm) NK_HIP(copy(Cm, ldc, mdl->C, m, d, m, ctx->stream_side));
  if (W) NK_HIP(copy(W, ldw, mdl->W, mp, d, mp, ctx->stream_side));
  NK_HIP(hipStreamSynchronize(ctx->stream_side));
  NK_HIP(hipStreamSynchronize(ctx->stream_main));
  return NK_OK;
}

int nk_model_get_ops_async(nk_ctx* ctx, nk_model* mdl, double* G, int64_t ldg, double* Cm, int64_t ldc, double* W,
                           int64_t ldw) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl != nullptr, "nk_model_get_ops_async: null model");
  if (!mdl->has_ops) {
    set_error("nk_model_get_ops_async: model holds no fitted operators");
    return NK_ERR_BAD_ARG;
  }
  const int m = mdl->m, d = mdl->d, mp = m + mdl->p;
  NK_REQUIRE((!G || ldg >= mp) && (!Cm || ldc >= m) && (!W || ldw >= mp),
             "nk_model_get_ops_async: leading dimension too small");
  NK_TRY(nk_model_wait(mdl));  // one fetch in flight per model
  // everything that produced the operators was synchronised by the producing call (fit / model_create)
  auto copy = [&](double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols) -> hipError_t {
    const hipMemcpyKind kind = is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (ldd == cols && lds == cols) return hipMemcpyAsync(dst, src, (size_t)rows * cols * 8, kind, ctx->stream_copy);
    return hipMemcpy2DAsync(dst, (size_t)ldd * 8, src, (size_t)lds * 8, (size_t)cols * 8, (size_t)rows, kind,
                            ctx->stream_copy);
  };
  if (G) NK_HIP(copy(G, ldg, mdl->A, mp, m, mp));
  if (Cm) NK_HIP(copy(Cm, ldc, mdl->C, m, d, m));
  if (W) NK_HIP(copy(W, ldw, mdl->W, mp, d, mp));
  if (ctx_recording(ctx)) {
    // member of a lock-step group: the copies above were RECORDED in this member's sequence and no event exists that
    // another thread could wait on, so the fetch completes here (one flush), and nothing is left pending on the model
    NK_HIP(hipStreamSynchronize(ctx->stream));
    return NK_OK;
  }
  hipEvent_t ev = nullptr;
  NK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, ctx->stream_copy);
  if (e != hipSuccess) {
    (void)hipEventDestroy(ev);
    (void)hipStreamSynchronize(ctx->stream_copy);
    set_error("hipEventRecord failed: %s", hipGetErrorString(e));
    return NK_ERR_HIP;
  }
  mdl->ev_fetch = ev;
  return NK_OK;
}

static int model_wait_unchecked(nk_model* mdl);

int nk_model_wait(nk_model* mdl) {
  NK_REQUIRE(mdl != nullptr, "nk_model_wait: null model");
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_models.count(mdl) == 0) return NK_OK;  // released by nk_shutdown (which waited for the fetch itself)
  }
  return model_wait_unchecked(mdl);
}

static int model_wait_unchecked(nk_model* mdl) {
  if (mdl->ev_fetch) {
    // the REAL wait, whatever context the calling thread last used: a pending fetch is always a recorded event on a
    // copy stream (fetches by group members complete inside nk_model_get_ops_async), so this must not be turned into a
    // round barrier of the caller's group
    hipError_t e = real_event_sync(mdl->ev_fetch);
    (void)hipEventDestroy(mdl->ev_fetch);
    mdl->ev_fetch = nullptr;
    if (e != hipSuccess) {
      set_error("hipEventSynchronize failed: %s", hipGetErrorString(e));
      return NK_ERR_HIP;
    }
  }
  return NK_OK;
}

// ---- small-call staging: the latency-bound entry points (rollouts, closed loops) read their host inputs from, and write
//      their host outputs into, one page-locked block that the GPU addresses directly -- no DMA descriptors, no staging
//      copies on the stream; the host moves the bytes with memcpy before the launch and after the one synchronisation.
struct SmallStage {
  nk_ctx* ctx = nullptr;
  size_t off = 0;
  struct Out { double* stage; double* user; int64_t user_ld; int64_t rows, cols; };
  std::vector<Out> outs;
};
constexpr size_t SMALL_STAGE_LIMIT = (size_t)4 << 20;
static int small_reserve(nk_ctx* ctx, size_t bytes) {
  if (ctx->h_stage_bytes >= bytes) return NK_OK;
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  ctx->h_stage = nullptr;
  ctx->h_stage_bytes = 0;
  NK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), bytes, hipHostMallocDefault));
  ctx->h_stage_bytes = bytes;
  return NK_OK;
}
static const double* small_in(SmallStage& st, const double* host, int64_t ld, int64_t rows, int64_t cols) {
  double* dst = reinterpret_cast<double*>(reinterpret_cast<char*>(st.ctx->h_stage) + st.off);
  for (int64_t r = 0; r < rows; ++r) memcpy(dst + r * cols, host + r * ld, (size_t)cols * 8);
  st.off += (((size_t)rows * cols * 8) + 255) & ~(size_t)255;
  return dst;
}
static double* small_out(SmallStage& st, double* user, int64_t user_ld, int64_t rows, int64_t cols) {
  double* dst = reinterpret_cast<double*>(reinterpret_cast<char*>(st.ctx->h_stage) + st.off);
  st.off += (((size_t)rows * cols * 8) + 255) & ~(size_t)255;
  st.outs.push_back(SmallStage::Out{dst, user, user_ld, rows, cols});
  return dst;
}
static void small_finish(SmallStage& st) {  // after the stream has been synchronised
  for (auto& o : st.outs)
    for (int64_t r = 0; r < o.rows; ++r) memcpy(o.user + r * o.user_ld, o.stage + r * o.cols, (size_t)o.cols * 8);
}
static inline size_t pad256(size_t doubles) { return ((doubles * 8) + 255) & ~(size_t)255; }

int nk_lift(nk_ctx* ctx, const nk_model* mdl, const double* Xq, int64_t ldx, int64_t nq, double* out, int64_t ldo) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl && Xq && out, "nk_lift: null argument");
  NK_REQUIRE(nq >= 0 && ldx >= mdl->d && ldo >= mdl->m, "nk_lift: bad sizes");
  if (nq == 0) return NK_OK;
  // a few states (one per tick of a controller that closes the loop on a plant, benchmark_lqr_hjb.py:73-97): through the
  // page-locked block, no hipMemcpy of pageable memory (whose completion wait alone costs 30-100 us, depending on how
  // the runtime decides to wait)
  const size_t need = pad256((size_t)nq * mdl->d) + pad256((size_t)nq * mdl->m);
  if (need <= SMALL_STAGE_LIMIT && !is_device_ptr(Xq) && !is_device_ptr(out)) {
    SmallStage st;
    st.ctx = ctx;
    NK_TRY(small_reserve(ctx, need));
    const double* xh = small_in(st, Xq, ldx, nq, mdl->d);
    double* oh = small_out(st, out, ldo, nq, mdl->m);
    double* xd = nullptr;  // every wave of the kernel-matrix kernel reads its state row: from HBM, not over PCIe
    NK_TRY(arena_alloc_t(ctx, (size_t)nq * mdl->d, &xd));
    NK_TRY(launch_copy2d(ctx, xh, mdl->d, xd, mdl->d, nq, mdl->d));
    NK_TRY(lift_device(ctx, mdl, xd, mdl->d, nq, oh, mdl->m));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    small_finish(st);
    return NK_OK;
  }
  MatIn x;
  NK_TRY(stage_in(ctx, Xq, ldx, nq, mdl->d, &x));
  MatOut o;
  NK_TRY(stage_out(ctx, out, ldo, nq, mdl->m, &o));
  NK_TRY(lift_device(ctx, mdl, x.ptr, x.ld, nq, o.dev, o.ld));
  NK_TRY(finish_out(ctx, o));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

int nk_predict(nk_ctx* ctx, const nk_model* mdl, const double* Xaug, int64_t ldx, int64_t nq, double* out,
               int64_t ldo) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl && Xaug && out, "nk_predict: null argument");
  NK_REQUIRE(mdl->has_ops, "nk_predict: model holds no fitted operators");
  NK_REQUIRE(nq >= 0 && ldx >= mdl->d + mdl->p && ldo >= mdl->d, "nk_predict: bad sizes");
  if (nq == 0) return NK_OK;
  const int64_t dp = mdl->d + mdl->p;
  const size_t need = pad256((size_t)nq * dp) + pad256((size_t)nq * mdl->d);
  if (need <= SMALL_STAGE_LIMIT && !is_device_ptr(Xaug) && !is_device_ptr(out)) {  // see nk_lift
    SmallStage st;
    st.ctx = ctx;
    NK_TRY(small_reserve(ctx, need));
    const double* xh = small_in(st, Xaug, ldx, nq, dp);
    double* oh = small_out(st, out, ldo, nq, mdl->d);
    const int64_t ldd = dp + (dp & 1);
    double* xd = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)nq * ldd, &xd));
    NK_TRY(launch_copy2d(ctx, xh, dp, xd, ldd, nq, dp));
    NK_TRY(predict_device(ctx, mdl, xd, ldd, nq, oh, mdl->d));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    small_finish(st);
    return NK_OK;
  }
  MatIn x;
  NK_TRY(stage_in(ctx, Xaug, ldx, nq, mdl->d + mdl->p, &x));
  MatOut o;
  NK_TRY(stage_out(ctx, out, ldo, nq, mdl->d, &o));
  NK_TRY(predict_device(ctx, mdl, x.ptr, x.ld, nq, o.dev, o.ld));
  NK_TRY(finish_out(ctx, o));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

int nk_score_neg_rmse(nk_ctx* ctx, const nk_model* mdl, const double* Xaug, int64_t ldx, const double* Ytrue,
                      int64_t ldy, int64_t nq, double* score) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl && Xaug && Ytrue && score, "nk_score_neg_rmse: null argument");
  NK_REQUIRE(mdl->has_ops, "nk_score_neg_rmse: model holds no fitted operators");
  NK_REQUIRE(nq > 0 && ldx >= mdl->d + mdl->p && ldy >= mdl->d, "nk_score_neg_rmse: bad sizes");
  const int d = mdl->d;
  MatIn x, y;
  NK_TRY(stage_in(ctx, Xaug, ldx, nq, d + mdl->p, &x));
  NK_TRY(stage_in(ctx, Ytrue, ldy, nq, d, &y));
  double *P = nullptr, *colsum = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)nq * d, &P));
  NK_TRY(arena_alloc_t(ctx, (size_t)d, &colsum));
  NK_TRY(predict_device(ctx, mdl, x.ptr, x.ld, nq, P, d));
  NK_TRY(launch_colsum_sqdiff(ctx, P, d, y.ptr, y.ld, nq, d, colsum));
  std::vector<double> h((size_t)d);
  NK_HIP(hipMemcpyAsync(h.data(), colsum, sizeof(double) * d, hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  double s = 0.0;
  for (int j = 0; j < d; ++j) s += std::sqrt(h[j] / (double)nq);
  *score = -s / d;
  return NK_OK;
}

// z_{t+1} = G [z_t; u_t] (+ bias) for t < T-1 on Zall ([b][t][m], row 0 of every trajectory already holds z_0 unless
// `chain.lift`), then x = C z for every (b, t).  One launch for the recursion when G fits in LDS, a matrix-vector /
// GEMM launch per step otherwise.
static bool chain_mw_enabled() {
  static const bool on = [] { const char* e = getenv("NYSKOOP_CHAIN_MW"); return !(e && e[0] == '0'); }();
  return on;
}
// One multi-workgroup recursion on the device at a time (two side by side can starve each other of workgroup slots,
// nk_rollout.hip): held from the launch to the synchronisation that ends the call.
static std::mutex g_chain_mw_mutex;

// Is the single-launch multi-workgroup recursion the path for this chain?  Not inside a lock-step group (its launches
// are deferred to the group's flush, the mutex could not cover them), not for more trajectories than fit beside each
// other unless they are few (<= 16: chunks of launches still beat a launch per step; beyond that the per-step GEMM
// amortises its launches over the batch).
static bool chain_mw_wanted(nk_ctx* ctx, const ChainArgs& chain) {
  if (!chain_mw_enabled() || ctx_recording(ctx) || chain.T < 3) return false;
  if (lifted_chain_ok(chain.m, chain.pu, chain.lift ? chain.d : 0)) return false;
  if (!lifted_chain_mw_ok(ctx, chain.m, chain.U ? chain.pu : 0)) return false;
  const int nt = chain_mw_group(chain.m, chain.batch);
  const int64_t groups = (chain.batch + nt - 1) / nt;
  return groups <= 16 || groups * chain_mw_workgroups(chain.m) <= ctx->num_cu;
}

static bool chain_mw_gave_up(nk_ctx* ctx) {
  int row = 0, step = 0, traj = 0;
  if (!lifted_chain_mw_timed_out(ctx, &row, &step, &traj)) {
    const char* hook = getenv("NYSKOOP_CHAIN_MW_TEST_GIVEUP");  // test hook: pretend the wait gave up (tests/)
    const bool forced = hook != nullptr && hook[0] == '1';
    if (forced) count_event(CNT_CHAIN_GIVEUP);
    return forced;
  }
  count_event(CNT_CHAIN_GIVEUP);
  if (getenv("NYSKOOP_TRACE"))
    fprintf(stderr, "[nyskoop] single-launch recursion timed out (row %d, step %d, trajectory %d): repeating stepwise\n", row,
            step, traj);
  return true;
}

static int rollout_steps(nk_ctx* ctx, ChainArgs chain, bool z0_in_place, bool use_mw) {
  const int m = chain.m, p = chain.pu, T = chain.T, batch = chain.batch;
  if (lifted_chain_ok(m, p, chain.lift ? chain.d : 0)) return launch_lifted_chain(ctx, chain);
  NK_REQUIRE(!chain.lift && z0_in_place, "rollout_steps: internal: the stepwise path needs z_0 in place");
  if (use_mw) {
    // trajectories that are resident side by side: (CUs / workgroups per trajectory group) groups of `nt`
    const int nt = chain_mw_group(m, batch);
    int nb = ctx->num_cu / chain_mw_workgroups(m);
    if (nb < 1) nb = 1;
    nb *= nt;
    NK_TRY(lifted_chain_mw_reset(ctx));
    for (int b0 = 0; b0 < batch; b0 += nb) {
      ChainArgs sub = chain;
      sub.batch = batch - b0 < nb ? batch - b0 : nb;
      sub.Zall = chain.Zall + (int64_t)b0 * chain.z_stride;
      if (chain.U) sub.U = chain.U + (int64_t)b0 * chain.u_stride;
      if (chain.bias) sub.bias = chain.bias + (int64_t)b0 * chain.bias_stride;
      NK_TRY(launch_lifted_chain_mw(ctx, sub));
    }
    return lifted_chain_mw_fetch_status(ctx);
  }
  double* Zall = chain.Zall;
  const int64_t ldz = chain.z_stride;
  for (int t = 0; t + 1 < T; ++t) {
    if (batch <= 16) {  // matrix-vector chain: one wave per row of G, trajectories in groups of 8
      for (int b0 = 0; b0 < batch; b0 += 8) {
        const int nb = batch - b0 < 8 ? batch - b0 : 8;
        // the kernel takes one bias vector: trajectories with their own bias go one by one
        if (chain.bias && chain.bias_stride != 0) {
          for (int b = b0; b < b0 + nb; ++b)
            NK_TRY(launch_lifted_step(ctx, chain.G, chain.ldg, m, m, p, Zall + (int64_t)b * ldz + (int64_t)t * m, ldz,
                                      p > 0 ? chain.U + (int64_t)b * chain.u_stride + (int64_t)t * p : nullptr,
                                      chain.u_stride, chain.bias + (int64_t)b * chain.bias_stride,
                                      Zall + (int64_t)b * ldz + (int64_t)(t + 1) * m, ldz, 1));
        } else {
          NK_TRY(launch_lifted_step(ctx, chain.G, chain.ldg, m, m, p, Zall + (int64_t)b0 * ldz + (int64_t)t * m, ldz,
                                    p > 0 ? chain.U + (int64_t)b0 * chain.u_stride + (int64_t)t * p : nullptr,
                                    chain.u_stride, chain.bias, Zall + (int64_t)b0 * ldz + (int64_t)(t + 1) * m, ldz, nb));
        }
      }
    } else {
      double* zn = Zall + (int64_t)(t + 1) * m;
      double beta = 0.0;
      if (chain.bias) {  // z' = bias + ...
        NK_TRY(launch_copy2d(ctx, chain.bias, chain.bias_stride, zn, ldz, batch, m));
        beta = 1.0;
      }
      NK_TRY(launch_gemm(ctx, false, true, batch, m, m, 1.0, Zall + (int64_t)t * m, ldz, chain.G, chain.ldg, beta, zn, ldz));
      if (p > 0)
        NK_TRY(launch_gemm(ctx, false, true, batch, m, p, 1.0, chain.U + (int64_t)t * p, chain.u_stride, chain.G + m,
                           chain.ldg, 1.0, zn, ldz));
    }
  }
  return NK_OK;
}

static int rollout_impl(nk_ctx* ctx, const nk_model* mdl, const double* G, int64_t ldg, const double* Cop, int64_t ldc,
                        int m, int d, int p, const double* x0, int64_t ldx0, const double* z0, const double* U, int32_t T,
                        int32_t batch, double* out_x, double* out_z) {
  // x0 != nullptr: lift through the model; otherwise z0 (batch x m) holds the lifted initial states
  const int64_t nin = x0 ? d : m;
  const double* first = x0 ? x0 : z0;
  const bool have_u = p > 0 && T > 1;
  const size_t need = pad256((size_t)batch * nin) + (have_u ? pad256((size_t)batch * T * p) : 0) +
                      pad256((size_t)batch * T * d) + (out_z ? pad256((size_t)batch * T * m) : 0);
  const bool small = need <= SMALL_STAGE_LIMIT && !is_device_ptr(first) && !(have_u && is_device_ptr(U)) &&
                     !is_device_ptr(out_x) && !(out_z && is_device_ptr(out_z));
  double* Zall = nullptr;  // [batch][T][m]
  NK_TRY(arena_alloc_t(ctx, (size_t)batch * T * m, &Zall));
  const int64_t ldz = (int64_t)T * m;
  ChainArgs ch;
  ch.G = G; ch.ldg = ldg; ch.m = m; ch.pu = p; ch.T = T; ch.batch = batch; ch.Zall = Zall; ch.z_stride = ldz;
  ch.u_stride = (int64_t)T * p;
  SmallStage st;
  st.ctx = ctx;
  MatIn xin, uin;
  MatOut ox, oz;
  double *xdev = nullptr, *zdev = nullptr;
  if (small) {
    NK_TRY(small_reserve(ctx, need));
    xin.ptr = small_in(st, first, x0 ? ldx0 : m, batch, nin);
    xin.ld = nin;
    if (have_u) { uin.ptr = small_in(st, U, (int64_t)T * p, batch, (int64_t)T * p); uin.ld = (int64_t)T * p; }
    xdev = small_out(st, out_x, d, (int64_t)batch * T, d);
    if (out_z) zdev = small_out(st, out_z, m, (int64_t)batch * T, m);
  } else {
    NK_TRY(stage_in(ctx, first, x0 ? ldx0 : m, batch, nin, &xin));
    if (have_u) NK_TRY(stage_in(ctx, U, (int64_t)T * p, batch, (int64_t)T * p, &uin));
    NK_TRY(stage_out(ctx, out_x, d, (int64_t)batch * T, d, &ox));
    xdev = ox.dev;
    if (out_z) { NK_TRY(stage_out(ctx, out_z, m, (int64_t)batch * T, m, &oz)); zdev = oz.dev; }
  }
  ch.U = have_u ? uin.ptr : nullptr;
  if (!have_u) ch.pu = (T > 1) ? p : 0;
  const int64_t ldxo = small ? d : ox.ld, ldzo = small ? m : (out_z ? oz.ld : m);
  bool z0_in_place = false;
  if (x0 && lifted_chain_ok(m, ch.pu, d)) {  // the lift is done by the chain kernel itself
    ch.lift = true; ch.x0 = xin.ptr; ch.x0_stride = xin.ld; ch.Zl = mdl->Z; ch.d = d; ch.winv = mdl->winv;
    ch.Sinv = mdl->Sinv; ch.ktype = mdl->ktype; ch.sigma0 = mdl->sigma0;
  } else if (x0) {
    NK_TRY(lift_device(ctx, mdl, xin.ptr, xin.ld, batch, Zall, ldz));  // z_0 = phi(x_0) for every trajectory
    z0_in_place = true;
  } else if (lifted_chain_ok(m, ch.pu, 0)) {
    ch.z0 = xin.ptr; ch.z0_stride = xin.ld;
  } else {
    NK_TRY(launch_copy2d(ctx, xin.ptr, xin.ld, Zall, ldz, batch, m));
    z0_in_place = true;
  }
  if (z0_in_place && small && have_u) {
    // the stepwise / multi-workgroup paths read the controls from every wave of every step: not from page-locked host
    // memory (an uncached PCIe read per wave, ~30 us per step at m = 500) but from a device copy
    double* Udev = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)batch * T * p, &Udev));
    NK_TRY(launch_copy2d(ctx, uin.ptr, uin.ld, Udev, (int64_t)T * p, batch, (int64_t)T * p));
    ch.U = Udev;
  }
  const bool try_mw = chain_mw_wanted(ctx, ch);
  std::unique_lock<std::mutex> mw_lock(g_chain_mw_mutex, std::defer_lock);
  if (try_mw) mw_lock.lock();
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool mw = try_mw && attempt == 0;
    NK_TRY(rollout_steps(ctx, ch, z0_in_place, mw));
    NK_TRY(launch_gemm(ctx, false, true, (int64_t)batch * T, d, m, 1.0, Zall, m, Cop, ldc, 0.0, xdev, ldxo));
    if (out_z) NK_TRY(launch_copy2d(ctx, Zall, m, zdev, ldzo, (int64_t)batch * T, m));
    if (!small) {
      NK_TRY(finish_out(ctx, ox));
      if (out_z) NK_TRY(finish_out(ctx, oz));
    }
    NK_HIP(hipStreamSynchronize(ctx->stream));
    // a wave of the single-launch recursion gave up waiting for its neighbours (the device was oversubscribed): the
    // trajectories are not valid; z_0 is untouched, so the recursion is repeated with one launch per step
    if (!(mw && chain_mw_gave_up(ctx))) break;
  }
  if (small) small_finish(st);
  return NK_OK;
}

int nk_rollout(nk_ctx* ctx, const nk_model* mdl, const double* x0, int64_t ldx0, const double* U, int32_t T,
               int32_t batch, double* out_x, double* out_z) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl && x0 && out_x, "nk_rollout: null argument");
  NK_REQUIRE(mdl->has_ops, "nk_rollout: model holds no fitted operators");
  NK_REQUIRE(T >= 1 && batch >= 1 && ldx0 >= mdl->d, "nk_rollout: bad sizes");
  const int m = mdl->m, d = mdl->d, p = mdl->p, mp = m + p;
  NK_REQUIRE(p == 0 || T == 1 || U != nullptr, "nk_rollout: controls missing");
  return rollout_impl(ctx, mdl, mdl->A, mp, mdl->C, m, m, d, p, x0, ldx0, nullptr, U, T, batch, out_x, out_z);
}

int nk_linear_rollout(nk_ctx* ctx, const double* A, const double* B, const double* Cop, int32_t m, int32_t d, int32_t p,
                      const double* z0, const double* U, int32_t T, int32_t batch, double* out_x, double* out_z) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(A && Cop && z0 && out_x, "nk_linear_rollout: null argument");
  NK_REQUIRE(m >= 1 && d >= 1 && p >= 0 && T >= 1 && batch >= 1, "nk_linear_rollout: bad sizes");
  NK_REQUIRE(p == 0 || (B != nullptr && (T == 1 || U != nullptr)), "nk_linear_rollout: B or controls missing");
  const int mp = m + p;
  const int64_t ldg = mp + (mp & 1);
  double* G = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)m * ldg, &G));
  MatIn a, b, c;
  NK_TRY(stage_in(ctx, A, m, m, m, &a));
  NK_TRY(launch_copy2d(ctx, a.ptr, a.ld, G, ldg, m, m));
  if (p > 0) {
    NK_TRY(stage_in(ctx, B, p, m, p, &b));
    NK_TRY(launch_copy2d(ctx, b.ptr, b.ld, G + m, ldg, m, p));
  }
  NK_TRY(stage_in(ctx, Cop, m, d, m, &c));
  return rollout_impl(ctx, nullptr, G, ldg, c.ptr, c.ld, m, d, p, nullptr, 0, z0, U, T, batch, out_x, out_z);
}

int nk_closed_loop_batch(nk_ctx* ctx, const nk_model* mdl, const double* K, const double* phi0, const double* phi_ref,
                         int32_t steps, int32_t batch, double* out_x, double* out_u) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(mdl && K && phi0 && phi_ref && out_x && out_u, "nk_closed_loop: null argument");
  NK_REQUIRE(mdl->has_ops && steps >= 1 && batch >= 1 && mdl->p > 0, "nk_closed_loop: bad model or sizes");
  const int m = mdl->m, d = mdl->d, p = mdl->p, mp = m + p;
  const size_t need = pad256((size_t)p * m) + 2 * pad256((size_t)batch * m) + pad256((size_t)batch * steps * d) +
                      pad256((size_t)batch * steps * p);
  const bool small = need <= SMALL_STAGE_LIMIT && !is_device_ptr(K) && !is_device_ptr(phi0) && !is_device_ptr(phi_ref) &&
                     !is_device_ptr(out_x) && !is_device_ptr(out_u);
  SmallStage st;
  st.ctx = ctx;
  MatIn k, f0, fr;
  MatOut ox, ou;
  double *xdev = nullptr, *udev = nullptr;
  int64_t ldxo = d, lduo = p;
  if (small) {
    NK_TRY(small_reserve(ctx, need));
    k.ptr = small_in(st, K, m, p, m); k.ld = m;
    f0.ptr = small_in(st, phi0, m, batch, m); f0.ld = m;
    fr.ptr = small_in(st, phi_ref, m, batch, m); fr.ld = m;
    xdev = small_out(st, out_x, d, (int64_t)batch * steps, d);
    udev = small_out(st, out_u, p, (int64_t)batch * steps, p);
  } else {
    NK_TRY(stage_in(ctx, K, m, p, m, &k));
    NK_TRY(stage_in(ctx, phi0, m, batch, m, &f0));
    NK_TRY(stage_in(ctx, phi_ref, m, batch, m, &fr));
    NK_TRY(stage_out(ctx, out_x, d, (int64_t)batch * steps, d, &ox));
    NK_TRY(stage_out(ctx, out_u, p, (int64_t)batch * steps, p, &ou));
    xdev = ox.dev; udev = ou.dev; ldxo = ox.ld; lduo = ou.ld;
  }
  // phi_{t+1} = A phi_t + B K (phi_ref - phi_t) = (A - B K) phi_t + B K phi_ref: one matrix-vector step per time step
  // (algebraically the loop of benchmark_lqr_cloth.py:79-84; the controls u_t = K (phi_ref - phi_t) are recovered for all
  // steps at once afterwards)
  double *Phi = nullptr, *Acl = nullptr, *kref = nullptr, *cvec = nullptr, *Dm = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)batch * steps * m, &Phi));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &Acl));
  NK_TRY(arena_alloc_t(ctx, (size_t)batch * p + 8, &kref));
  NK_TRY(arena_alloc_t(ctx, (size_t)batch * m, &cvec));
  NK_TRY(arena_alloc_t(ctx, (size_t)batch * steps * m, &Dm));
  NK_TRY(launch_copy2d(ctx, mdl->A, mp, Acl, m, m, m));
  NK_TRY(launch_gemm(ctx, false, false, m, m, p, -1.0, mdl->B, mp, k.ptr, k.ld, 1.0, Acl, m));     // A - B K
  NK_TRY(launch_gemm(ctx, false, true, batch, p, m, 1.0, fr.ptr, fr.ld, k.ptr, k.ld, 0.0, kref, p));  // K phi_ref
  NK_TRY(launch_gemm(ctx, false, true, batch, m, p, 1.0, kref, p, mdl->B, mp, 0.0, cvec, m));         // B K phi_ref
  ChainArgs ch;
  ch.G = Acl; ch.ldg = m; ch.m = m; ch.pu = 0; ch.T = steps; ch.batch = batch; ch.Zall = Phi;
  ch.z_stride = (int64_t)steps * m; ch.bias = cvec; ch.bias_stride = m;
  bool z0_in_place = false;
  if (lifted_chain_ok(m, 0, 0)) {
    ch.z0 = f0.ptr; ch.z0_stride = f0.ld;
  } else {
    NK_TRY(launch_copy2d(ctx, f0.ptr, f0.ld, Phi, ch.z_stride, batch, m));
    z0_in_place = true;
  }
  const bool try_mw = chain_mw_wanted(ctx, ch);
  std::unique_lock<std::mutex> mw_lock(g_chain_mw_mutex, std::defer_lock);
  if (try_mw) mw_lock.lock();
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool mw = try_mw && attempt == 0;
    NK_TRY(rollout_steps(ctx, ch, z0_in_place, mw));
    // u_t = K (phi_ref - phi_t) for all t: D = 1 phi_ref^T - Phi, U = D K^T
    NK_TRY(launch_ref_minus_traj(ctx, fr.ptr, fr.ld, Phi, ch.z_stride, Dm, ch.z_stride, steps, m, batch));
    NK_TRY(launch_gemm(ctx, false, true, (int64_t)batch * steps, p, m, 1.0, Dm, m, k.ptr, k.ld, 0.0, udev, lduo));
    NK_TRY(launch_gemm(ctx, false, true, (int64_t)batch * steps, d, m, 1.0, Phi, m, mdl->C, m, 0.0, xdev, ldxo));  // x_t = C phi_t
    if (!small) {
      NK_TRY(finish_out(ctx, ox));
      NK_TRY(finish_out(ctx, ou));
    }
    NK_HIP(hipStreamSynchronize(ctx->stream));
    if (!(mw && chain_mw_gave_up(ctx))) break;  // see rollout_impl
  }
  if (small) small_finish(st);
  return NK_OK;
}

int nk_closed_loop(nk_ctx* ctx, const nk_model* mdl, const double* K, const double* phi0, const double* phi_ref,
                   int32_t steps, double* out_x, double* out_u) {
  return nk_closed_loop_batch(ctx, mdl, K, phi0, phi_ref, steps, 1, out_x, out_u);
}

int nk_gemm(nk_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
            int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(A && B && C && M >= 0 && N >= 0 && K >= 0, "nk_gemm: bad argument");
  if (M == 0 || N == 0) return NK_OK;
  MatIn a, b;
  NK_TRY(stage_in(ctx, A, lda, transA ? K : M, transA ? M : K, &a));
  NK_TRY(stage_in(ctx, B, ldb, transB ? N : K, transB ? K : N, &b));
  MatOut c;
  NK_TRY(stage_out(ctx, C, ldc, M, N, &c));
  if (c.host && beta != 0.0)
    NK_HIP(hipMemcpy2DAsync(c.dev, (size_t)c.ld * 8, C, (size_t)ldc * 8, (size_t)N * 8, (size_t)M, hipMemcpyHostToDevice,
                            ctx->stream));
  NK_TRY(launch_gemm(ctx, transA != 0, transB != 0, M, N, K, alpha, a.ptr, a.ld, b.ptr, b.ld, beta, c.dev, c.ld));
  NK_TRY(finish_out(ctx, c));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

int nk_gemm_f32(nk_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                double* Cm, int64_t ldc) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(A && B && Cm && M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && lda >= M && ldb >= N && ldc >= N,
             "nk_gemm_f32: bad argument");
  const ArenaMark mk = arena_mark(ctx);
  const float *Ad = A, *Bd = B;
  double* Cd = Cm;
  double* tmp = nullptr;
  if (!is_device_ptr(A)) {
    NK_TRY(arena_alloc_t(ctx, ((size_t)K * lda + 1) / 2 + 2, &tmp));
    NK_HIP(hipMemcpyAsync(tmp, A, (size_t)K * lda * 4, hipMemcpyHostToDevice, ctx->stream));
    Ad = reinterpret_cast<const float*>(tmp);
  }
  if (!is_device_ptr(B)) {
    NK_TRY(arena_alloc_t(ctx, ((size_t)K * ldb + 1) / 2 + 2, &tmp));
    NK_HIP(hipMemcpyAsync(tmp, B, (size_t)K * ldb * 4, hipMemcpyHostToDevice, ctx->stream));
    Bd = reinterpret_cast<const float*>(tmp);
  }
  const bool c_host = !is_device_ptr(Cm);
  if (c_host) NK_TRY(arena_alloc_t(ctx, (size_t)M * ldc, &Cd));
  TnProblemF pf;
  pf.A = Ad; pf.B = Bd; pf.lda = lda; pf.ldb = ldb; pf.M = (int)M; pf.N = (int)N; pf.C = Cd; pf.ldc = ldc;
  NK_REQUIRE(tnf_fast_ok(pf), "nk_gemm_f32: operands must be 16-byte aligned with leading dimensions that are multiples of 4 and >= 4 columns");
  NK_TRY(launch_gemm_tn_f32_multi(ctx, &pf, 1, K, 0, nullptr, true));
  if (c_host) NK_HIP(hipMemcpyAsync(Cm, Cd, (size_t)M * ldc * 8, hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  arena_release(ctx, mk);
  return NK_OK;
}

int nk_sqrtm_spd(nk_ctx* ctx, const double* P, int64_t ldp, int32_t m, double* S, double* Sinv, int32_t* iters,
                 double* residual) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(P && S && Sinv && m > 0 && ldp >= m, "nk_sqrtm_spd: bad argument");
  MatIn p;
  NK_TRY(stage_in(ctx, P, ldp, m, m, &p));
  double *s = nullptr, *si = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &s));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &si));
  int it = 0;
  double r = 0.0;
  NK_TRY(sqrtm_spd(ctx, p.ptr, p.ld, m, s, si, &it, &r));
  if (iters) *iters = it;
  if (residual) *residual = r;
  const hipMemcpyKind k1 = is_device_ptr(S) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const hipMemcpyKind k2 = is_device_ptr(Sinv) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  NK_HIP(hipMemcpyAsync(S, s, sizeof(double) * m * m, k1, ctx->stream));
  NK_HIP(hipMemcpyAsync(Sinv, si, sizeof(double) * m * m, k2, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

int nk_solve_spd(nk_ctx* ctx, const double* P, int64_t ldp, int32_t m, const double* R, int64_t ldr, int32_t nrhs,
                 double* X, int64_t ldxo) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(P && R && X && m > 0 && nrhs > 0 && ldp >= m && ldr >= nrhs && ldxo >= nrhs, "nk_solve_spd: bad argument");
  MatIn p, r;
  NK_TRY(stage_in(ctx, P, ldp, m, m, &p));
  NK_TRY(stage_in(ctx, R, ldr, m, nrhs, &r));
  double *L = nullptr, *Linv = nullptr, *W = nullptr;
  const int nblk = (m + CHOL_NB - 1) / CHOL_NB;
  const int64_t ldw = nrhs + (nrhs & 1);
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &L));
  NK_TRY(arena_alloc_t(ctx, (size_t)nblk * CHOL_WS, &Linv));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * ldw, &W));
  NK_TRY(launch_copy2d(ctx, p.ptr, p.ld, L, m, m, m));
  NK_TRY(launch_copy2d(ctx, r.ptr, r.ld, W, ldw, m, nrhs));
  double* pivlog = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)m, &pivlog));
  CholSys csys;
  csys.P = L; csys.ldp = m; csys.m = m; csys.Linv = Linv; csys.pivlog = pivlog;
  int rc_chol = cholesky_lower_pair(ctx, &csys, 1);
  if (getenv("NYSKOOP_FORCE_PINV")) rc_chol = NK_ERR_NOT_SPD;  // testing hook: always take the SVD path
  if (rc_chol == NK_ERR_NOT_SPD && ctx->strict_spd != 1) {
    // numerically singular: X = P^+ R with gelsd's cut-off, i.e. X^T = R^T P^+ (P symmetric)
    double *Rt = nullptr, *Xt = nullptr;
    NK_TRY(arena_alloc_t(ctx, (size_t)nrhs * m, &Rt));
    NK_TRY(arena_alloc_t(ctx, (size_t)nrhs * m, &Xt));
    NK_TRY(launch_transpose(ctx, r.ptr, r.ld, Rt, m, m, nrhs));
    PinvInfo pi;
    const double rcond_use = getenv("NYSKOOP_PINV_RCOND") ? atof(getenv("NYSKOOP_PINV_RCOND")) : 2.220446049250313e-16;
    NK_TRY(pinv_right_divide(ctx, p.ptr, p.ld, m, Rt, m, nrhs, Xt, m, rcond_use, &pi));
    if (getenv("NYSKOOP_PINV_TRACE"))
      fprintf(stderr, "[nk pinv] rank %d of %d, sigma_max %.3e, smallest kept %.3e, smallest %.3e, %d sweeps\n", pi.rank, m,
              pi.sigma_max, pi.sigma_min_kept, pi.sigma_min, pi.sweeps);
    if (!pi.converged) {
      set_error("nk_solve_spd: Jacobi SVD of the singular system did not converge in %d sweeps", pi.sweeps);
      return NK_ERR_NO_CONVERGENCE;
    }
    NK_TRY(launch_transpose(ctx, Xt, m, W, ldw, nrhs, m));
  } else {
    NK_TRY(rc_chol);
    NK_TRY(cholesky_solve(ctx, L, m, m, Linv, W, ldw, nrhs));
  }
  NK_HIP(hipMemcpy2DAsync(X, (size_t)ldxo * 8, W, (size_t)ldw * 8, (size_t)nrhs * 8, (size_t)m,
                          is_device_ptr(X) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

__global__ void nk_fill_pattern_kernel(double* p, int64_t n, unsigned seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long x = (unsigned long long)i * 6364136223846793005ULL + seed * 1442695040888963407ULL + 1;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;  // uniform [-0.5, 0.5), full mantissa
  }
}

int nk_bench_gram(nk_ctx* ctx, int64_t n, int32_t m, int32_t p, int32_t d, int32_t reps, double* ms_avg, double* flop) {
  NK_TRY(check_ctx(ctx));
  NK_REQUIRE(n > 0 && m > 0 && p >= 0 && d > 0 && reps > 0 && ms_avg, "nk_bench_gram: bad argument");
  const int mp = m + p;
  const int64_t off_out = (mp + 1) & ~1;
  const int64_t ldf = (off_out + m + 1) & ~(int64_t)1;
  const int64_t ldd = d + (d & 1);
  double *F = nullptr, *Y = nullptr, *G1 = nullptr, *G2t = nullptr, *G3 = nullptr, *G4t = nullptr;
  NK_TRY(arena_alloc_t(ctx, (size_t)n * ldf + 64, &F));
  NK_TRY(arena_alloc_t(ctx, (size_t)n * ldd, &Y));
  NK_TRY(arena_alloc_t(ctx, (size_t)mp * mp, &G1));
  NK_TRY(arena_alloc_t(ctx, (size_t)mp * m, &G2t));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * m, &G3));
  NK_TRY(arena_alloc_t(ctx, (size_t)m * ldd, &G4t));
  hipLaunchKernelGGL(nk_fill_pattern_kernel, dim3(4096), dim3(256), 0, ctx->stream, F, n * ldf, 1u);
  hipLaunchKernelGGL(nk_fill_pattern_kernel, dim3(4096), dim3(256), 0, ctx->stream, Y, n * ldd, 2u);
  NK_HIP(hipGetLastError());
  TnProblem pr[4];
  pr[0].A = F; pr[0].B = F; pr[0].lda = pr[0].ldb = ldf; pr[0].M = pr[0].N = mp; pr[0].C = G1; pr[0].ldc = mp;
  pr[0].tri = TRI_UPPER_MIRROR;
  pr[1].A = F; pr[1].B = F + off_out; pr[1].lda = pr[1].ldb = ldf; pr[1].M = mp; pr[1].N = m; pr[1].C = G2t; pr[1].ldc = m;
  pr[2].A = F + off_out; pr[2].B = F + off_out; pr[2].lda = pr[2].ldb = ldf; pr[2].M = pr[2].N = m; pr[2].C = G3;
  pr[2].ldc = m; pr[2].tri = TRI_UPPER_MIRROR;
  pr[3].A = F + off_out; pr[3].lda = ldf; pr[3].M = m; pr[3].N = d; pr[3].C = G4t; pr[3].ldc = ldd; pr[3].B = Y;
  pr[3].ldb = ldd;
  for (int q = 0; q < 4; ++q) NK_REQUIRE(tn_fast_ok(pr[q]), "nk_bench_gram: shape violates the alignment contract");
  double total = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    float ms = 0.f;
    NK_TRY(launch_gemm_tn_multi(ctx, pr, 4, n, 0, &ms));
    if (r > 0) total += ms;
  }
  NK_HIP(hipStreamSynchronize(ctx->stream));
  *ms_avg = total / reps;
  if (flop) *flop = ((double)mp * (mp + 1) + 2.0 * m * mp + (double)m * (m + 1) + 2.0 * d * m) * (double)n;
  return NK_OK;
}

}  // extern "C"
