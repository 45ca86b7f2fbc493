// Lock-step groups (nk_lockstep.h): recording of stream operations per member, the barrier at synchronisation points and the
// merged flush onto the shared stream.
#define NK_LOCKSTEP_NO_MACROS
#include "nk_common.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

namespace nk {

thread_local nk_ctx* tl_ctx = nullptr;

struct TwinInfo {
  const void* twin;
  size_t bytes;
  const char* name;
};
static std::unordered_map<const void*, TwinInfo>& twin_map() {
  static std::unordered_map<const void*, TwinInfo> m;
  return m;
}
void register_twin(const void* direct, const void* twin, size_t pack_bytes, const char* name) {
  twin_map()[direct] = TwinInfo{twin, pack_bytes, name};
}
const void* find_twin(const void* direct, size_t* pack_bytes) {
  auto it = twin_map().find(direct);
  if (it == twin_map().end()) return nullptr;
  // diagnostics: NYSKOOP_GROUP_ONLY=a,b merges only kernels whose name contains a or b; NYSKOOP_GROUP_SKIP=a,b leaves those out
  static const char* only = getenv("NYSKOOP_GROUP_ONLY");
  static const char* skip = getenv("NYSKOOP_GROUP_SKIP");
  auto matches = [&](const char* list) {
    std::string l(list);
    size_t b = 0;
    while (b <= l.size()) {
      size_t e = l.find(',', b);
      if (e == std::string::npos) e = l.size();
      if (e > b && strstr(it->second.name, l.substr(b, e - b).c_str())) return true;
      b = e + 1;
    }
    return false;
  };
  if (only && !matches(only)) return nullptr;
  if (skip && matches(skip)) return nullptr;
  if (pack_bytes) *pack_bytes = it->second.bytes;
  return it->second.twin;
}

}  // namespace nk

enum { OP_KERNEL = 0, OP_MEMCPY = 1, OP_MEMCPY2D = 2, OP_MEMSET = 3 };

struct GroupOp {
  int kind = OP_KERNEL;
  // kernel
  const void* fn = nullptr;
  const void* twin = nullptr;
  dim3 grid, block;
  uint32_t lds = 0, psize = 0;
  size_t poff = 0;         // offset of the argument pack in the member's blob
  uint32_t noff = 0;       // number of arguments
  size_t ooff = 0;         // offset of their byte offsets in the member's offset store
  // copies / memset
  void* dst = nullptr;
  const void* src = nullptr;
  size_t bytes = 0, dpitch = 0, spitch = 0, width = 0, height = 0;
  int value = 0;
  hipMemcpyKind mk = hipMemcpyDefault;
};

// Merge key of a recorded launch: two launches share ONE merged launch only if every field is equal (the merged launch
// takes its geometry from the first member).  The hash below only buckets; equality is exact.
struct SigKey {
  const void* twin = nullptr;
  unsigned gx = 0, gy = 0, bx = 0, by = 0, bz = 0;
  uint32_t lds = 0, psize = 0;
  bool operator==(const SigKey& o) const {
    return twin == o.twin && gx == o.gx && gy == o.gy && bx == o.bx && by == o.by && bz == o.bz && lds == o.lds &&
           psize == o.psize;
  }
};
struct SigHash {
  size_t operator()(const SigKey& k) const {
    uint64_t h = reinterpret_cast<uint64_t>(k.twin) * 0x9E3779B97F4A7C15ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0xBF58476D1CE4E5B9ull; h ^= h >> 29; };
    mix(k.gx); mix(k.gy); mix(k.bx); mix(((uint64_t)k.by << 32) | k.bz); mix(((uint64_t)k.lds << 32) | k.psize);
    return (size_t)h;
  }
};

struct nk_member_state {
  std::vector<GroupOp> ops;
  std::vector<char> blob;
  std::vector<uint32_t> offs;
  bool entered = false;
  bool waiting = false;
  uint64_t align_seen = 0;
};

struct nk_group {
  std::mutex mu;
  std::condition_variable cv;
  int device = 0;
  int active = 0;    // members inside a unit of work (nk_group_enter .. nk_group_leave)
  int waiting = 0;   // of those, blocked at a synchronisation point
  uint64_t epoch = 0;
  hipStream_t stream = nullptr;
  std::vector<nk_ctx*> members;
  char* h_tab = nullptr;  // page-locked staging of the argument tables of one flush
  char* d_tab = nullptr;
  size_t tab_cap = 0;
  int live = 0;           // members not yet destroyed
  int at_align = 0;       // members parked at an alignment point
  uint64_t align_gen = 1;
  // HIP's last error is per host thread and a merged launch is issued by whichever member arrived last, so the first
  // failure of a flush (launch / copy return codes, hipGetLastError and the stream synchronisation, all taken in the
  // flushing thread) is kept per epoch and handed to EVERY member that epoch releases.
  int epoch_err = 0;  // NK_OK or NK_ERR_HIP: verdict of the last completed epoch
  // statistics
  uint64_t n_flush = 0, n_launch_merged = 0, n_launch_single = 0, n_units_merged = 0;
};

namespace nk {

bool ctx_recording(const nk_ctx* c) { return c->group != nullptr; }

static nk_member_state* st(nk_ctx* c) { return c->gstate; }

int group_record_kernel(nk_ctx* c, const void* direct, dim3 grid, dim3 block, size_t lds, const void* pack, size_t bytes,
                        const std::vector<uint32_t>& offsets) {
  nk_member_state* s = st(c);
  GroupOp op;
  op.kind = OP_KERNEL;
  op.fn = direct;
  size_t tb = 0;
  op.twin = (grid.z == 1) ? find_twin(direct, &tb) : nullptr;
  if (op.twin && tb != bytes) op.twin = nullptr;
  op.grid = grid; op.block = block; op.lds = (uint32_t)lds; op.psize = (uint32_t)bytes;
  op.poff = (s->blob.size() + 15) & ~(size_t)15;
  s->blob.resize(op.poff + bytes);
  memcpy(s->blob.data() + op.poff, pack, bytes);
  op.noff = (uint32_t)offsets.size();
  op.ooff = s->offs.size();
  s->offs.insert(s->offs.end(), offsets.begin(), offsets.end());
  s->ops.push_back(op);
  return NK_OK;
}

static hipError_t launch_direct(nk_group* g, nk_member_state* s, const GroupOp& op) {
  void* args[64];
  char* base = s->blob.data() + op.poff;
  for (uint32_t i = 0; i < op.noff && i < 64; ++i) args[i] = base + s->offs[op.ooff + i];
  return hipLaunchKernel(op.fn, op.grid, op.block, args, op.lds, g->stream);
}

// test hook (tests/test_gpu_round3.py): NYSKOOP_GROUP_TEST_FAIL_MERGED=k makes the k-th merged launch of the process
// invalid (a block of 4096 threads), so that the error path of a flush can be exercised on a healthy device
static bool test_fail_this_merged_launch() {
  static const char* e = getenv("NYSKOOP_GROUP_TEST_FAIL_MERGED");
  if (!e) return false;
  static std::atomic<long> count{0};
  return ++count == atol(e);
}

// Issue the recorded operations of `ready` members on the shared stream, merging equal kernel launches position by position.
// Called with g->mu held; the caller synchronises the stream afterwards (the staging block is reused by the next flush).
// Returns the first HIP error of the flush (hipSuccess if none).
static hipError_t flush_locked(nk_group* g, const std::vector<nk_ctx*>& ready) {
  size_t maxlen = 0;
  for (nk_ctx* c : ready) maxlen = std::max(maxlen, st(c)->ops.size());
  if (maxlen == 0) return hipSuccess;
  hipError_t first_err = hipSuccess;
  auto note = [&](hipError_t e) { if (e != hipSuccess && first_err == hipSuccess) first_err = e; };
  (void)hipGetLastError();  // errors of this thread's earlier, unrelated calls are not this flush's
  g->n_flush++;
  struct Item {
    bool merged;
    const void* twin; dim3 grid, block; uint32_t lds; size_t tab_off; int count;  // merged launch
    nk_ctx* c; size_t idx;                                                        // single operation
  };
  std::vector<Item> items;
  size_t used = 0;
  // lay out argument tables and remember the order; emit = one upload, then everything in order
  auto emit = [&](bool reuse_staging) {
    const bool host_tab = getenv("NYSKOOP_GROUP_HOST_TABLE") != nullptr;
    if (used > 0 && !host_tab) note(hipMemcpyAsync(g->d_tab, g->h_tab, used, hipMemcpyHostToDevice, g->stream));
    for (const Item& it : items) {
      if (it.merged) {
        const void* tab = (host_tab ? g->h_tab : g->d_tab) + it.tab_off;
        void* args[1] = {reinterpret_cast<void*>(&tab)};
        dim3 grid = it.grid;
        grid.z = (unsigned)it.count;
        dim3 block = it.block;
        if (test_fail_this_merged_launch()) block.x = 4096;
        note(hipLaunchKernel(it.twin, grid, block, args, it.lds, g->stream));
        g->n_launch_merged++;
        g->n_units_merged += (uint64_t)it.count;
      } else {
        nk_member_state* s = st(it.c);
        const GroupOp& op = s->ops[it.idx];
        switch (op.kind) {
          case OP_KERNEL: note(launch_direct(g, s, op)); g->n_launch_single++; break;
          case OP_MEMCPY: note(hipMemcpyAsync(op.dst, op.src, op.bytes, op.mk, g->stream)); break;
          case OP_MEMCPY2D:
            note(hipMemcpy2DAsync(op.dst, op.dpitch, op.src, op.spitch, op.width, op.height, op.mk, g->stream));
            break;
          case OP_MEMSET: note(hipMemsetAsync(op.dst, op.value, op.bytes, g->stream)); break;
        }
      }
    }
    items.clear();
    note(hipGetLastError());
    if (reuse_staging) {
      note(hipStreamSynchronize(g->stream));
      used = 0;
    }
  };
  // Content-aligned merge.  Every member has a cursor into its recorded sequence (its own order is preserved).  At each
  // step: operations that cannot merge (copies, kernels without twin) are issued as they come; among the kernel launches
  // at the cursors, one signature is chosen and issued for all members currently at it.  A signature that some other
  // member still has AHEAD of its cursor is held back (that member will catch up and join the launch) unless nothing else
  // can move -- this re-aligns sequences that differ by a few member-specific operations (fold-dependent copies, an
  // extra iteration) instead of letting one shift break every later merge.
  const size_t R = ready.size();
  std::vector<size_t> cur(R, 0);
  auto sig_of = [](const GroupOp& o) -> SigKey {  // (grid.z == 1 for every launch that has a twin, group_record_kernel)
    SigKey k;
    k.twin = o.twin; k.gx = o.grid.x; k.gy = o.grid.y; k.bx = o.block.x; k.by = o.block.y; k.bz = o.block.z;
    k.lds = o.lds; k.psize = o.psize;
    return k;
  };
  std::vector<std::unordered_map<SigKey, int, SigHash>> remaining(R);  // mergeable signatures at or after the cursor, with counts
  for (size_t a = 0; a < R; ++a)
    for (const GroupOp& o : st(ready[a])->ops)
      if (o.kind == OP_KERNEL && o.twin) remaining[a][sig_of(o)]++;
  std::vector<size_t> grp;
  for (;;) {
    // 1. single operations at the cursors
    bool any = false;
    for (size_t a = 0; a < R; ++a) {
      nk_member_state* sa = st(ready[a]);
      while (cur[a] < sa->ops.size() && !(sa->ops[cur[a]].kind == OP_KERNEL && sa->ops[cur[a]].twin)) {
        items.push_back(Item{false, nullptr, dim3(), dim3(), 0, 0, 0, ready[a], cur[a]});
        ++cur[a];
      }
      any = any || cur[a] < sa->ops.size();
    }
    if (!any) break;
    // 2. choose a signature among the launches at the cursors
    std::unordered_map<SigKey, int, SigHash> holders;
    for (size_t a = 0; a < R; ++a) {
      nk_member_state* sa = st(ready[a]);
      if (cur[a] < sa->ops.size()) holders[sig_of(sa->ops[cur[a]])]++;
    }
    SigKey best, best_any;
    int best_n = -1, best_any_n = -1;
    if (holders.size() == 1) {  // the common case: everybody is at the same launch
      best = holders.begin()->first;
      best_n = holders.begin()->second;
    } else
    for (auto& kv : holders) {
      bool ahead_elsewhere = false;
      for (size_t b = 0; b < R && !ahead_elsewhere; ++b) {
        nk_member_state* sb = st(ready[b]);
        if (cur[b] >= sb->ops.size() || sig_of(sb->ops[cur[b]]) == kv.first) continue;
        auto it = remaining[b].find(kv.first);
        ahead_elsewhere = it != remaining[b].end() && it->second > 0;
      }
      if (kv.second > best_any_n) { best_any_n = kv.second; best_any = kv.first; }
      if (!ahead_elsewhere && kv.second > best_n) { best_n = kv.second; best = kv.first; }
    }
    const SigKey pick = best_n > 0 ? best : best_any;
    grp.clear();
    for (size_t a = 0; a < R; ++a) {
      nk_member_state* sa = st(ready[a]);
      if (cur[a] < sa->ops.size() && sig_of(sa->ops[cur[a]]) == pick) grp.push_back(a);
    }
    const GroupOp& o0 = st(ready[grp[0]])->ops[cur[grp[0]]];
    if (grp.size() == 1) {
      items.push_back(Item{false, nullptr, dim3(), dim3(), 0, 0, 0, ready[grp[0]], cur[grp[0]]});
    } else {
      const size_t stride = o0.psize;  // packs are PODs; the table is an array of them
      used = (used + 63) & ~(size_t)63;
      if (used + stride * grp.size() > g->tab_cap) emit(true);
      for (size_t k = 0; k < grp.size(); ++k) {
        nk_member_state* sk = st(ready[grp[k]]);
        memcpy(g->h_tab + used + k * stride, sk->blob.data() + sk->ops[cur[grp[k]]].poff, stride);
      }
      items.push_back(Item{true, o0.twin, o0.grid, o0.block, o0.lds, used, (int)grp.size(), nullptr, 0});
      used += stride * grp.size();
    }
    for (size_t a : grp) {
      remaining[a][pick]--;
      ++cur[a];
    }
  }
  emit(false);
  for (nk_ctx* c : ready) {
    nk_member_state* s = st(c);
    s->ops.clear();
    s->blob.clear();
    s->offs.clear();
  }
  return first_err;
}

// flush + real synchronisation of the shared stream; NK_OK or NK_ERR_HIP (message in the calling thread's error slot)
static int flush_and_sync_locked(nk_group* g, const std::vector<nk_ctx*>& ready) {
  hipError_t e = flush_locked(g, ready);
  const hipError_t es = hipStreamSynchronize(g->stream);
  if (e == hipSuccess) e = es;
  if (e == hipSuccess) return NK_OK;
  set_error("lock-step group: a merged flush failed (%s); every member of the round gets this error", hipGetErrorString(e));
  return NK_ERR_HIP;
}
// release the members that were waiting for the flush just done (g->mu held): they all see its verdict
static void release_waiters_locked(nk_group* g, const std::vector<nk_ctx*>& ready, int verdict) {
  for (nk_ctx* m : ready) st(m)->waiting = false;
  g->waiting = 0;
  g->epoch_err = verdict;
  g->epoch++;
}
static std::vector<nk_ctx*> waiting_members(nk_group* g) {
  std::vector<nk_ctx*> ready;
  for (nk_ctx* m : g->members)
    if (m && st(m)->waiting) ready.push_back(m);
  return ready;
}

int group_sync(nk_ctx* c) {
  nk_group* g = c->group;
  nk_member_state* s = st(c);
  std::unique_lock<std::mutex> lk(g->mu);
  if (!s->entered)  // outside a unit of work: nobody to wait for
    return flush_and_sync_locked(g, std::vector<nk_ctx*>{c});
  s->waiting = true;
  g->waiting++;
  if (g->waiting >= g->active) {
    const std::vector<nk_ctx*> ready = waiting_members(g);
    const int rc = flush_and_sync_locked(g, ready);
    release_waiters_locked(g, ready, rc);
    lk.unlock();
    g->cv.notify_all();
    return rc;
  }
  const uint64_t ep = g->epoch;
  g->cv.wait(lk, [&] { return g->epoch != ep; });
  // (the next epoch cannot complete before this member arrives again, so epoch_err still belongs to the epoch that
  // released it)
  const int rc = g->epoch_err;
  if (rc != NK_OK) set_error("lock-step group: the merged flush of this round failed in another member's thread");
  return rc;
}

// Alignment point: a member that leaves a data-dependent region (an iteration whose length differs from unit to unit)
// waits here until every member inside a unit has left it too, so that the launch sequences are position-aligned again
// and keep merging.  While it waits it keeps taking part in the flush barrier (with nothing to flush), so the members
// still iterating make progress.
int group_align(nk_ctx* c) {
  nk_group* g = c->group;
  if (!g) return NK_OK;
  nk_member_state* s = st(c);
  {
    std::unique_lock<std::mutex> lk(g->mu);
    if (!s->entered) return NK_OK;
    s->align_seen = g->align_gen;
    g->at_align++;
    if (g->at_align >= g->active) {  // the last one: open the gate, then release the others through a (possibly empty) flush
      g->at_align = 0;
      g->align_gen++;
    }
  }
  for (;;) {
    int rc = group_sync(c);
    if (rc != NK_OK) return rc;
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->align_gen != s->align_seen || !s->entered) return NK_OK;
  }
}

static inline nk_ctx* rec_ctx() {
  nk_ctx* c = tl_ctx;
  return (c && c->group) ? c : nullptr;
}

int x_align() {
  nk_ctx* c = rec_ctx();
  return c ? group_align(c) : NK_OK;
}

// Copies a recorded sequence can MERGE: a plain hipMemcpyAsync is one stream operation per member (45 000 of them in six
// sweeps of the 405-unit grid, 13 % of the sweep's GPU time at ~5 us each, none of them shared), a copy KERNEL has a batched
// twin like every other kernel.  Device-to-device copies and the small device-to-host copies into the context's own pinned
// mirrors (h_scalars / h_info: host memory the device can store to) are therefore recorded as launches of this kernel.
__device__ __forceinline__ void copy_words_kernel_body(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n) { copy_words_kernel_body(src, dst, n); }
NK_BATCHED_TWIN(copy_words_kernel, (256), const uint32_t*, uint32_t*, int64_t)

static bool ctx_pinned_mirror(const nk_ctx* c, const void* p, size_t bytes) {
  const char* q = static_cast<const char*>(p);
  const char* hs = reinterpret_cast<const char*>(c->h_scalars);
  const char* hi = reinterpret_cast<const char*>(c->h_info);
  return (hs && q >= hs && q + bytes <= hs + 64 * sizeof(double)) || (hi && q >= hi && q + bytes <= hi + 64 * sizeof(int));
}

hipError_t x_memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
  nk_ctx* c = rec_ctx();
  if (!c) return hipMemcpyAsync(dst, src, bytes, kind, s);
  static const bool as_kernels = !(getenv("NYSKOOP_GROUP_COPY_KERNELS") && getenv("NYSKOOP_GROUP_COPY_KERNELS")[0] == '0');
  const bool words = bytes > 0 && bytes % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 4 == 0;
  if (as_kernels && words &&
      (kind == hipMemcpyDeviceToDevice || (kind == hipMemcpyDeviceToHost && bytes <= 512 && ctx_pinned_mirror(c, dst, bytes)))) {
    const int64_t n = (int64_t)(bytes / 4);
    const int64_t blocks = (n + 1023) / 1024;  // four words per thread
    ArgPack<const uint32_t*, uint32_t*, int64_t> pk;
    memset(static_cast<void*>(&pk), 0, sizeof(pk));
    pack_fill(pk, static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst), n);
    std::vector<uint32_t> offs;
    pack_offsets(pk, reinterpret_cast<const char*>(&pk), offs);
    (void)group_record_kernel(c, reinterpret_cast<const void*>(copy_words_kernel), dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))),
                              dim3(256), 0, &pk, sizeof(pk), offs);
    return hipSuccess;
  }
  GroupOp op;
  op.kind = OP_MEMCPY; op.dst = dst; op.src = src; op.bytes = bytes; op.mk = kind;
  st(c)->ops.push_back(op);
  return hipSuccess;
}
hipError_t x_memcpy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind kind, hipStream_t s) {
  nk_ctx* c = rec_ctx();
  if (!c) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, s);
  GroupOp op;
  op.kind = OP_MEMCPY2D; op.dst = dst; op.src = src; op.dpitch = dpitch; op.spitch = spitch; op.width = width;
  op.height = height; op.mk = kind;
  st(c)->ops.push_back(op);
  return hipSuccess;
}
hipError_t x_memset_async(void* dst, int value, size_t bytes, hipStream_t s) {
  nk_ctx* c = rec_ctx();
  if (!c) return hipMemsetAsync(dst, value, bytes, s);
  GroupOp op;
  op.kind = OP_MEMSET; op.dst = dst; op.value = value; op.bytes = bytes;
  st(c)->ops.push_back(op);
  return hipSuccess;
}
hipError_t x_stream_sync(hipStream_t s) {
  nk_ctx* c = rec_ctx();
  if (!c) return hipStreamSynchronize(s);
  return group_sync(c) == NK_OK ? hipSuccess : hipErrorUnknown;
}
hipError_t x_event_sync(hipEvent_t e) {
  nk_ctx* c = rec_ctx();
  if (!c) return hipEventSynchronize(e);
  return group_sync(c) == NK_OK ? hipSuccess : hipErrorUnknown;
}
hipError_t x_event_record(hipEvent_t e, hipStream_t s) {
  if (rec_ctx()) return hipSuccess;  // one shared stream: program order is the order
  return hipEventRecord(e, s);
}
hipError_t x_stream_wait_event(hipStream_t s, hipEvent_t e, unsigned flags) {
  if (rec_ctx()) return hipSuccess;
  return hipStreamWaitEvent(s, e, flags);
}
hipError_t x_event_elapsed(float* ms, hipEvent_t a, hipEvent_t b) {
  if (rec_ctx()) {
    if (ms) *ms = 0.f;  // no per-member device timing inside a group
    return hipSuccess;
  }
  return hipEventElapsedTime(ms, a, b);
}

// ---- group life cycle (called from nk_api.hip) ---------------------------------------------------------------------
nk_group* group_new(int device, int size) {
  nk_group* g = new nk_group();
  g->device = device;
  g->members.assign((size_t)size, nullptr);
  g->live = size;
  if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { delete g; return nullptr; }
  g->tab_cap = (size_t)8 << 20;
  if (hipHostMalloc(reinterpret_cast<void**>(&g->h_tab), g->tab_cap, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&g->d_tab), g->tab_cap) != hipSuccess) {
    if (g->h_tab) (void)hipHostFree(g->h_tab);
    (void)hipStreamDestroy(g->stream);
    delete g;
    return nullptr;
  }
  return g;
}
void group_attach(nk_group* g, int slot, nk_ctx* c) {
  g->members[(size_t)slot] = c;
  c->group = g;
  c->gstate = new nk_member_state();
}
static std::atomic<uint64_t> g_counters[CNT_N];
void count_event(int which) { if (which >= 0 && which < CNT_N) g_counters[which].fetch_add(1, std::memory_order_relaxed); }
uint64_t read_counter(int which) { return g_counters[which].load(std::memory_order_relaxed); }
hipError_t real_stream_sync(hipStream_t s) { return hipStreamSynchronize(s); }
hipError_t real_event_sync(hipEvent_t e) { return hipEventSynchronize(e); }
hipStream_t group_stream(nk_group* g) { return g->stream; }
// member going away: flush what it recorded, leave the barrier set, free the group with its last member
void group_detach(nk_ctx* c) {
  nk_group* g = c->group;
  if (!g) return;
  bool last = false;
  {
    std::unique_lock<std::mutex> lk(g->mu);
    nk_member_state* s = st(c);
    if (!s->ops.empty()) (void)flush_and_sync_locked(g, std::vector<nk_ctx*>{c});  // a context being destroyed: nobody to tell
    if (s->entered) { s->entered = false; g->active--; }
    for (auto& m : g->members) if (m == c) m = nullptr;
    last = --g->live == 0;
    // members still waiting may now be complete
    if (!last && g->active > 0 && g->waiting >= g->active) {
      const std::vector<nk_ctx*> ready = waiting_members(g);
      const int rc = flush_and_sync_locked(g, ready);
      release_waiters_locked(g, ready, rc);
      g->cv.notify_all();
    }
  }
  delete c->gstate;
  c->gstate = nullptr;
  c->group = nullptr;
  if (last) {
    (void)hipStreamSynchronize(g->stream);
    (void)hipFree(g->d_tab);
    (void)hipHostFree(g->h_tab);
    (void)hipStreamDestroy(g->stream);
    delete g;
  }
}
int group_enter(nk_ctx* c) {
  nk_group* g = c->group;
  if (!g) return NK_OK;
  std::lock_guard<std::mutex> lk(g->mu);
  if (!st(c)->entered) { st(c)->entered = true; g->active++; }
  return NK_OK;
}
int group_leave(nk_ctx* c) {
  nk_group* g = c->group;
  if (!g) return NK_OK;
  std::unique_lock<std::mutex> lk(g->mu);
  nk_member_state* s = st(c);
  if (!s->entered) return NK_OK;
  int rc_own = NK_OK;
  if (!s->ops.empty())  // work recorded after the last synchronisation point
    rc_own = flush_and_sync_locked(g, std::vector<nk_ctx*>{c});
  s->entered = false;
  g->active--;
  if (g->active > 0 && g->at_align >= g->active) {  // the others were parked at an alignment point waiting for this member
    g->at_align = 0;
    g->align_gen++;
  }
  if (g->active > 0 && g->waiting >= g->active) {  // the others were only waiting for this member
    const std::vector<nk_ctx*> ready = waiting_members(g);
    const int rc = flush_and_sync_locked(g, ready);
    release_waiters_locked(g, ready, rc);
    lk.unlock();
    g->cv.notify_all();
  }
  return rc_own;
}
void group_stats(nk_ctx* c, uint64_t out[4]) {
  nk_group* g = c->group;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!g) return;
  std::lock_guard<std::mutex> lk(g->mu);
  out[0] = g->n_flush; out[1] = g->n_launch_merged; out[2] = g->n_launch_single; out[3] = g->n_units_merged;
}

}  // namespace nk
