// Lock-step batching of small fits (the CV units of benchmark_lqr_cloth.py:39-66: 405 fits of 6e9 flop each, where one fit
// is a chain of ~300 launch-bound kernels of 2-64 workgroups that leaves the chip 97 % idle).
//
// A GROUP is a set of contexts that share one HIP stream.  Each member context is driven by its own host thread through
// the ordinary API (nk_nystrom_fit, nk_score_neg_rmse, ...).  Inside the library every stream operation of a member
// (kernel launch, async copy, memset) is RECORDED instead of issued; whenever a member has to wait for the device (a
// stream / event synchronisation) it blocks, and when all members that are inside a unit of work have arrived the
// recorded sequences are merged position by position: launches of the same kernel with the same grid become ONE launch
// whose blockIdx.z selects the member, with the members' argument blocks in a device table.  The merged work is issued
// on the shared stream, the stream is synchronised, and all members continue.  No kernel body knows about batching: a
// kernel K(args...) has a twin K_batched(const ArgPack<args...>* table) that loads table[blockIdx.z] and runs the same
// body (NK_BATCHED_TWIN below), so a batched unit computes exactly the bits of an unbatched one.  Kernels without a twin
// are launched one member at a time.  Multi-stream overlap inside a fit (prep / side streams) is replaced by program
// order on the single shared stream, which is a valid order of the event graph because the host enqueues producers
// before consumers.
//
// Interception is by macro: after this header, hipLaunchKernelGGL / hipMemcpyAsync / hipStreamSynchronize / ... inside the
// library's translation units resolve to the wrappers below, which look at the calling thread's current context
// (nk::tl_ctx, set by every API entry point) and either pass through (no group) or record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

struct nk_ctx;
struct nk_group;

namespace nk {

// ---- POD argument pack shared by host and device ------------------------------------------------------------------
template <typename... Ts> struct ArgPack;
template <> struct ArgPack<> {};
template <typename T, typename... Ts> struct ArgPack<T, Ts...> {
  T v;
  ArgPack<Ts...> rest;
};
template <typename F, typename... Us>
__device__ __forceinline__ void pack_apply(F&& f, const ArgPack<>&, Us... us) { f(us...); }
template <typename F, typename T, typename... Ts, typename... Us>
__device__ __forceinline__ void pack_apply(F&& f, const ArgPack<T, Ts...>& p, Us... us) {
  pack_apply(static_cast<F&&>(f), p.rest, us..., p.v);
}
// byte offsets of the members (for launching the direct kernel from a recorded pack)
inline void pack_offsets(const ArgPack<>&, const char*, std::vector<uint32_t>&) {}
template <typename T, typename... Ts>
inline void pack_offsets(const ArgPack<T, Ts...>& p, const char* base, std::vector<uint32_t>& out) {
  out.push_back((uint32_t)(reinterpret_cast<const char*>(&p.v) - base));
  pack_offsets(p.rest, base, out);
}
inline void pack_fill(ArgPack<>&) {}
template <typename T, typename... Ts, typename A, typename... As>
inline void pack_fill(ArgPack<T, Ts...>& p, A&& a, As&&... as) {
  p.v = static_cast<T>(a);
  pack_fill(p.rest, static_cast<As&&>(as)...);
}

// ---- registry: direct kernel -> batched twin ------------------------------------------------------------------------
void register_twin(const void* direct, const void* twin, size_t pack_bytes, const char* name);
const void* find_twin(const void* direct, size_t* pack_bytes);
struct TwinReg {
  TwinReg(const void* d, const void* t, size_t b, const char* name = "?") { register_twin(d, t, b, name); }
};

extern thread_local nk_ctx* tl_ctx;
bool ctx_recording(const nk_ctx* c);  // member of a group: stream operations are recorded
int group_record_kernel(nk_ctx* c, const void* direct, dim3 grid, dim3 block, size_t lds, const void* pack, size_t bytes,
                        const std::vector<uint32_t>& offsets);
int group_sync(nk_ctx* c);  // barrier + merged flush + real synchronisation
int x_align();               // alignment point after a data-dependent region (no-op outside a group)

// kernel launch: pass-through or record
template <typename... KA, typename... A>
inline void launch_k(void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A&&... args) {
  nk_ctx* c = tl_ctx;
  if (c == nullptr || !ctx_recording(c)) {
    kernel<<<grid, block, lds, stream>>>(static_cast<KA>(args)...);
    return;
  }
  ArgPack<KA...> p;
  memset(static_cast<void*>(&p), 0, sizeof(p));
  pack_fill(p, static_cast<A&&>(args)...);
  std::vector<uint32_t> offs;
  pack_offsets(p, reinterpret_cast<const char*>(&p), offs);
  (void)group_record_kernel(c, reinterpret_cast<const void*>(kernel), grid, block, lds, &p, sizeof(p), offs);
}

// stream-operation wrappers
hipError_t x_memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t x_memcpy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind kind, hipStream_t s);
hipError_t x_memset_async(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t x_stream_sync(hipStream_t s);
hipError_t x_event_sync(hipEvent_t e);
hipError_t x_event_record(hipEvent_t e, hipStream_t s);
hipError_t x_stream_wait_event(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t x_event_elapsed(float* ms, hipEvent_t a, hipEvent_t b);

}  // namespace nk

// A batched twin for kernel `name` whose parameter types are listed: same body, arguments from table[blockIdx.z].
// The kernel itself must be written as   __global__ void name(T1 a1, ...) { name##_body(a1, ...); }
#define NK_BATCHED_TWIN(name, bounds, ...)                                                                   \
  __global__ void __launch_bounds__ bounds name##_batched(const nk::ArgPack<__VA_ARGS__>* table) {           \
    const nk::ArgPack<__VA_ARGS__> p = table[blockIdx.z];                                                    \
    nk::pack_apply([](auto... a) { name##_body(a...); }, p);                                                 \
  }                                                                                                          \
  static nk::TwinReg name##_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(__VA_ARGS__)>(name)), \
                                     reinterpret_cast<const void*>(name##_batched), sizeof(nk::ArgPack<__VA_ARGS__>), #name);

#ifndef NK_LOCKSTEP_NO_MACROS
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) nk::launch_k(kernel, grid, block, lds, stream, __VA_ARGS__)
#define hipMemcpyAsync(...) nk::x_memcpy_async(__VA_ARGS__)
#define hipMemcpy2DAsync(...) nk::x_memcpy2d_async(__VA_ARGS__)
#define hipMemsetAsync(...) nk::x_memset_async(__VA_ARGS__)
#define hipStreamSynchronize(...) nk::x_stream_sync(__VA_ARGS__)
#define hipEventSynchronize(...) nk::x_event_sync(__VA_ARGS__)
#define hipEventRecord(...) nk::x_event_record(__VA_ARGS__)
#define hipStreamWaitEvent(...) nk::x_stream_wait_event(__VA_ARGS__)
#define hipEventElapsedTime(...) nk::x_event_elapsed(__VA_ARGS__)
#endif
