// Diagonal-block Cholesky of the blocked factorisation: the stand-alone launch (body: nk_potrf_body.h).
#include "nk_common.h"
#include "nk_potrf_body.h"

namespace nk {

__global__ void __launch_bounds__(64) potrf_diag_kernel(PotrfBatch pb, int blk) { potrf_diag_kernel_body(pb, blk, blockIdx.x, threadIdx.x); }
__global__ void __launch_bounds__(64) potrf_diag_kernel_batched(const nk::ArgPack<PotrfBatch, int>* table) {
  potrf_diag_kernel_body(table[blockIdx.z].v, table[blockIdx.z].rest.v, blockIdx.x, threadIdx.x);
}
static nk::TwinReg potrf_twin_reg(reinterpret_cast<const void*>(static_cast<void (*)(PotrfBatch, int)>(potrf_diag_kernel)),
                                  reinterpret_cast<const void*>(potrf_diag_kernel_batched),
                                  sizeof(nk::ArgPack<PotrfBatch, int>), "potrf_diag_kernel");

// Ajj/lda/nb/Linv: per system (nb <= 0 skips a system); failures are flagged in ctx->d_info[info_base + system]
int launch_potrf_diag_pair(nk_ctx* ctx, double* const* Ajj, const int64_t* lda, const int* nb, double* const* Linv,
                           int nsys, int blk, double* const* plog) {
  PotrfBatch pb;
  for (int q = 0; q < 2; ++q) {
    const bool on = q < nsys;
    pb.A[q] = on ? Ajj[q] : nullptr;
    pb.lda[q] = on ? lda[q] : 0;
    pb.nb[q] = on ? nb[q] : 0;
    pb.Linv[q] = on ? Linv[q] : nullptr;
    pb.info[q] = ctx->d_info + info_base(ctx) + q;
    pb.piv[q] = ctx->d_piv + 2 * (info_base(ctx) + q);
    pb.plog[q] = (on && plog) ? plog[q] : nullptr;
  }
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(2), dim3(64), 0, ctx->stream, pb, blk);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

}  // namespace nk
