"""How much of the Gram-form kernel-block launch (gemm_tn_f64_kernel<1>: 2 n m d flop + exp epilogue + n x m store) is the
epilogue?  The same product through the plain TN engine (gemm_tn_f64_kernel<0>: alpha/beta epilogue, same store) and the
kernel block itself, device resident, at the headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
ctx = nk.get_context()
n, m, d = 100000, 2000, 384
dev = torch.device("cuda", 0)
At = torch.randn((d, n), dtype=torch.float64, device=dev) * 0.05   # contraction-major operands, as the fit prepares them
Bt = torch.randn((d, m), dtype=torch.float64, device=dev) * 0.05
Cd = torch.empty((n, m), dtype=torch.float64, device=dev)
def gemm():
    _lib.check(ctx.lib.nk_gemm(ctx.handle, 1, 0, n, m, d, 1.0, At.data_ptr(), n, Bt.data_ptr(), m, 0.0, Cd.data_ptr(), m))
for _ in range(3): gemm()
t0 = time.perf_counter()
for _ in range(10): gemm()
dt = (time.perf_counter() - t0) / 10
print(f"plain TN product {n} x {m} x {d}: {dt * 1e3:.3f} ms per call = {2.0 * n * m * d / dt / 1e12:.1f} TF (epilogue: store only)")
X = torch.randn((n, d), dtype=torch.float64, device=dev); Z = torch.randn((m, d), dtype=torch.float64, device=dev)
k = nk.ThreeDimensionalKernel(20., 20., 20., d).kernel
out = torch.empty((n, m), dtype=torch.float64, device=dev)
ctx.lib.nk_set_kmat_mode(ctx.handle, 0)
for _ in range(3): k(X, Z, out=out)
t0 = time.perf_counter()
for _ in range(10): k(X, Z, out=out)
dt2 = (time.perf_counter() - t0) / 10
print(f"nk_kernel_matrix (direct differences, fp64 VALU) same shape: {dt2 * 1e3:.3f} ms per call")
