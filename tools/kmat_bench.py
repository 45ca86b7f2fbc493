"""The direct-difference kernel-matrix kernel alone (nk::kmat_kernel<KTYPE>), device-resident operands and output, for
rocprofv3 --kernel-trace / --pmc runs (north_star: HBM GB/s of the distance kernel).

    python3 tools/kmat_bench.py duffing   # n = 69 900, m = 200, d = 2, Matern-5/2  (benchmark_lqr_classic.py: HBM-write bound)
    python3 tools/kmat_bench.py cloth     # n = 30 300, m = 500, d = 192, anisotropic RBF (fp64-VALU bound)
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device memory only)
import nys_koop_lqr_amd as nk  # noqa: E402
from nys_koop_lqr_amd import _lib  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "duffing"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "duffing":
    n, m, d, kern = 69900, 200, 2, nk.KernelWrapper([1.0, 1.0]).kernel
elif which == "duffing_rbf":
    n, m, d, kern = 69900, 200, 2, nk.ThreeDimensionalKernel(1.0, 1.0, 1.0, 2).kernel
elif which == "duffing_linear":
    n, m, d, kern = 69900, 200, 2, nk.LinearKernelWrapper(0.5).kernel
else:
    n, m, d, kern = 30300, 500, 192, nk.ThreeDimensionalKernel(10.0, 10.0, 10.0, 192).kernel
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.standard_normal((n, d))).cuda()
B = torch.from_numpy(rng.standard_normal((m, d))).cuda()
out = torch.empty((n, m), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
ctx = nk.get_context()
kd, keep = kern.desc(d)
call = lambda: _lib.check(ctx.lib.nk_kernel_matrix(ctx.handle, C.byref(kd), A.data_ptr(), d, n, B.data_ptr(), d, m,
                                                   out.data_ptr(), m))
call()
t0 = time.perf_counter()
for _ in range(reps):
    call()
dt = (time.perf_counter() - t0) / reps
alg = 8.0 * (n * m + n * d + m * d)
print(f"{which}: n={n} m={m} d={d}: {dt * 1e6:.1f} us per call (host wall, includes launch + sync); algorithmic bytes "
      f"{alg / 1e6:.1f} MB (output {8.0 * n * m / 1e6:.1f} MB); pair-dims {n * m * d / 1e9:.3f} G")
