"""Diagnostic (GPU box): at the full C4 size compare GPU fits (direct vs Gram-form kernel blocks) with the oracle in
faithful (reference call sequence) and fast (eigh + Cholesky) modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threadpoolctl import threadpool_limits
import nys_koop_lqr_amd as nk
from oracle import nk_oracle as O
from bench import make_c4
relf = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
n, m, d, p = int(os.environ.get("DIAG_N", 100000)), 2000, 384, 6
X, Y, idx = make_c4(n, d, p, m)
ctx = nk.get_context(0)
fits = {}
for mode, name in ((1, "gpu_direct"), (0, "gpu_gram")):
    ctx.set_kmat_mode(mode)
    r = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-6, m=m)
    r.nystrom_centers_output = Y.T[:, idx]; r.fit(X, Y); fits[name] = r
    print(name, r.fit_stats_["sqrt_iters"], r.fit_stats_["sqrt_residual"], flush=True)
ctx.set_kmat_mode(0)
with threadpool_limits(limits=16):
    for faithful, name in ((False, "oracle_fast"), (True, "oracle_faithful")):
        t0 = time.perf_counter()
        r = O.KoopmanNystromOracle(p, kernel=O.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-6, m=m, faithful=faithful)
        r.nystrom_centers_output = Y.T[:, idx]; r.fit(X, Y); fits[name] = r
        print(name, "%.1f s" % (time.perf_counter() - t0), flush=True)
ref = fits["oracle_faithful"]
print("cond(inner) %.2e  cond(K_mm) %.2e" % (np.linalg.cond(ref.stages["inner"]), np.linalg.cond(ref.stages["K_mm"])))
for name, r in fits.items():
    if r is ref: continue
    print(f"{name:16s} vs faithful: A %.2e B %.2e C %.2e W %.2e" % (relf(r.A, ref.A), relf(r.B, ref.B), relf(r.C, ref.C), relf(r.weights, ref.weights)))
print("gpu_gram vs gpu_direct: A %.2e W %.2e" % (relf(fits["gpu_gram"].A, fits["gpu_direct"].A), relf(fits["gpu_gram"].weights, fits["gpu_direct"].weights)))
print("gpu_gram vs oracle_fast: A %.2e W %.2e" % (relf(fits["gpu_gram"].A, fits["oracle_fast"].A), relf(fits["gpu_gram"].weights, fits["oracle_fast"].weights)))
