"""fit(X, Y) with HOST arrays at the headline shape: fits/s with the uploads pipelined against the passes (NYSKOOP_HOST_PASSES)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
import bench
n, m, d, p = 100000, 2000, 384, 6
X, Y, idx = bench.make_c4(n, d, p, m)
Z = np.ascontiguousarray(Y[idx])
def host_fit():
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20.0, 20.0, 20.0, d), gamma=1e-6, m=m)
    reg.nystrom_centers_output = Z.T
    reg.fit(X, Y)
    return reg.A, reg.B, reg.C, reg.weights
ops = host_fit(); host_fit()
t0 = time.perf_counter()
for _ in range(5): host_fit()
dt = (time.perf_counter() - t0) / 5
print(f"NYSKOOP_HOST_PASSES={os.environ.get('NYSKOOP_HOST_PASSES', '(default 6)')}: {dt * 1e3:.1f} ms per fit = {1 / dt:.2f} fits/s", flush=True)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", f"hostfit_A_{os.environ.get('NYSKOOP_HOST_PASSES', '6')}.npy"), ops[0][:64, :64])
