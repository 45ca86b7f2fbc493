"""One model at the real cloth size (m = 500, d = 192, p = 6): kernels of a 100-step rollout (run under rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(0)
n, d, p, m = 6000, 192, 6, 500
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-5, m=m)
np.random.seed(1); reg.fit(X, Y)
x0 = X[0, :d]; Useq = rng.standard_normal((p, 100))
for _ in range(3): reg.rollout(x0, Useq)
os.environ["NYSKOOP_TRACE"] = "1"
t0 = time.perf_counter()
for _ in range(5): reg.rollout(x0, Useq)
print("rollout m=500 T=100: %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
