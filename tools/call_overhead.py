"""Where a 0.13 ms rollout call spends its time: Python wrapper vs the C entry point vs the kernels."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
rng = np.random.default_rng(0)
n, d, p, m = 3030, 192, 6, 100
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-5, m=m)
np.random.seed(1); reg.fit(X, Y)
x0 = X[0, :d]; Useq = rng.standard_normal((p, 100))
def tm(f, reps=200):
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
print("reg.rollout            %.1f us" % tm(lambda: reg.rollout(x0, Useq)))
ctx = _lib.get_context(); h = reg._ensure_model()
x0b = np.ascontiguousarray(x0.reshape(1, d)); Ub = np.ascontiguousarray(Useq.T).reshape(1, 100, p); out = np.empty((1, 100, d))
f = lambda: ctx.lib.nk_rollout(ctx.handle, h, x0b.ctypes.data, d, Ub.ctypes.data, 100, 1, out.ctypes.data, None)
print("nk_rollout (ctypes)    %.1f us" % tm(f))
print("nk_synchronize         %.1f us" % tm(lambda: ctx.lib.nk_synchronize(ctx.handle)))
print("nk_version             %.1f us" % tm(lambda: ctx.lib.nk_version()))
f1 = lambda: ctx.lib.nk_rollout(ctx.handle, h, x0b.ctypes.data, d, Ub.ctypes.data, 1, 1, out.ctypes.data, None)
print("nk_rollout T=1         %.1f us" % tm(f1))
os.environ["NYSKOOP_TRACE"] = "1"
