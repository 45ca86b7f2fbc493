"""Where do the sporadic ~70 ms calls come from?  Per-call latency of (a) a call without device work, (b) a small-staged
rollout (no hipMemcpy), (c) a rollout whose operands go through hipMemcpy2DAsync of pageable memory, with the Python
garbage collector on and off."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
rng = np.random.default_rng(0)
def dist(name, f, reps):
    f(); f()
    ts = np.empty(reps)
    for i in range(reps):
        t0 = time.perf_counter(); f(); ts[i] = (time.perf_counter() - t0) * 1e3
    slow = np.nonzero(ts > 20 * np.median(ts))[0]
    print(f"{name}: {reps} calls, median {np.median(ts):.3f} ms, p99.9 {np.percentile(ts, 99.9):.3f}, max {ts.max():.3f}; "
          f"calls > 20 x median: {[(int(i), round(float(ts[i]), 1)) for i in slow[:8]]}", flush=True)
ctx = _lib.get_context()
n, d, p, m = 3030, 192, 6, 100
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-5, m=m)
np.random.seed(1); reg.fit(X, Y)
x0 = X[0, :d]; Useq = rng.standard_normal((p, 100))
A = rng.standard_normal((500, 500)) * 0.04; B = rng.standard_normal((500, 6)); Cm = rng.standard_normal((6, 500))
z0 = rng.standard_normal((4, 500)); U4 = rng.standard_normal((4, 100, 6))
for gc_on in (True, False):
    (gc.enable if gc_on else gc.disable)()
    tag = "gc on " if gc_on else "gc off"
    dist(f"[{tag}] nk_synchronize only", lambda: ctx.lib.nk_synchronize(ctx.handle), 20000)
    dist(f"[{tag}] model rollout m=100 (page-locked block, 2 launches)", lambda: reg.rollout(x0, Useq), 3000)
    dist(f"[{tag}] linear_rollout m=500 batch 4 (hipMemcpy2DAsync of 2 MB operands)", lambda: nk.linear_rollout(A, B, Cm, z0, U4), 1500)
