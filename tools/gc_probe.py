import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(0)
n, d, p, m = 3030, 192, 6, 100
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
gc.collect()
gc.set_debug(gc.DEBUG_STATS)
for rep in range(3):
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(20., 20., 20., d), gamma=1e-5, m=m)
    np.random.seed(1); reg.fit(X, Y)
    x0 = X[0, :d]; Useq = rng.standard_normal((p, 100))
    for _ in range(50): reg.rollout(x0, Useq)
    for gen in (0, 1, 2):
        t0 = time.perf_counter(); k = gc.collect(gen); dt = (time.perf_counter() - t0) * 1e3
        print(f"rep {rep}: gc.collect({gen}) -> {k} objects, {dt:.2f} ms", flush=True)
